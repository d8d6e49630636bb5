// TEST INFRASTRUCTURE — a stand-in for librccl that lets SEVERAL RANKS SHARE ONE GPU, stream-ordered like the real thing.
//
// RCCL refuses two ranks on one device, so on a one-GPU box the multi-rank tests used to run over a BLOCKING transport (a host
// callback staged through gloo): the loop whose trust-region decisions are taken on the device (solver.hip), its stamp polling and
// its exchanges "enqueued like kernels" had therefore never run asynchronously with more than one rank.  This library exports the
// eight nccl* symbols rsba_amd/csrc/exchange_rccl.hip resolves (RSBA_RCCL_LIB=<this .so>); its ncclAllReduce ENQUEUES its kernels on
// the caller's stream and returns at once (every wait in a one-workgroup gate kernel in front of the kernel it guards):
//   publish: copy the buffer into this rank's staging block (device memory every rank has opened through hipIpc), then raise
//            this rank's "arrived" word to the collective's sequence number;
//   reduce : wait (on the device) until every rank's word has reached the number, then out[i] = the ranks' values combined in RANK
//            ORDER (so every rank computes the same bits), and raise this rank's "done" word.
// Staging blocks are double-buffered by sequence parity; before a block is written again its readers' "done" words are awaited.
// Device waits are bounded (RSBA_MOCK_RCCL_TIMEOUT_S, default 20 s): a rank that never arrives makes the others give up and
// report it (the error word is checked at the next call and at ncclCommDestroy) instead of hanging the GPU.
// Nothing here is part of the product: librsba_amd.so neither links nor names this file.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef enum { ncclFloat64 = 8 } ncclDataType_t;
}

namespace {

constexpr int kMaxRanks = 16;
constexpr size_t kCap = (size_t)4 << 20;   // doubles per staging block (32 MB); larger collectives go in pieces

struct Words { unsigned long long arrived, done, error, pad; unsigned int count_pub, count_red, pad2[2]; };   // device memory, at the head of a rank's block

struct Board {   // the rendezvous file in /dev/shm
  std::atomic<int> ready[kMaxRanks];
  std::atomic<int> opened[kMaxRanks];
  hipIpcMemHandle_t handle[kMaxRanks];
};

struct Peers { const double* stage[kMaxRanks]; Words* words[kMaxRanks]; };

}  // namespace

struct ncclComm {
  int rank = 0, world = 1, device = 0;
  char path[128] = {0};
  Board* board = nullptr;
  void* block = nullptr;          // this rank's allocation: Words | stage[2][kCap]
  void* mapped[kMaxRanks] = {};
  Peers peers{};
  unsigned long long seq = 0;     // collectives (pieces) issued so far
  long long timeout_ticks = 0;    // 100 MHz ticks
};

namespace {

__device__ __forceinline__ bool wait_for(const unsigned long long* word, unsigned long long want, long long timeout) {
  const long long t0 = wall_clock64();
  for (;;) {
    if (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return true;
    if (wall_clock64() - t0 > timeout) return false;
    __builtin_amdgcn_s_sleep(32);
  }
}

// The waits run in ONE-workgroup kernels of their own, in front of the kernels that copy and combine (round 6).  They used to sit in
// thread 0 of EVERY workgroup of the publish / reduce kernels — up to 512 workgroups per rank spinning while they hold their slots:
// with five to eight ranks on one GPU the spinning workgroups of the early ranks filled the device, the publish kernels of the late ranks
// could not start, and everybody waited for everybody until the bound ran out (20 s) and the collective was skipped — each rank then
// went on with its own, unreduced numbers (five and eight ranks: every time; two, three, four, six: never, by luck of the arrival order).
// which = 0: wait until every rank has finished reading what this parity's staging block held two collectives ago ("done" >= want);
// which = 1: until every rank's block of this collective is there ("arrived" >= want).
__global__ __launch_bounds__(64) void gate_kernel(Peers p, int me, int world, unsigned long long want, int which, unsigned long long seq, long long timeout) {
  if (threadIdx.x != 0) return;
  Words* mine = p.words[me];
  if (__hip_atomic_load(&mine->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;   // (one wait that gave up poisons the rest: nothing waits twice)
  for (int r = 0; r < world; ++r)
    if (!wait_for(which ? &p.words[r]->arrived : &p.words[r]->done, want, timeout)) { __hip_atomic_store(&mine->error, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
}

__global__ __launch_bounds__(256) void publish_kernel(const double* __restrict__ buf, size_t n, Peers p, int me, int world, unsigned long long seq, long long timeout) {
  __shared__ int ok;
  Words* mine = p.words[me];
  if (threadIdx.x == 0) ok = __hip_atomic_load(&mine->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0;
  __syncthreads();
  if (!ok) return;
  double* stage = const_cast<double*>(p.stage[me]) + (seq & 1) * kCap;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) stage[i] = buf[i];
  __threadfence_system();   // my part of the block is in memory before I count myself in
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(&mine->count_pub, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1 == gridDim.x) {   // the last workgroup: everything is there
      __hip_atomic_store(&mine->count_pub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&mine->arrived, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(256) void reduce_kernel(double* __restrict__ buf, size_t n, Peers p, int me, int world, unsigned long long seq, int op_max, long long timeout) {
  __shared__ int ok;
  Words* mine = p.words[me];
  if (threadIdx.x == 0) ok = __hip_atomic_load(&mine->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0;
  __syncthreads();
  if (ok) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // every wave: nothing of the peers' blocks from before their release
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      double v = p.stage[0][(seq & 1) * kCap + i];
      for (int r = 1; r < world; ++r) { const double x = p.stage[r][(seq & 1) * kCap + i]; v = op_max ? (x > v ? x : v) : v + x; }
      buf[i] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(&mine->count_red, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1 == gridDim.x) {
      __hip_atomic_store(&mine->count_red, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&mine->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (also after a wait that gave up: the others must not wait for me twice)
    }
  }
}

bool wait_host(std::atomic<int>* flags, int world, double seconds) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    bool all = true;
    for (int r = 0; r < world; ++r) all = all && flags[r].load(std::memory_order_acquire) != 0;
    if (all) return true;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
}

unsigned long long read_error(ncclComm* c) {
  Words w{};
  if (hipMemcpy(&w, c->block, sizeof w, hipMemcpyDeviceToHost) != hipSuccess) return ~0ull;
  return w.error;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* version) { if (version) *version = 99900; return ncclSuccess; }   // (no RCCL has this number: rsba_rccl_describe shows which transport ran)

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "mock rccl: a HIP call failed";
    case ncclSystemError: return "mock rccl: a rank did not arrive in time (device-side wait gave up) or the rendezvous failed";
    case ncclInvalidArgument: return "mock rccl: invalid argument (fp64, in place, sum or max only)";
    default: return "mock rccl: internal error";
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof *id);
  std::snprintf(id->internal, sizeof id->internal, "/rsba_mock_rccl_%d_%llx", (int)getpid(), (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  const bool ok = ftruncate(fd, sizeof(Board)) == 0;   // (zero-filled: nobody is ready)
  close(fd);
  return ok ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank; c->world = nranks;
  std::memcpy(c->path, id.internal, sizeof c->path); c->path[sizeof c->path - 1] = 0;
  double secs = 20.0;
  if (const char* e = std::getenv("RSBA_MOCK_RCCL_TIMEOUT_S")) secs = std::atof(e);
  c->timeout_ticks = (long long)(secs * 1e8);
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
  int fd = -1;
  for (int tries = 0; tries < 20000 && fd < 0; ++tries) { fd = shm_open(c->path, O_RDWR, 0600); if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
  if (fd < 0) { delete c; return ncclSystemError; }
  c->board = static_cast<Board*>(mmap(nullptr, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->board == MAP_FAILED) { delete c; return ncclSystemError; }
  const size_t bytes = 256 + 2 * kCap * sizeof(double);
  if (hipMalloc(&c->block, bytes) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
  if (hipMemset(c->block, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { delete c; return ncclUnhandledCudaError; }
  if (nranks > 1) {
    if (hipIpcGetMemHandle(&c->board->handle[rank], c->block) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    c->board->ready[rank].store(1, std::memory_order_release);
    if (!wait_host(c->board->ready, nranks, 120.0)) { delete c; return ncclSystemError; }
  }
  for (int r = 0; r < nranks; ++r) {
    void* base = c->block;
    if (r != rank) {
      if (hipIpcOpenMemHandle(&base, c->board->handle[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
      c->mapped[r] = base;
    }
    c->peers.words[r] = static_cast<Words*>(base);
    c->peers.stage[r] = reinterpret_cast<const double*>(static_cast<char*>(base) + 256);
  }
  c->board->opened[rank].store(1, std::memory_order_release);
  if (!wait_host(c->board->opened, nranks, 120.0)) { delete c; return ncclSystemError; }
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  const unsigned long long err = read_error(c);
  // nobody frees a block another rank may still be reading: everybody says "closing" first (the ready words count down)
  c->board->ready[c->rank].store(0, std::memory_order_release);
  for (int tries = 0; tries < 20000; ++tries) {
    bool any = false;
    for (int r = 0; r < c->world; ++r) any = any || c->board->ready[r].load(std::memory_order_acquire) != 0;
    if (!any) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  for (int r = 0; r < c->world; ++r) if (c->mapped[r]) (void)hipIpcCloseMemHandle(c->mapped[r]);
  (void)hipFree(c->block);
  munmap(c->board, sizeof(Board));
  if (c->rank == 0) shm_unlink(c->path);
  delete c;
  return err ? ncclSystemError : ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count) { if (!c || !count) return ncclInvalidArgument; *count = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* rank) { if (!c || !rank) return ncclInvalidArgument; *rank = c->rank; return ncclSuccess; }

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
  if (!c || sendbuff != recvbuff || datatype != ncclFloat64 || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
  double* buf = static_cast<double*>(recvbuff);
  for (size_t at = 0; at < count || (count == 0 && at == 0); at += kCap) {
    const size_t n = count - at < kCap ? count - at : kCap;
    const unsigned long long seq = ++c->seq;
    const unsigned grid = (unsigned)((n + 256 * 8 - 1) / (256 * 8) < 1 ? 1 : ((n + 256 * 8 - 1) / (256 * 8) > 512 ? 512 : (n + 256 * 8 - 1) / (256 * 8)));
    if (seq > 2) hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, stream, c->peers, c->rank, c->world, seq - 2, 0, seq, c->timeout_ticks);
    hipLaunchKernelGGL(publish_kernel, dim3(grid), dim3(256), 0, stream, buf + at, n, c->peers, c->rank, c->world, seq, c->timeout_ticks);
    hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, stream, c->peers, c->rank, c->world, seq, 1, seq, c->timeout_ticks);
    hipLaunchKernelGGL(reduce_kernel, dim3(grid), dim3(256), 0, stream, buf + at, n, c->peers, c->rank, c->world, seq, op == ncclMax ? 1 : 0, c->timeout_ticks);
    if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    if (count == 0) break;
  }
  // a device-side wait that gave up on an EARLIER collective is reported now (reading the word would wait for the stream: only every 256th call looks)
  if ((c->seq & 255) == 0 && hipStreamQuery(stream) == hipSuccess && read_error(c)) return ncclSystemError;
  return ncclSuccess;
}

}  // extern "C"
