// Does a kernel that starts AFTER another stream's writes (ordered by an event) ever read a stale line out of its XCD's L2?
//
// cholesky.hip's look-ahead loads (Frag::load, times_inverse_transposed: round 5 / 6) read the write-once cells with ordinary cached loads and
// fall back on a coherent read only where a double still looks empty.  A FILLED value that is stale — left in some XCD's L2 by the solve
// that used the same set of cells two solves ago, while the cells were re-armed by a memset on another stream — would be taken for this
// solve's.  What rules it out is that every kernel dispatch begins with an acquire that invalidates its XCDs' L2 lines of memory another
// agent / XCD may have written; nothing in the library can enforce that, so this probe watches it ("nothing enforces or tests
// [it] directly"):
//   round r:  (stream A) every workgroup of a chip-filling launch reads all lines with plain loads and checks them against r  -> the
//             lines now sit in all eight L2s;   (stream B, after an event) the lines are rewritten with r + 1 — by a memset-like kernel
//             of ONE workgroup (one XCD), or by hipMemsetD32Async —;   (stream A, after an event) the next check expects r + 1.
// Reports the number of stale reads over all rounds (expected: 0) and, as a control, the same with the checker's loads made coherent.
//   hipcc -O3 --offload-arch=gfx950 tools/l2_boundary_probe.hip -o tools/l2_boundary_probe && tools/l2_boundary_probe [rounds] [lines]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kStride = 16;   // doubles: one 128-byte line per cell

__global__ void write_kernel(double* buf, int lines, double v) {
  for (int i = threadIdx.x; i < lines; i += blockDim.x) buf[(size_t)i * kStride] = v;
}
template <bool COHERENT>
__global__ void check_kernel(const double* buf, int lines, double want, unsigned long long* stale, unsigned* xcd_seen) {
  unsigned long long bad = 0;
  for (int i = threadIdx.x; i < lines; i += blockDim.x) {
    const double* p = buf + (size_t)i * kStride;
    const double v = COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *(const __attribute__((address_space(1))) double*)p;   // (NOT volatile: a volatile load is emitted system-coherent, sc0 sc1)
    bad += v != want;
  }
  if (bad) atomicAdd(stale, bad);
  if (threadIdx.x == 0) atomicOr(xcd_seen, 1u << (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15));   // XCC_ID
}

// Positive control — the probe must be able to SEE a stale line: inside ONE launch nothing invalidates.  Every workgroup reads the lines (old
// value, now in its XCD's L2), all meet at a counter, workgroup 0 rewrites them (write-through + fence) and raises a flag; the others then
// read again — plain loads (expected: the old value out of their own L2, unless they share workgroup 0's XCD) and coherent loads (expected: new).
__global__ void inlaunch_kernel(double* buf, int lines, double old_v, double new_v, unsigned* sync, unsigned long long* out) {
  unsigned long long bad = 0;
  for (int i = threadIdx.x; i < lines; i += blockDim.x) bad += buf[(size_t)i * kStride] != old_v;
  if (bad) atomicAdd(out + 0, bad);                       // (first look: everyone must see the old value)
  __syncthreads();
  if (threadIdx.x == 0) { atomicAdd(sync, 1u); while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(8); }
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < lines; i += blockDim.x) __hip_atomic_store(buf + (size_t)i * kStride, new_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (threadIdx.x == 0) while (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  unsigned long long stale_plain = 0, stale_coh = 0;
  for (int i = threadIdx.x; i < lines; i += blockDim.x) {
    const double* p = buf + (size_t)i * kStride;
    stale_plain += *(const __attribute__((address_space(1))) double*)p != new_v;
    stale_coh += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != new_v;
  }
  if (stale_plain) atomicAdd(out + 1, stale_plain);
  if (stale_coh) atomicAdd(out + 2, stale_coh);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 2000, lines = argc > 2 ? std::atoi(argv[2]) : 8192;
  double* buf; unsigned long long* stale; unsigned* seen;
  CHECK(hipMalloc(&buf, (size_t)lines * kStride * sizeof(double)));
  CHECK(hipMalloc(&stale, 2 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&seen, sizeof(unsigned)));
  CHECK(hipMemset(stale, 0, 2 * sizeof(unsigned long long)));
  CHECK(hipMemset(seen, 0, sizeof(unsigned)));
  hipStream_t A, B; hipEvent_t ea, eb;
  CHECK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  CHECK(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  for (int mode = 0; mode < 3; ++mode) {   // 0: rewritten by a one-workgroup kernel, plain checks; 1: by hipMemsetD32Async (value = repeated 32-bit pattern), plain checks; 2: control, coherent checks
    CHECK(hipMemset(stale, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(write_kernel, dim3(1), dim3(256), 0, A, buf, lines, 0.0);
    for (int r = 0; r < rounds; ++r) {
      double want, next;
      if (mode == 1) { const unsigned pat = 0x3f000000u + (unsigned)r, pn = pat + 1; unsigned long long w = ((unsigned long long)pat << 32) | pat, n = ((unsigned long long)pn << 32) | pn; want = *reinterpret_cast<double*>(&w); next = *reinterpret_cast<double*>(&n); if (r == 0) { CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(buf), (int)pat, (size_t)lines * kStride * 2, A)); } }
      else { want = (double)r; next = (double)(r + 1); }
      if (mode == 2) hipLaunchKernelGGL(check_kernel<true>, dim3(2048), dim3(256), 0, A, buf, lines, want, stale, seen);
      else hipLaunchKernelGGL(check_kernel<false>, dim3(2048), dim3(256), 0, A, buf, lines, want, stale, seen);
      CHECK(hipEventRecord(ea, A));
      CHECK(hipStreamWaitEvent(B, ea, 0));
      if (mode == 1) { const unsigned pn = 0x3f000000u + (unsigned)r + 1; CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(buf), (int)pn, (size_t)lines * kStride * 2, B)); }
      else hipLaunchKernelGGL(write_kernel, dim3(1), dim3(256), 0, B, buf, lines, next);
      CHECK(hipEventRecord(eb, B));
      CHECK(hipStreamWaitEvent(A, eb, 0));
    }
    CHECK(hipStreamSynchronize(A)); CHECK(hipStreamSynchronize(B));
    unsigned long long bad = 0; unsigned xs = 0;
    CHECK(hipMemcpy(&bad, stale, sizeof bad, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&xs, seen, sizeof xs, hipMemcpyDeviceToHost));
    std::printf("%-62s rounds %d, lines %d (x 2048 workgroups per check, XCDs seen: mask 0x%x): stale reads %llu\n",
                mode == 0 ? "rewritten by a one-workgroup kernel on another stream, plain loads" : mode == 1 ? "rewritten by hipMemsetD32Async on another stream, plain loads" : "control: the same with coherent (agent-scope) loads",
                rounds, lines, xs, bad);
  }
  {   // positive control: 64 workgroups (eight per XCD), one launch
    unsigned* sync; unsigned long long* out;
    CHECK(hipMalloc(&sync, 2 * sizeof(unsigned))); CHECK(hipMalloc(&out, 3 * sizeof(unsigned long long)));
    unsigned long long tot[3] = {0, 0, 0};
    const int reps = 50, wgs = 64;
    for (int r = 0; r < reps; ++r) {
      CHECK(hipMemsetAsync(sync, 0, 2 * sizeof(unsigned), A)); CHECK(hipMemsetAsync(out, 0, 3 * sizeof(unsigned long long), A));
      hipLaunchKernelGGL(write_kernel, dim3(1), dim3(256), 0, A, buf, lines, (double)(2 * r));
      hipLaunchKernelGGL(inlaunch_kernel, dim3(wgs), dim3(256), 0, A, buf, lines, (double)(2 * r), (double)(2 * r + 1), sync, out);
      unsigned long long o[3];
      CHECK(hipMemcpyAsync(o, out, sizeof o, hipMemcpyDeviceToHost, A)); CHECK(hipStreamSynchronize(A));
      for (int k = 0; k < 3; ++k) tot[k] += o[k];
    }
    std::printf("positive control, ONE launch (%d workgroups, workgroup 0 rewrites %d lines after all have read them; %d repetitions):\n"
                "  first look not the old value %llu;  second look, plain loads: STALE %llu of %llu;  second look, coherent loads: stale %llu\n",
                wgs, lines, reps, tot[0], tot[1], (unsigned long long)reps * (wgs - 1) * lines, tot[2]);
  }
  return 0;
}
