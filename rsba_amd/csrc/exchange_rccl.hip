// Native multi-GPU transport of the LM exchange: RCCL (ncclAllReduce over xGMI) called straight from the solver's
// stream — no callback into the host language, nothing between the kernels that fill the buffer and the collective
// but stream order.  The reference is single-process; this serves the exchange step SURVEY §8e derives for the path
// (per-camera J^T J / J^T r blocks summed over the ranks every LM iteration).
//
// librccl is looked up at run time (RSBA_RCCL_LIB, an RCCL already present in the process, librccl.so.1), so that a
// single-GPU user of librsba_amd.so does not load it at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// The library is resolved with dlopen at run time; a build host without the RCCL headers only needs the handful of
// declarations used below (NCCL's stable C API: nccl.h of NCCL 2.x / RCCL).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGetVersion(int* version);
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "handle.hpp"

namespace {

struct Rccl {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;      // (optional: only rsba_rccl_describe asks)
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  std::string error;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = nullptr;
    if (const char* path = std::getenv("RSBA_RCCL_LIB")) {
      lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
      if (!lib) { r.error = std::string("RSBA_RCCL_LIB: ") + dlerror(); return; }
    } else if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
      lib = RTLD_DEFAULT;   // the process already carries an RCCL in its global scope
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
      if (!lib) { r.error = std::string("librccl not found: ") + dlerror(); return; }
    }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(lib, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(lib, "ncclGetVersion"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(lib, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
    if (!r.ok) r.error = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
  });
  return r;
}

int32_t rccl_fail(const char* what, ncclResult_t e) {
  return rsba_set_error(RSBA_ERR_COMM, (std::string(what) + ": " + rccl().GetErrorString(e)).c_str());
}

// rsba_allreduce_fn over a communicator: in place, fp64, ordered on the solver's stream
int32_t rccl_allreduce(void* ctx, double* buf, int64_t count, int32_t op, void* stream) {
  const ncclResult_t e = rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, op == 0 ? ncclSum : ncclMax, static_cast<ncclComm_t>(ctx), static_cast<hipStream_t>(stream));
  if (e != ncclSuccess) { rccl_fail("ncclAllReduce", e); return 1; }
  return 0;
}

}  // namespace

extern "C" {

int32_t rsba_rccl_get_unique_id(void* id) {
  if (!id) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  Rccl& r = rccl();
  if (!r.ok) return rsba_set_error(RSBA_ERR_COMM, r.error.c_str());
  static_assert(sizeof(ncclUniqueId) == RSBA_RCCL_UNIQUE_ID_BYTES, "unique id size");
  const ncclResult_t e = r.GetUniqueId(static_cast<ncclUniqueId*>(id));
  return e == ncclSuccess ? RSBA_OK : rccl_fail("ncclGetUniqueId", e);
}

int32_t rsba_rccl_comm_create(const void* id, int32_t rank, int32_t world, int32_t device, void** comm_out) {
  if (!id || !comm_out || world < 1 || rank < 0 || rank >= world) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "bad communicator arguments");
  *comm_out = nullptr;
  Rccl& r = rccl();
  if (!r.ok) return rsba_set_error(RSBA_ERR_COMM, r.error.c_str());
  const hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) return rsba_set_error(RSBA_ERR_HIP, hipGetErrorString(he));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  ncclComm_t comm = nullptr;
  const ncclResult_t e = r.CommInitRank(&comm, world, uid, rank);
  if (e != ncclSuccess) return rccl_fail("ncclCommInitRank", e);
  *comm_out = comm;
  return RSBA_OK;
}

void rsba_rccl_comm_destroy(void* comm) {
  if (comm && rccl().ok) (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm));
}

int32_t rsba_rccl_describe(void* nccl_comm, int32_t* version, int32_t* nranks, int32_t* rank) {
  Rccl& r = rccl();
  if (!r.ok) return rsba_set_error(RSBA_ERR_COMM, r.error.c_str());
  int v = -1, n = -1, k = -1;
  if (r.GetVersion && r.GetVersion(&v) != ncclSuccess) v = -1;
  if (nccl_comm && r.CommCount && r.CommCount(static_cast<ncclComm_t>(nccl_comm), &n) != ncclSuccess) n = -1;
  if (nccl_comm && r.CommUserRank && r.CommUserRank(static_cast<ncclComm_t>(nccl_comm), &k) != ncclSuccess) k = -1;
  if (version) *version = v;
  if (nranks) *nranks = n;
  if (rank) *rank = k;
  return RSBA_OK;
}

int32_t rsba_set_exchange_rccl(rsba_handle* h, void* nccl_comm, int32_t rank, int32_t world) {
  if (!h || !nccl_comm) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  Rccl& r = rccl();
  if (!r.ok) return rsba_set_error(RSBA_ERR_COMM, r.error.c_str());
  return rsba_set_exchange(h, rccl_allreduce, nccl_comm, rank, world);
}

}  // extern "C"
