// C-ABI of the stateless per-frame reprojection / validation filter (include/rsba_amd.h: rsba_validate_frame,
// rsba_reproject_frame) — what vision::sfm::validate / reproject (struct/VideoSfM.cc:139-169) are called for, one frame
// at a time, by CeresHandler::Add (CeresHandler.h:220-243) and the track bookkeeping of VideoSfMHandler.cc.  No handle:
// each host thread keeps one growing device arena + pinned staging buffer + stream, so a call costs one upload, one
// launch and one download (the first round built a whole rsba_handle per call).  Kernels: kernels_filter.hip.
#include "../../include/rsba_amd.h"

#include <algorithm>
#include <cstring>
#include <string>

#include "handle.hpp"

using namespace rsba;

namespace rsba {
hipError_t launch_iota(int32_t* zeros, int32_t* iota, int64_t first, int64_t n, hipStream_t st);   // kernels_filter.hip
}

namespace {

#define FILTER_TRY(expr)                                                                           \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return rsba_set_error(e_ == hipErrorOutOfMemory ? RSBA_ERR_OUT_OF_MEMORY : RSBA_ERR_HIP, \
                                                (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
  } while (0)

// per host thread and device: [cam 9 | poses 12 | points 3n | xy 2n | poses of a frame with more than two: 6 np] doubles in,
// [xy 2n doubles | flags n bytes] out
struct Arena {
  int device = -1;
  hipStream_t stream = nullptr;
  int64_t cap = 0, pcap = 0;       // items the buffers hold; poses of a pose-per-scan-line frame behind them
  double* d_in = nullptr; double* d_xy = nullptr; uint8_t* d_flags = nullptr; int32_t *d_zero = nullptr, *d_iota = nullptr;
  double* h_in = nullptr; double* h_xy = nullptr; uint8_t* h_flags = nullptr;   // pinned
  void release() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    (void)hipFree(d_in); (void)hipFree(d_xy); (void)hipFree(d_flags); (void)hipFree(d_zero); (void)hipFree(d_iota);
    (void)hipHostFree(h_in); (void)hipHostFree(h_xy); (void)hipHostFree(h_flags);
    d_in = d_xy = nullptr; d_flags = nullptr; d_zero = d_iota = nullptr; h_in = h_xy = nullptr; h_flags = nullptr; cap = 0; pcap = 0;
  }
  // (thread_local: destroyed when its thread ends — for the main thread before the static destructors of libamdhip64, which this
  // library depends on and which was therefore loaded, and registered its teardown, first; threads still alive at exit never run
  // it.  Errors of the frees are ignored either way.)
  ~Arena() { release(); if (stream) (void)hipStreamDestroy(stream); }
  int32_t reserve(int dev, int64_t n, int64_t np) {
    if (dev != device) { release(); if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; } device = dev; }
    FILTER_TRY(hipSetDevice(dev));
    if (!stream) FILTER_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (n <= cap && np <= pcap) return RSBA_OK;
    const int64_t c0 = cap, p0 = pcap;
    release(); device = dev;
    int64_t c = std::max<int64_t>(c0, 1024); while (c < n) c *= 2;
    int64_t pc = std::max<int64_t>(p0, 2); while (pc < np) pc *= 2;
    FILTER_TRY(hipMalloc(reinterpret_cast<void**>(&d_in), (21 + 5 * (size_t)c + 6 * (size_t)pc) * sizeof(double)));
    FILTER_TRY(hipMalloc(reinterpret_cast<void**>(&d_xy), 2 * (size_t)c * sizeof(double)));
    FILTER_TRY(hipMalloc(reinterpret_cast<void**>(&d_flags), (size_t)c));
    FILTER_TRY(hipMalloc(reinterpret_cast<void**>(&d_zero), (size_t)c * sizeof(int32_t)));
    FILTER_TRY(hipMalloc(reinterpret_cast<void**>(&d_iota), (size_t)c * sizeof(int32_t)));
    FILTER_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_in), (21 + 5 * (size_t)c + 6 * (size_t)pc) * sizeof(double)));
    FILTER_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_xy), 2 * (size_t)c * sizeof(double)));
    FILTER_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_flags), (size_t)c));
    FILTER_TRY(launch_iota(d_zero, d_iota, 0, c, stream));
    cap = c; pcap = pc;
    return RSBA_OK;
  }
};
thread_local Arena g_arena;

int32_t stage(Arena& A, DeviceProblem& dp, int32_t device, const double* cam, const double* poses, int32_t num_poses, int32_t shutter, const int32_t* scanlines,
              int32_t interpolate_rotation, const double* points, const double* obs_xy, int64_t n) {
  if (!cam || !poses || !scanlines || !points) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  if (num_poses < 1) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "empty frame");   // (getPose throws, struct/VideoSfM.cc:105)
  if (shutter < 0 || shutter > 2) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "shutter");
  if (shutter != RSBA_SHUTTER_GLOBAL && num_poses == 2 && scanlines[0] == scanlines[1]) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "scanlines[0] == scanlines[1]");
  int32_t ndev = 0;
  int32_t rc = rsba_device_count(&ndev);
  if (rc) return rc;
  if (device < 0 || device >= ndev) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  if ((rc = A.reserve(device, n, num_poses))) return rc;
  // one or two poses sit in front; the poses of a frame with more ("fullDoF": one per scan line, picked per item by the kernels as
  // getPose does, struct/VideoSfM.cc:118-132) behind the items, still ONE upload
  const size_t items = (21 + (obs_xy ? 5 : 3) * (size_t)n), pose_off = num_poses <= 2 ? 9 : items;
  std::memcpy(A.h_in, cam, 9 * sizeof(double));
  std::memcpy(A.h_in + pose_off, poses, 6 * (size_t)num_poses * sizeof(double));
  std::memcpy(A.h_in + 21, points, 3 * (size_t)n * sizeof(double));
  if (obs_xy) std::memcpy(A.h_in + 21 + 3 * (size_t)n, obs_xy, 2 * (size_t)n * sizeof(double));
  FILTER_TRY(hipMemcpyAsync(A.d_in, A.h_in, (items + (num_poses <= 2 ? 0 : 6 * (size_t)num_poses)) * sizeof(double), hipMemcpyHostToDevice, A.stream));
  std::memset(&dp, 0, sizeof dp);
  dp.pp_spherical = -1;   // "no SphericalPrior" is -1, not 0 (only the filter kernels see this dp, but a zero would name pose block 0)
  dp.shutter = shutter; dp.scan0 = scanlines[0]; dp.scan1 = scanlines[1]; dp.interp_rotation = interpolate_rotation != 0; dp.calibrated = 1; dp.P = num_poses;
  dp.F = 1; dp.M = (int)n; dp.NI = 1; dp.N = n;
  dp.intr = A.d_in; dp.poses = A.d_in + pose_off; dp.points = A.d_in + 21;
  dp.xy = reinterpret_cast<const double2*>(A.d_in + 21 + 3 * (size_t)n);
  dp.obs_frame = A.d_zero; dp.obs_point = A.d_iota; dp.frame_intr = A.d_zero;
  return RSBA_OK;
}

}  // namespace

extern "C" int32_t rsba_validate_frame(int32_t device, const double* cam, const double* poses, int32_t num_poses, int32_t shutter, const int32_t* scanlines,
                                       int32_t interpolate_rotation, const double* points, const double* obs_xy, int64_t n, double sq_threshold,
                                       double min_distance, uint8_t* valid) {
  if (n <= 0) return RSBA_OK;
  if (n > 0x7fffffff) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "too many items for one frame");
  if (!obs_xy || !valid) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  Arena& A = g_arena; DeviceProblem dp;
  int32_t rc = stage(A, dp, device, cam, poses, num_poses, shutter, scanlines, interpolate_rotation, points, obs_xy, n);
  if (rc) return rc;
  FILTER_TRY(launch_validate(dp, sq_threshold, min_distance, A.d_flags, A.stream));
  FILTER_TRY(hipMemcpyAsync(A.h_flags, A.d_flags, (size_t)n, hipMemcpyDeviceToHost, A.stream));
  FILTER_TRY(hipStreamSynchronize(A.stream));
  std::memcpy(valid, A.h_flags, (size_t)n);
  return RSBA_OK;
}

extern "C" int32_t rsba_reproject_frame(int32_t device, const double* cam, const double* poses, int32_t num_poses, int32_t shutter, const int32_t* scanlines,
                                        int32_t interpolate_rotation, const double* points, int64_t n, double* xy_out, uint8_t* ok_out) {
  if (n <= 0) return RSBA_OK;
  if (n > 0x7fffffff) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "too many items for one frame");
  if (!xy_out || !ok_out) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  Arena& A = g_arena; DeviceProblem dp;
  int32_t rc = stage(A, dp, device, cam, poses, num_poses, shutter, scanlines, interpolate_rotation, points, nullptr, n);
  if (rc) return rc;
  FILTER_TRY(launch_reproject(dp, A.d_zero, A.d_iota, n, A.d_xy, A.d_flags, A.stream));
  FILTER_TRY(hipMemcpyAsync(A.h_xy, A.d_xy, 2 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, A.stream));
  FILTER_TRY(hipMemcpyAsync(A.h_flags, A.d_flags, (size_t)n, hipMemcpyDeviceToHost, A.stream));
  FILTER_TRY(hipStreamSynchronize(A.stream));
  std::memcpy(xy_out, A.h_xy, 2 * (size_t)n * sizeof(double));
  std::memcpy(ok_out, A.h_flags, (size_t)n);
  return RSBA_OK;
}
