// C-ABI of the RS-PnP hypothesis path (include/rsba_amd.h: rsba_pnp_tasks, rsba_pnp_inliers).  Host-side glue only:
// staging of the caller's arrays and the launches; every number comes from kernels_pnp.hip.
#include "../../include/rsba_amd.h"

#include <string>
#include <vector>

#include "handle.hpp"
#include "pnp_state.hpp"

using namespace rsba;

namespace {

struct DeviceBuffers {
  std::vector<void*> ptrs;
  ~DeviceBuffers() { for (void* p : ptrs) (void)hipFree(p); }
  template <class T> hipError_t upload(const T** out, const T* src, size_t count) {
    T* p = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) return e;
    ptrs.push_back(p);
    *out = p;
    return count ? hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice) : hipSuccess;
  }
  template <class T> hipError_t alloc(T** out, size_t count) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(out), (count ? count : 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(*out);
    return e;
  }
};

#define PNP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return rsba_set_error(e_ == hipErrorOutOfMemory ? RSBA_ERR_OUT_OF_MEMORY : RSBA_ERR_HIP, \
                                                (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
  } while (0)

int32_t select_device(int32_t device) {
  int32_t ndev = 0;
  int32_t rc = rsba_device_count(&ndev);
  if (rc) return rc;
  if (device < 0 || device >= ndev) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  PNP_TRY(hipSetDevice(device));
  return RSBA_OK;
}

int32_t fill_camera(PnpArgs& A, const double* cam, int32_t shutter, const int32_t* scanlines, float reprojection_error) {
  if (!cam || !scanlines) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  if (shutter < 0 || shutter > 2) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "shutter must be 0 (GLOBAL), 1 (HORIZONTAL) or 2 (VERTICAL)");
  if (shutter != 0 && scanlines[0] == scanlines[1]) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "scanlines[0] == scanlines[1]");
  for (int k = 0; k < 9; ++k) A.cam[k] = cam[k];
  A.shutter = shutter; A.scan0 = scanlines[0]; A.scan1 = scanlines[1];
  A.reprojection_error = reprojection_error;
  return RSBA_OK;
}

}  // namespace

extern "C" int32_t rsba_pnp_tasks(int32_t device, const double* cam, int32_t shutter, const int32_t* scanlines, const float* object_points,
                                  const float* image_points, int32_t n, const int32_t* subsets, int32_t m, int32_t num_tasks,
                                  const double* init_poses, int32_t init_stride, int32_t max_num_iterations, int32_t drop_coincident, float reprojection_error,
                                  double* poses_out, uint8_t* status, double* final_cost, int32_t* num_inliers) {
  if (!object_points || !image_points || !subsets || !init_poses || !poses_out || !status)
    return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  if (n <= 0 || m <= 0 || num_tasks <= 0 || max_num_iterations < 0 || (init_stride != 0 && init_stride != 12))
    return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "bad sizes (init_stride is 0 or 12)");
  for (size_t k = 0; k < (size_t)num_tasks * m; ++k)
    if (subsets[k] < 0 || subsets[k] >= n) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "subset index out of range");
  PnpArgs A{};
  int32_t rc = fill_camera(A, cam, shutter, scanlines, reprojection_error);
  if (rc) return rc;
  if ((rc = select_device(device))) return rc;
  A.n = n; A.m = m; A.num_tasks = num_tasks; A.init_stride = init_stride; A.max_num_iterations = max_num_iterations; A.drop_coincident = drop_coincident;
  DeviceBuffers B;
  PNP_TRY(B.upload(&A.object_points, object_points, (size_t)n * 3));
  PNP_TRY(B.upload(&A.image_points, image_points, (size_t)n * 2));
  PNP_TRY(B.upload(&A.subsets, subsets, (size_t)num_tasks * m));
  PNP_TRY(B.upload(&A.init_poses, init_poses, init_stride ? (size_t)num_tasks * 12 : 12));
  PNP_TRY(B.alloc(&A.poses_out, (size_t)num_tasks * 12));
  PNP_TRY(B.alloc(&A.status, (size_t)num_tasks));
  PNP_TRY(B.alloc(&A.final_cost, (size_t)num_tasks));
  PNP_TRY(B.alloc(&A.num_inliers, (size_t)num_tasks));
  PNP_TRY(hipMemset(A.poses_out, 0, (size_t)num_tasks * 12 * sizeof(double)));
  PNP_TRY(hipMemset(A.final_cost, 0, (size_t)num_tasks * sizeof(double)));
  PNP_TRY(hipMemset(A.num_inliers, 0, (size_t)num_tasks * sizeof(int32_t)));
  PNP_TRY(launch_pnp_tasks(A, nullptr));
  PNP_TRY(hipMemcpy(poses_out, A.poses_out, (size_t)num_tasks * 12 * sizeof(double), hipMemcpyDeviceToHost));
  PNP_TRY(hipMemcpy(status, A.status, (size_t)num_tasks, hipMemcpyDeviceToHost));
  if (final_cost) PNP_TRY(hipMemcpy(final_cost, A.final_cost, (size_t)num_tasks * sizeof(double), hipMemcpyDeviceToHost));
  if (num_inliers) PNP_TRY(hipMemcpy(num_inliers, A.num_inliers, (size_t)num_tasks * sizeof(int32_t), hipMemcpyDeviceToHost));
  return RSBA_OK;
}

extern "C" int32_t rsba_pnp_inliers(int32_t device, const double* cam, int32_t shutter, const int32_t* scanlines, const float* object_points,
                                    const float* image_points, int32_t n, const double* poses, float reprojection_error, uint8_t* inlier_mask) {
  if (!object_points || !image_points || !poses || !inlier_mask) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  if (n <= 0) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "bad sizes");
  PnpArgs A{};
  int32_t rc = fill_camera(A, cam, shutter, scanlines, reprojection_error);
  if (rc) return rc;
  if ((rc = select_device(device))) return rc;
  A.n = n;
  DeviceBuffers B;
  const double* d_poses = nullptr; uint8_t* d_mask = nullptr;
  PNP_TRY(B.upload(&A.object_points, object_points, (size_t)n * 3));
  PNP_TRY(B.upload(&A.image_points, image_points, (size_t)n * 2));
  PNP_TRY(B.upload(&d_poses, poses, 12));
  PNP_TRY(B.alloc(&d_mask, (size_t)n));
  PNP_TRY(launch_pnp_inliers(A, d_poses, d_mask, nullptr));
  PNP_TRY(hipMemcpy(inlier_mask, d_mask, (size_t)n, hipMemcpyDeviceToHost));
  return RSBA_OK;
}
