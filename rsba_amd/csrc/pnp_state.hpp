// Arguments of the RS-PnP hypothesis kernels (kernels_pnp.hip); all pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rsba {

struct PnpArgs {
  double cam[9];                 // {fx,fy,k1,k2,p1,p2,k3,cx,cy}
  int shutter, scan0, scan1;
  int n, m, num_tasks, init_stride, max_num_iterations, drop_coincident;
  float reprojection_error;
  const float* object_points;    // [n][3]
  const float* image_points;     // [n][2]
  const int32_t* subsets;        // [num_tasks][m]
  const double* init_poses;      // [num_tasks][12] (init_stride 12) or [12] shared (init_stride 0)
  double* poses_out;             // [num_tasks][12]
  uint8_t* status;               // [num_tasks] 0 skipped, 1 solved and usable, 2 solve failed (initial poses kept)
  double* final_cost;            // [num_tasks]
  int32_t* num_inliers;          // [num_tasks]
};

hipError_t launch_pnp_tasks(const PnpArgs& args, hipStream_t st);
hipError_t launch_pnp_inliers(const PnpArgs& args, const double* poses, uint8_t* mask, hipStream_t st);

}  // namespace rsba
