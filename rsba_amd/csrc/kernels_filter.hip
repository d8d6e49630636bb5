// SURVEY §8f row f2 — the steps either side of the solve: batched reprojection / validation filter.
//   vision::sfm::getPose (copying overload)   /root/reference/src/rsba/struct/VideoSfM.cc:103-133
//   vision::sfm::reproject                    /root/reference/src/rsba/struct/VideoSfM.cc:139-155
//   vision::sfm::validate(sess,f,opt,pt,obs)  /root/reference/src/rsba/struct/VideoSfM.cc:159-169
// used by createTracks / evalTracks (VideoSfMHandler.cc:231-410) and CeresHandler::Add's
// revalidateReprojections (CeresHandler.h:239-243).  Same camera math as the residual kernel, but — unlike
// the cost functor (VideoSfmBaRs.h:31) — tau comes from the TRUE observation: x for HORIZONTAL, y for VERTICAL.
// One lane per observation / per (frame, point) pair; embarrassingly parallel, HBM-bound (24-40 B per item).
#include "device_state.hpp"
#include "obs_math.hpp"

namespace rsba {

namespace {

// interpolate_rs with the true observation (mat/cam.h:315-349): pose at the observation's scan line
// P == 0: a frame with MORE than two poses — one per scan line ("fullDoF"): the pose whose index is the rounded, clamped scan
// line of the observation (struct/VideoSfM.cc:118-132: x for HORIZONTAL, y otherwise — a GLOBAL session included; std::round,
// halves away from zero); np = f.poses.size()
template <int P>
__device__ __forceinline__ void pose_at(const Model& m, const double* __restrict__ poses, double ox, double oy, double out[6], int np = P) {
  if (P == 0) {
    double line = (m.shutter == kHorizontal) ? ox : oy;
    if (line < 0.0) line = 0.0; else if (line > double(np - 1)) line = double(np - 1);
    const double* q = poses + 6 * (size_t)round(line);
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] = q[k];
    return;
  }
  if (P == 1 || m.shutter == kGlobal) {
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] = poses[k];
    return;
  }
  const double coord = (m.shutter == kVertical) ? oy : ox;            // cam.h:325-331
  double tau = (coord - double(m.scan0)) / double(m.scan1 - m.scan0);
  if (tau < 0.0) tau = 0.0;
  if (tau > 1.0) tau = 1.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k] = (k < 3 && !m.interp_rotation) ? poses[k] : poses[k] + (poses[6 + k] - poses[k]) * tau;
}

// w2i(cam, pose, X, proj, validate = true) (mat/cam.h:400-419) through the shared single-pose evaluation
__device__ __forceinline__ bool project(const double* cam, const double pose[6], const double X[3], double proj[2]) {
  const Model gs = {kGlobal, 0, 1, 1};
  ObsOut<true, 1> o;
  eval_observation<true, 1, false>(gs, cam, pose, X, 0.0, 0.0, o);     // residual against (0,0) = the projection
  proj[0] = o.r[0]; proj[1] = o.r[1];
  return o.ok;
}

template <int P>
__global__ __launch_bounds__(256) void validate_kernel(const DeviceProblem dp, double sq_threshold, double min_distance, uint8_t* valid) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= dp.N) return;
  const double2 xy = dp.xy[i];
  const int f = dp.obs_frame[i], j = dp.obs_point[i];
  const Model m = {(dp.frame_global && dp.frame_global[f]) ? (int)kGlobal : dp.shutter, dp.scan0, dp.scan1, dp.interp_rotation};   // (a one-pose frame: getPose returns poses[0], VideoSfM.cc:103-133)
  const double* cam = dp.intr + (size_t)((dp.NI == 1) ? 0 : dp.frame_intr[f]) * 9;
  double camr[9], pose[6], X[3], proj[2];
#pragma unroll
  for (int k = 0; k < 9; ++k) camr[k] = cam[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) X[k] = dp.points[(size_t)j * 3 + k];
  pose_at<P>(m, dp.poses + (size_t)f * 6 * dp.P, xy.x, xy.y, pose, dp.P);
  const double dx = pose[3] - X[0], dy = pose[4] - X[1], dz = pose[5] - X[2];
  const bool far_enough = sqrt(dx * dx + dy * dy + dz * dz) >= min_distance;          // VideoSfM.cc:164-167
  const bool ok = project(camr, pose, X, proj);
  const double ex = proj[0] - xy.x, ey = proj[1] - xy.y;
  valid[i] = (far_enough && ok && (ex * ex + ey * ey) < sq_threshold) ? 1 : 0;        // cam.h:444-457
}

template <int P>
__global__ __launch_bounds__(256) void reproject_kernel(const DeviceProblem dp, const int32_t* frames, const int32_t* points, int64_t n,
                                                        double2* xy_out, uint8_t* ok_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int f = frames[i], j = points[i];
  const Model m = {(dp.frame_global && dp.frame_global[f]) ? (int)kGlobal : dp.shutter, dp.scan0, dp.scan1, dp.interp_rotation};
  const double* cam = dp.intr + (size_t)((dp.NI == 1) ? 0 : dp.frame_intr[f]) * 9;
  double camr[9], pose[6], X[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) camr[k] = cam[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) X[k] = dp.points[(size_t)j * 3 + k];
  double proj[2] = {camr[7], camr[8]};                                   // start at the principal point (:142)
  bool ok = true;
  int limit = 50;
  for (;;) {
    if (--limit < 1) { ok = false; break; }                              // :146
    const double px = proj[0], py = proj[1];
    pose_at<P>(m, dp.poses + (size_t)f * 6 * dp.P, px, py, pose, dp.P);   // :148
    if (!project(camr, pose, X, proj)) { ok = false; break; }            // :149
    const double mx = px - proj[0], my = py - proj[1];
    if (!(P != 1 && mx * mx + my * my > 1e-6)) break;                     // :151 (f.poses.size() > 1)
  }
  xy_out[i] = make_double2(proj[0], proj[1]);
  ok_out[i] = ok ? 1 : 0;   // the closing validate (:154) compares the projection with itself: true whenever w2i succeeded
}

__global__ void iota_kernel(int32_t* zeros, int32_t* iota, int64_t first, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { zeros[i] = 0; iota[i] = (int32_t)(first + i); }
}
// caller order: out[order[i]] = in[i]
__global__ void scatter_flags_kernel(const uint8_t* in, const int64_t* order, int64_t n, uint8_t* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[order[i]] = in[i];
}

}  // namespace

hipError_t launch_iota(int32_t* zeros, int32_t* iota, int64_t first, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, zeros, iota, first, n);
  return hipGetLastError();
}
hipError_t launch_scatter_flags(const uint8_t* in, const int64_t* order, int64_t n, uint8_t* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, order, n, out);
  return hipGetLastError();
}

hipError_t launch_validate(const DeviceProblem& dp, double sq_threshold, double min_distance, uint8_t* valid, hipStream_t st) {
  if (dp.N <= 0) return hipSuccess;
  const int grid = (int)((dp.N + 255) / 256);
  if (dp.P > 2) hipLaunchKernelGGL(validate_kernel<0>, dim3(grid), dim3(256), 0, st, dp, sq_threshold, min_distance, valid);   // (stateless per-frame calls only: a handle has 1 or 2)
  else if (dp.P == 2) hipLaunchKernelGGL(validate_kernel<2>, dim3(grid), dim3(256), 0, st, dp, sq_threshold, min_distance, valid);
  else hipLaunchKernelGGL(validate_kernel<1>, dim3(grid), dim3(256), 0, st, dp, sq_threshold, min_distance, valid);
  return hipGetLastError();
}
hipError_t launch_reproject(const DeviceProblem& dp, const int32_t* frames, const int32_t* points, int64_t n, double* xy_out, uint8_t* ok_out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const int grid = (int)((n + 255) / 256);
  if (dp.P > 2) hipLaunchKernelGGL(reproject_kernel<0>, dim3(grid), dim3(256), 0, st, dp, frames, points, n, reinterpret_cast<double2*>(xy_out), ok_out);
  else if (dp.P == 2) hipLaunchKernelGGL(reproject_kernel<2>, dim3(grid), dim3(256), 0, st, dp, frames, points, n, reinterpret_cast<double2*>(xy_out), ok_out);
  else hipLaunchKernelGGL(reproject_kernel<1>, dim3(grid), dim3(256), 0, st, dp, frames, points, n, reinterpret_cast<double2*>(xy_out), ok_out);
  return hipGetLastError();
}

}  // namespace rsba
