// RS-PnP RANSAC hypotheses: what solveRsPnPRansac's pnpTask does for ONE random subset
// (/root/reference/src/rsba/solveRSpnp.cpp:265-335), for thousands of subsets at once:
//   * drop the subset if two of its 3-D points coincide (:283-293),
//   * solveRsPnP (:100-192): ceres::Solve over the two pose blocks of one rolling-shutter frame with one
//     RsBA<float> residual block per point (:25-97: observations / points stored as float, w2i WITHOUT validation),
//     max_num_iterations = 10, every other option at Ceres' defaults; the poses are kept if the solution is usable,
//   * count the inliers among ALL points (:225-258 project3dPoints, :304-310) — its own kernel, one lane per
//     (hypothesis, point).
// One lane per hypothesis runs the whole trust-region loop (the same rules as solver.hip's rsba_solve — Ceres 1.9
// TrustRegionMinimizer + LevenbergMarquardtStrategy, SURVEY Appendix C.5 — on a dense 12 x 12 system): the problems are
// independent and tiny (m x 2 residuals, 12 unknowns), so there is nothing to tile; the 78-entry normal matrix and
// its Cholesky factor of each lane live in LDS (lane-interleaved: conflict-free, dynamically indexable), vectors in
// registers.  The residual / Jacobian arithmetic is obs_math.hpp's, the same as the BA path.
#include "obs_math.hpp"
#include "pnp_state.hpp"

namespace rsba {

namespace {

constexpr int kPnpBlock = 64;                 // one wave per workgroup: 78 x 64 doubles of LDS (the normal matrix; its factor is in registers)
constexpr int kTri = 78;                      // lower triangle of 12 x 12
__device__ __forceinline__ constexpr int tri(int a, int b) { return a * (a + 1) / 2 + b; }   // a >= b

// residuals (and, WANT_J, H = J^T J lower + g = J^T r, unscaled) of the subset at poses x; returns the cost
template <bool WANT_J>
__device__ __forceinline__ double pnp_linearize(const PnpArgs& A, const Model& mdl, const int32_t* sub, const double x[12],
                                                double* __restrict__ H, double g[12], int lane) {
  if (WANT_J) {
    for (int k = 0; k < kTri; ++k) H[k * kPnpBlock + lane] = 0.0;
#pragma unroll
    for (int a = 0; a < 12; ++a) g[a] = 0.0;
  }
  double cost = 0.0;
  for (int i = 0; i < A.m; ++i) {
    const int idx = sub[i];
    const double X[3] = {(double)A.object_points[3 * idx], (double)A.object_points[3 * idx + 1], (double)A.object_points[3 * idx + 2]};
    const double ox = (double)A.image_points[2 * idx], oy = (double)A.image_points[2 * idx + 1];
    ObsOut<true, 2> o;
    eval_observation<true, 2, WANT_J>(mdl, A.cam, x, X, ox, oy, o);   // w2i(..., validate = false): o.ok is not consulted
    cost += o.r[0] * o.r[0] + o.r[1] * o.r[1];
    if (WANT_J) {
#pragma unroll
      for (int a = 0; a < 12; ++a) {
        g[a] += o.J[0][a] * o.r[0] + o.J[1][a] * o.r[1];
#pragma unroll
        for (int b = 0; b <= a; ++b) H[tri(a, b) * kPnpBlock + lane] += o.J[0][a] * o.J[0][b] + o.J[1][a] * o.J[1][b];
      }
    }
  }
  return 0.5 * cost;
}

__global__ __launch_bounds__(kPnpBlock) void pnp_tasks_kernel(const PnpArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* H = smem;                              // [78][64]
  const int lane = threadIdx.x, h = blockIdx.x * kPnpBlock + lane;
  if (h >= A.num_tasks) return;
  const int32_t* sub = A.subsets + (size_t)h * A.m;
  const Model mdl{A.shutter, A.scan0, A.scan1, 1};
  // coincident 3-D points (float differences, norm in double)
  if (A.drop_coincident) for (int i = 0; i < A.m; ++i) for (int j = i + 1; j < A.m; ++j) {
    const float* a = A.object_points + 3 * (size_t)sub[i]; const float* b = A.object_points + 3 * (size_t)sub[j];
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    if (sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz) < 1e-10) { A.status[h] = 0; return; }
  }
  double x[12], x0[12];
#pragma unroll
  for (int a = 0; a < 12; ++a) { x0[a] = A.init_poses[(size_t)h * A.init_stride + a]; x[a] = x0[a]; }

  // ---- ceres::Solve, defaults ----
  const double ftol = 1e-6, gtol = 1e-10, ptol = 1e-8, min_rel = 1e-3, lo = 1e-6, hi = 1e32, max_radius = 1e16, min_radius = 1e-32;
  double g[12], sc[12], diag[12], y[12], xn[12];
  double cost = pnp_linearize<true>(A, mdl, sub, x, H, g, lane);
  double final_cost = cost;
  int term = 1;   // 0 convergence, 1 no convergence, 2 failure
  double gmax = 0.0;
#pragma unroll
  for (int a = 0; a < 12; ++a) gmax = fmax(gmax, fabs(g[a]));
  bool done = false;
  if (!isfinite(cost)) { term = 2; done = true; }
  else if (gmax <= gtol) { term = 0; done = true; }
#pragma unroll
  for (int a = 0; a < 12; ++a) sc[a] = 1.0 / (1.0 + sqrt(H[tri(a, a) * kPnpBlock + lane]));   // EstimateScale, once
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int invalid_streak = 0;
  for (int iteration = 0; !done; ) {
    if (iteration >= A.max_num_iterations) { term = 1; break; }
    if (!reuse_diagonal) {
#pragma unroll
      for (int a = 0; a < 12; ++a) diag[a] = fmin(fmax(sc[a] * sc[a] * H[tri(a, a) * kPnpBlock + lane], lo), hi);
    }
    reuse_diagonal = true;
    // (Hs + D^2) y = gs, Hs = S H S, gs = S g: Cholesky in L
    // (the factor lives in registers, every loop spelled out: with ONE wave per SIMD — 16 384 hypotheses are one wave per CU — a
    // factor in LDS meant ~650 dependent LDS round trips per solve, most of the kernel's time; same operations, same order, same bits)
    bool solved = true;
    double L[kTri];
#pragma unroll
    for (int a = 0; a < 12; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        double v = sc[a] * sc[b] * H[tri(a, b) * kPnpBlock + lane];
        if (a == b) v += diag[a] / radius;
#pragma unroll
        for (int k = 0; k < b; ++k) v -= L[tri(a, k)] * L[tri(b, k)];
        if (a == b) { if (!(v > 0.0)) solved = false; L[tri(a, a)] = sqrt(v); }
        else L[tri(a, b)] = v / L[tri(b, b)];
      }
    }
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      double v = sc[a] * g[a];
#pragma unroll
      for (int k = 0; k < a; ++k) v -= L[tri(a, k)] * y[k];
      y[a] = v / L[tri(a, a)];
    }
#pragma unroll
    for (int a = 11; a >= 0; --a) {
      double v = y[a];
#pragma unroll
      for (int k = a + 1; k < 12; ++k) v -= L[tri(k, a)] * y[k];
      y[a] = v / L[tri(a, a)];
    }
    // model_cost_change = gs.y - y^T Hs y / 2
    double gy = 0.0, yHy = 0.0;
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      gy += sc[a] * g[a] * y[a];
      double row = 0.0;
#pragma unroll
      for (int b = 0; b < 12; ++b) row += sc[b] * y[b] * H[(a >= b ? tri(a, b) : tri(b, a)) * kPnpBlock + lane];
      yHy += sc[a] * y[a] * row;
    }
    const double model_cost_change = gy - 0.5 * yHy;
    for (int a = 0; a < 12; ++a) solved = solved && isfinite(y[a]);
    solved = solved && isfinite(model_cost_change);
    ++iteration;
    if (!(solved && model_cost_change >= 0.0)) {
      if (++invalid_streak >= 5) { term = 2; break; }
      radius /= decrease_factor; decrease_factor *= 2.0;
    } else {
      invalid_streak = 0;
      double step_sq = 0.0, x_sq = 0.0;
#pragma unroll
      for (int a = 0; a < 12; ++a) { xn[a] = x[a] + (-y[a] * sc[a]); const double e = x[a] - xn[a]; step_sq += e * e; x_sq += x[a] * x[a]; }
      double new_cost = pnp_linearize<false>(A, mdl, sub, xn, H, g, lane);
      if (!isfinite(new_cost)) new_cost = 1.7976931348623157e308;
      if (sqrt(step_sq) <= ptol * (sqrt(x_sq) + ptol)) { term = 0; break; }
      const double cost_change = cost - new_cost;
      if (fabs(cost_change) < ftol * cost) { term = 0; break; }
      const double rho = cost_change / model_cost_change;
      if (rho > min_rel) {
        radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3.0));
        radius = fmin(max_radius, radius); decrease_factor = 2.0; reuse_diagonal = false;
#pragma unroll
        for (int a = 0; a < 12; ++a) x[a] = xn[a];
        cost = pnp_linearize<true>(A, mdl, sub, x, H, g, lane);
        final_cost = fmin(final_cost, cost);
        gmax = 0.0;
#pragma unroll
        for (int a = 0; a < 12; ++a) gmax = fmax(gmax, fabs(g[a]));
        if (gmax <= gtol) { term = 0; break; }
      } else {
        radius /= decrease_factor; decrease_factor *= 2.0;
      }
    }
    if (radius < min_radius) { term = 0; break; }
  }
  const bool usable = term != 2;
#pragma unroll
  for (int a = 0; a < 12; ++a) { if (!usable) x[a] = x0[a]; A.poses_out[(size_t)h * 12 + a] = x[a]; }
  A.status[h] = usable ? 1 : 2;
  A.final_cost[h] = final_cost;

}

// inliers of every hypothesis among all points (solveRSpnp.cpp:225-258, :304-310): tau from the TRUE observation,
// float-rounded projection, float distance.  One lane per (hypothesis, point): H x n independent projections; the
// count is an integer sum (order-independent), one atomic per wave.
__global__ __launch_bounds__(256) void pnp_score_kernel(const PnpArgs A) {
  const int h = blockIdx.x, i = blockIdx.y * 256 + threadIdx.x;   // hypotheses along x (no 65535 limit)
  if (A.status[h] == 0) return;                       // skipped hypothesis: nothing is written
  int in = 0;
  if (i < A.n) {
    const Model mdl{A.shutter, A.scan0, A.scan1, 1};
    double x[12];
#pragma unroll
    for (int a = 0; a < 12; ++a) x[a] = A.poses_out[(size_t)h * 12 + a];
    const double X[3] = {(double)A.object_points[3 * i], (double)A.object_points[3 * i + 1], (double)A.object_points[3 * i + 2]};
    const float ix = A.image_points[2 * i], iy = A.image_points[2 * i + 1];
    const double src = A.shutter == kVertical ? (double)iy : (double)ix;
    ObsOut<true, 2> o;
    eval_observation<true, 2, false>(mdl, A.cam, x, X, src, 0.0, o);   // r = projection - (src, 0)
    const float dx = ix - (float)(o.r[0] + src), dy = iy - (float)o.r[1];
    in = sqrt((double)dx * dx + (double)dy * dy) < (double)A.reprojection_error;
  }
  const unsigned long long ballot = __ballot(in);
  if ((threadIdx.x & 63) == 0 && ballot) atomicAdd(A.num_inliers + h, (int)__popcll(ballot));
}

// inlier flags of ONE pose pair over all points (the winning hypothesis' list, solveRSpnp.cpp:312-326)
__global__ __launch_bounds__(256) void pnp_inliers_kernel(const PnpArgs A, const double* __restrict__ poses, uint8_t* __restrict__ mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n) return;
  const Model mdl{A.shutter, A.scan0, A.scan1, 1};
  double x[12];
#pragma unroll
  for (int a = 0; a < 12; ++a) x[a] = poses[a];
  const double X[3] = {(double)A.object_points[3 * i], (double)A.object_points[3 * i + 1], (double)A.object_points[3 * i + 2]};
  const float ix = A.image_points[2 * i], iy = A.image_points[2 * i + 1];
  const double src = A.shutter == kVertical ? (double)iy : (double)ix;
  ObsOut<true, 2> o;
  eval_observation<true, 2, false>(mdl, A.cam, x, X, src, 0.0, o);
  const float dx = ix - (float)(o.r[0] + src), dy = iy - (float)o.r[1];
  mask[i] = sqrt((double)dx * dx + (double)dy * dy) < (double)A.reprojection_error;
}

}  // namespace

hipError_t launch_pnp_tasks(const PnpArgs& A, hipStream_t st) {
  const size_t lds = (size_t)kTri * kPnpBlock * sizeof(double);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pnp_tasks_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(pnp_tasks_kernel, dim3((A.num_tasks + kPnpBlock - 1) / kPnpBlock), dim3(kPnpBlock), lds, st, A);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  // num_inliers must be zero on entry (the C ABI clears it)
  hipLaunchKernelGGL(pnp_score_kernel, dim3(A.num_tasks, (A.n + 255) / 256), dim3(256), 0, st, A);
  return hipGetLastError();
}
hipError_t launch_pnp_inliers(const PnpArgs& A, const double* poses, uint8_t* mask, hipStream_t st) {
  hipLaunchKernelGGL(pnp_inliers_kernel, dim3((A.n + 255) / 256), dim3(256), 0, st, A, poses, mask);
  return hipGetLastError();
}

}  // namespace rsba
