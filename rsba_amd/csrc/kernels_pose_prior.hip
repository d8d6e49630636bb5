// SURVEY §8f row f1, second half — the per-pose prior blocks CeresHandler::Add attaches:
//   GoodPosePrior   /root/reference/src/rsba/CeresHandler.h:52-73, attached at :188-204.  Six residuals
//                   r = W (prior - pose), W = diag(rotation x3, position x3), over TWO parameter blocks: the frame's
//                   priorPoses[i] — a FREE block, like everything Ceres is handed — and poses[i].  Fails when r[0] >= 1.
//   SphericalPrior  :36-50, attached at :127-130 to poses[0] of frame 1 of a session that starts at the origin.  Two
//                   residuals |rot|^2 and 1e20 (1 - |cx| - |cy| - |cz|).  Fails when |rot|^2 >= 1.
// Neither block carries a loss function (nullptr in the reference).
//
// A priorPoses block meets exactly one other block (its pose) and its Jacobian is diagonal, so it is eliminated in
// closed form, coordinate by coordinate, the way the points are: with v0 = (w s0)^2, g0 = w s0 r, c = -w^2 s0 sp
// (s0 / sp the column scales of the prior / pose coordinate) and V0' = v0 + D0^2,
//     S_pp -= c^2 / V0',   rhs_p -= c g0 / V0',   y0 = (g0 - c y_p) / V0'.
// Everything here is O(#poses): one thread per block, reductions in block order by one workgroup (deterministic).
#include "obs_math.hpp"
#include "solver_state.hpp"

namespace rsba {

namespace {

constexpr int kPPThreads = 256;

__device__ __forceinline__ double block_sum(double v, double* s_red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  const double t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  __syncthreads();
  return t;
}
__device__ __forceinline__ double block_max(double v, double* s_red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  const double t = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  return t;
}

__device__ __forceinline__ double pp_weight(const DeviceProblem& dp, int i) { return i < 3 ? dp.pp_rotation : dp.pp_position; }

// SphericalPrior at one pose: residuals, validity, Jacobian rows (row 0 over the rotation, row 1 over the position)
struct Spherical { double r0, r1, j0[3], j1[3]; bool ok; };
__device__ __forceinline__ Spherical spherical_at(const double* pose) {
  Spherical s;
  s.r0 = pose[0] * pose[0] + pose[1] * pose[1] + pose[2] * pose[2];
  s.r1 = 1e20 * (1.0 - fabs(pose[3]) - fabs(pose[4]) - fabs(pose[5]));
  for (int k = 0; k < 3; ++k) { s.j0[k] = pose[k] + pose[k]; s.j1[k] = pose[3 + k] < 0.0 ? 1e20 : -1e20; }   // Jet abs: -x for x < 0, x otherwise
  s.ok = s.r0 < 1.0;
  return s;
}

// cost of the blocks at (poses, prior values): cost2[0] += 1/2 sum |r|^2 (cost2[1] for a SphericalPrior on a constant pose),
// fail_count += blocks whose functor returns false
__global__ __launch_bounds__(kPPThreads) void pose_prior_cost_kernel(const DeviceProblem dp, double* cost2) {
  if (lm_stopped(dp.ctl)) return;   // (device-side trust region: iterations enqueued behind a termination fall through)
  __shared__ double s_red[4];
  double c = 0.0, bad = 0.0;
  for (int k = threadIdx.x; k < dp.pp_count; k += kPPThreads) {
    const double* pose = dp.poses + (size_t)dp.pp_block[k] * 6; const double* p0 = dp.pp_value + (size_t)k * 6;
    double sq = 0.0, r0 = 0.0;
    for (int i = 0; i < 6; ++i) { const double r = (p0[i] - pose[i]) * pp_weight(dp, i); if (i == 0) r0 = r; sq += r * r; }
    if (r0 < 1.0) c += 0.5 * sq; else bad += 1.0;
  }
  c = block_sum(c, s_red); bad = block_sum(bad, s_red);
  if (threadIdx.x != 0) return;
  double fixed = 0.0;
  if (dp.pp_spherical >= 0) {
    const Spherical s = spherical_at(dp.poses + (size_t)dp.pp_spherical * 6);
    bool all_const = true;
    for (int i = 0; i < 6; ++i) all_const = all_const && dp.scale_pose[(size_t)dp.pp_spherical * 6 + i] == 0.0;
    const double v = 0.5 * (s.r0 * s.r0 + s.r1 * s.r1);
    if (!s.ok) bad += 1.0; else if (all_const) fixed = v; else c += v;
  }
  cost2[0] += c; cost2[1] += fixed;
  if (bad > 0.0) *dp.fail_count += (int)bad;
}

// linearisation: U_f, g_f += the blocks' J^T J / J^T r over the pose coordinates (lead rank only); per priorPoses coordinate v0, g0, c (every rank)
__global__ __launch_bounds__(kPPThreads) void pose_prior_blocks_kernel(const DeviceProblem dp, const SolverDev sv, double* __restrict__ v0, double* __restrict__ g0,
                                                                        double* __restrict__ cross) {
  if (lm_not_accepted(sv.ctl)) return;   // (device-side trust region: a rejected candidate is not linearised)
  const int CD = sv.CD;
  for (int k = threadIdx.x; k < dp.pp_count; k += kPPThreads) {
    const int b = dp.pp_block[k], f = b / dp.P, q = b % dp.P;
    const double* pose = dp.poses + (size_t)b * 6; const double* p0 = dp.pp_value + (size_t)k * 6;
    for (int i = 0; i < 6; ++i) {
      const double w = pp_weight(dp, i), r = (p0[i] - pose[i]) * w;
      const double s0 = dp.pp_scale[(size_t)k * 6 + i], sp = dp.scale_pose[(size_t)b * 6 + i];
      const int a = 6 * q + i;
      if (sv.frame_lead ? sv.frame_lead[f] != 0.0 : sv.lead != 0) {   // replicated terms of the normal equations: contributed once — by the lead rank, or (sharded factorisation) by the rank whose part holds the frame
        sv.U[((size_t)f * CD + a) * CD + a] += (w * sp) * (w * sp);
        sv.gc[(size_t)f * CD + a] += -(w * sp) * r;
      }
      v0[(size_t)k * 6 + i] = (w * s0) * (w * s0);
      g0[(size_t)k * 6 + i] = (w * s0) * r;
      cross[(size_t)k * 6 + i] = -(w * s0) * (w * sp);
    }
  }
  __syncthreads();   // a SphericalPrior may sit on a pose that also carries a GoodPosePrior: after the loop above
  if (threadIdx.x == 0 && dp.pp_spherical >= 0 && (sv.frame_lead ? sv.frame_lead[dp.pp_spherical / dp.P] != 0.0 : sv.lead != 0)) {   // (contributed once: the lead rank, or — sharded factorisation — the rank whose part holds the pose, like a GoodPosePrior's terms)
    const int b = dp.pp_spherical, f = b / dp.P, q = b % dp.P;
    const Spherical s = spherical_at(dp.poses + (size_t)b * 6);
    double J[2][6];
    for (int i = 0; i < 6; ++i) {
      const double sp = dp.scale_pose[(size_t)b * 6 + i];
      J[0][i] = i < 3 ? s.j0[i] * sp : 0.0; J[1][i] = i < 3 ? 0.0 : s.j1[i - 3] * sp;
    }
    for (int i = 0; i < 6; ++i) {
      sv.gc[(size_t)f * CD + 6 * q + i] += J[0][i] * s.r0 + J[1][i] * s.r1;
      for (int j = 0; j < 6; ++j) sv.U[((size_t)f * CD + 6 * q + i) * CD + 6 * q + j] += J[0][i] * J[0][j] + J[1][i] * J[1][j];
    }
  }
}

// Jacobi scale of the priorPoses coordinates, 1 / (1 + sqrt(|J_i|^2)), from the first linearisation (their scale is 1 then)
__global__ void pose_prior_scale_kernel(const DeviceProblem dp, const double* v0) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < 6 * dp.pp_count) dp.pp_scale[t] *= 1.0 / (1.0 + sqrt(v0[t]));
}
__global__ void pose_prior_clamp_kernel(int n, const double* v0, double* diag, double lo, double hi) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n) diag[t] = fmin(fmax(v0[t], lo), hi);
}
// max |g_i| of the UNSCALED gradient over the priorPoses coordinates, folded into scalars[kGradMax] — or (out != null: the loop that runs
// without the host) left as one more partial maximum for the kernel that reduces them and decides
__global__ __launch_bounds__(kPPThreads) void pose_prior_gradmax_kernel(const DeviceProblem dp, const SolverDev sv, const double* g0, double* out) {
  __shared__ double s_red[4];
  double m = 0.0;
  for (int t = threadIdx.x; t < 6 * dp.pp_count; t += kPPThreads) { const double s0 = dp.pp_scale[t]; if (s0 > 0.0) m = fmax(m, fabs(g0[t] / s0)); }
  m = block_max(m, s_red);
  if (threadIdx.x == 0) { if (out) *out = m; else sv.scalars[kGradMax] = fmax(sv.scalars[kGradMax], m); }
}
// x = x + delta for the priorPoses values, where the device-side loop has accepted the candidate (the host form swaps the two buffers)
__global__ void pose_prior_take_kernel(const DeviceProblem dp, const SolverDev sv) {
  if (lm_not_accepted(sv.ctl)) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < 6 * dp.pp_count) dp.pp_value[t] = dp.pp_trial[t];
}

// elimination of the priorPoses blocks from the reduced camera system (after the Schur merge)
__global__ void pose_prior_reduce_kernel(const DeviceProblem dp, const SolverDev sv, const double* v0, const double* g0, const double* cross,
                                         const double* diag, double inv_radius, const int32_t* tile_diag_slot) {
  if (sv.ctl) { if (lm_stopped(sv.ctl)) return; inv_radius = 1.0 / sv.ctl[kCtlRadius]; }   // (device-side trust region: the radius lives in HBM)
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 6 * dp.pp_count) return;
  const int k = t / 6, i = t % 6;
  if (!(sv.frame_lead ? sv.frame_lead[dp.pp_block[k] / dp.P] != 0.0 : sv.lead != 0)) return;   // (the rank that added the block's terms takes them out again)
  const int64_t cc = (int64_t)dp.pp_block[k] * 6 + i;               // camera-side coordinate of the pose entry
  const int I = (int)(cc / kTile), r = (int)(cc % kTile);
  const double vp = v0[t] + diag[t] * inv_radius;
  double* Sd = sv.S + (size_t)tile_diag_slot[I] * (kTile * kTile) + (size_t)r * kTile + r;
  *Sd -= cross[t] * cross[t] / vp;
  sv.rhs[cc] -= cross[t] * g0[t] / vp;
}

// after the camera step (sv.step) is known: y0, the candidate prior values, and the blocks' share of the model cost change,
// |step|^2 and |x|^2 (added to the scalars the point / camera kernels have already written)
__global__ __launch_bounds__(kPPThreads) void pose_prior_step_kernel(const DeviceProblem dp, const SolverDev sv, const double* v0, const double* g0,
                                                                      const double* cross, const double* diag, double inv_radius) {
  if (sv.ctl) { if (lm_stopped(sv.ctl)) return; inv_radius = 1.0 / sv.ctl[kCtlRadius]; }
  __shared__ double s_red[4];
  double acc = 0.0, st = 0.0, xx = 0.0;
  for (int k = threadIdx.x; k < dp.pp_count; k += kPPThreads) {
    const int b = dp.pp_block[k];
    const double* pose = dp.poses + (size_t)b * 6; const double* p0 = dp.pp_value + (size_t)k * 6;
    for (int i = 0; i < 6; ++i) {
      const int t = k * 6 + i;
      const double yp = sv.step[(size_t)b * 6 + i];
      const double y0 = (g0[t] - cross[t] * yp) / (v0[t] + diag[t] * inv_radius);
      const double w = pp_weight(dp, i), s0 = dp.pp_scale[t], sp = dp.scale_pose[(size_t)b * 6 + i];
      const double r = (p0[i] - pose[i]) * w;
      const double m = -(w * s0) * y0 + (w * sp) * yp;                  // J (-y), J = [w s0 | -w sp]
      acc += m * (r + 0.5 * m);
      const double xn = p0[i] + (-y0 * s0);
      dp.pp_trial[t] = xn;
      const double e = p0[i] - xn;
      st += e * e; xx += p0[i] * p0[i];
    }
  }
  acc = block_sum(acc, s_red); st = block_sum(st, s_red); xx = block_sum(xx, s_red);
  if (threadIdx.x != 0 || !sv.lead) return;   // every rank of a sharded solve forms the candidate values; the scalars are summed over the ranks
  if (dp.pp_spherical >= 0) {
    const int b = dp.pp_spherical;
    const Spherical s = spherical_at(dp.poses + (size_t)b * 6);
    double m0 = 0.0, m1 = 0.0;
    for (int i = 0; i < 3; ++i) {
      m0 += s.j0[i] * dp.scale_pose[(size_t)b * 6 + i] * -sv.step[(size_t)b * 6 + i];
      m1 += s.j1[i] * dp.scale_pose[(size_t)b * 6 + 3 + i] * -sv.step[(size_t)b * 6 + 3 + i];
    }
    acc += m0 * (s.r0 + 0.5 * m0) + m1 * (s.r1 + 0.5 * m1);
  }
  sv.scalars[kModelCostChange] += -acc;
  sv.scalars[kStepSq] += st;
  sv.scalars[kXSq] += xx;
}

}  // namespace

#define PP_LAUNCH(kernel, grid, block, st, ...)                    \
  do {                                                             \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, st, __VA_ARGS__); \
    hipError_t e_ = hipGetLastError();                             \
    if (e_ != hipSuccess) return e_;                               \
  } while (0)

static inline bool has_pose_priors(const DeviceProblem& dp) { return dp.pp_count > 0 || dp.pp_spherical >= 0; }
static inline int pp_grid(const DeviceProblem& dp) { return (6 * dp.pp_count + 255) / 256; }

hipError_t launch_pose_prior_cost(const DeviceProblem& dp, double* cost2, hipStream_t st) {
  if (!has_pose_priors(dp)) return hipSuccess;
  PP_LAUNCH(pose_prior_cost_kernel, 1, kPPThreads, st, dp, cost2);
  return hipSuccess;
}
hipError_t launch_pose_prior_blocks(const DeviceProblem& dp, const SolverDev& sv, const PosePriorDev& pp, hipStream_t st) {
  if (!has_pose_priors(dp)) return hipSuccess;
  PP_LAUNCH(pose_prior_blocks_kernel, 1, kPPThreads, st, dp, sv, pp.v0, pp.g0, pp.cross);
  return hipSuccess;
}
hipError_t launch_pose_prior_scale(const DeviceProblem& dp, const PosePriorDev& pp, hipStream_t st) {
  if (dp.pp_count <= 0) return hipSuccess;
  PP_LAUNCH(pose_prior_scale_kernel, pp_grid(dp), 256, st, dp, pp.v0);
  return hipSuccess;
}
hipError_t launch_pose_prior_clamp(const DeviceProblem& dp, const PosePriorDev& pp, double lo, double hi, hipStream_t st) {
  if (dp.pp_count <= 0) return hipSuccess;
  PP_LAUNCH(pose_prior_clamp_kernel, pp_grid(dp), 256, st, 6 * dp.pp_count, pp.v0, pp.diag, lo, hi);
  return hipSuccess;
}
hipError_t launch_pose_prior_gradmax(const DeviceProblem& dp, const SolverDev& sv, const PosePriorDev& pp, hipStream_t st, double* out) {
  if (dp.pp_count <= 0) return hipSuccess;
  PP_LAUNCH(pose_prior_gradmax_kernel, 1, kPPThreads, st, dp, sv, pp.g0, out);
  return hipSuccess;
}
hipError_t launch_pose_prior_take(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  if (dp.pp_count <= 0) return hipSuccess;
  PP_LAUNCH(pose_prior_take_kernel, pp_grid(dp), 256, st, dp, sv);
  return hipSuccess;
}
hipError_t launch_pose_prior_reduce(const DeviceProblem& dp, const SolverDev& sv, const PosePriorDev& pp, double radius, hipStream_t st) {
  if (dp.pp_count <= 0) return hipSuccess;
  PP_LAUNCH(pose_prior_reduce_kernel, pp_grid(dp), 256, st, dp, sv, pp.v0, pp.g0, pp.cross, pp.diag, 1.0 / radius, pp.tile_diag_slot);
  return hipSuccess;
}
hipError_t launch_pose_prior_step(const DeviceProblem& dp, const SolverDev& sv, const PosePriorDev& pp, double radius, hipStream_t st) {
  if (!has_pose_priors(dp)) return hipSuccess;
  PP_LAUNCH(pose_prior_step_kernel, 1, kPPThreads, st, dp, sv, pp.v0, pp.g0, pp.cross, pp.diag, 1.0 / radius);
  return hipSuccess;
}

}  // namespace rsba
