// The loss-corrected, masked, column-scaled residual and Jacobian of ONE observation as the LM iteration uses them — what
// Ceres' ResidualBlock::Evaluate + Corrector + the Jacobi column scaling leave of a residual block (SURVEY Appendix C.2-C.5;
// call site /root/reference/src/rsba/CeresHandler.h:419).  One function, used by every kernel that needs the record:
//   * the LM mode of the evaluation kernel (cost, per-frame camera blocks), and
//   * for calibrated problems the point-side passes of the solve (V_j / g_p, the P records, the point steps), which
//     RECOMPUTE it from the 24-byte observation instead of reading a 256-byte point-major copy back from HBM: an observation
//     is ~0.6 kflop of fp64, cheaper than the 256 B it saves three times per iteration (round 2 wrote the records once and read
//     them three times: 2.1 GB of the iteration's traffic at 1k cameras).
// The same inputs give the same bits wherever it is inlined (no fast-math reassociation).
#pragma once
#include "device_state.hpp"
#include "obs_math.hpp"

namespace rsba {

// pose / pose_scale: the frame's CD = 6 P doubles (any address space the caller staged them in).
// Out: o.r, o.J scaled by sqrt(rho') and the column scales, zeros when the functor failed; half_rho = rho0 / 2 of the block
// (0 when it failed), dropped = every parameter block of it is constant (its cost is the fixed cost then).
template <bool CAL, int P>
__device__ __forceinline__ void lm_observation(const DeviceProblem& dp, int f, int j, double x, double y, const double* __restrict__ pose,
                                               const double* __restrict__ pose_scale, ObsOut<CAL, P>& o, double& half_rho, bool& dropped) {
  constexpr int CD = 6 * P;
  constexpr int K = ObsOut<CAL, P>::K;
  constexpr int OFF_POSE = CAL ? 0 : 9;
  constexpr int OFF_PT = OFF_POSE + CD;
  double X[3], cam[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) X[k] = dp.points[(size_t)j * 3 + k];
  const int ci = (dp.NI == 1) ? 0 : dp.frame_intr[f];
#pragma unroll
  for (int k = 0; k < 9; ++k) cam[k] = dp.intr[(size_t)ci * 9 + k];
  const Model m = {(dp.frame_global && dp.frame_global[f]) ? (int)kGlobal : dp.shutter, dp.scan0, dp.scan1, dp.interp_rotation};   // (a one-pose frame of a two-pose session: CeresHandler.h:266-285)
  eval_observation<CAL, P, true>(m, cam, pose, X, x, y, o);
  // Ceres 1.9 ResidualBlock::Evaluate: cost = rho0/2 from the uncorrected residual
  const double s = o.r[0] * o.r[0] + o.r[1] * o.r[1];
  double rho[3] = {s, 1.0, 0.0};
  if (dp.huber_a > 0.0) huber_rho(dp.huber_a, s, rho);
  half_rho = o.ok ? 0.5 * rho[0] : 0.0;
  double sc[K];
  if (!CAL) {
#pragma unroll
    for (int k = 0; k < 9; ++k) sc[k] = dp.scale_intr[(size_t)ci * 9 + k];
  }
#pragma unroll
  for (int k = 0; k < CD; ++k) sc[OFF_POSE + k] = pose_scale[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) sc[OFF_PT + k] = dp.scale_point[(size_t)j * 3 + k];
  // a residual block whose parameter blocks are all constant leaves the reduced program; its
  // cost is carried as fixed_cost.  Column scale 0 <=> fixed coordinate.
  dropped = true;
#pragma unroll
  for (int k = 0; k < K; ++k) dropped = dropped && (sc[k] == 0.0);
  // Corrector (Ceres 1.9 corrector.cc) for rho'' <= 0, which always holds for Huber: residual
  // and Jacobian rows are scaled by sqrt(rho').
  const double sr1 = dp.huber_a > 0.0 ? sqrt(rho[1]) : 1.0;   // (without a loss rho' is exactly 1: the same bits, and no fp64 square root — thirty instructions — per observation)
  o.r[0] *= sr1; o.r[1] *= sr1;
#pragma unroll
  for (int k = 0; k < K; ++k) { const double c = sr1 * sc[k]; o.J[0][k] *= c; o.J[1][k] *= c; }
  if (P == 2) {   // the unscaled pose block gets the loss correction only: its column scales are applied where the Schur partials are merged
#pragma unroll
    for (int k = 0; k < 6; ++k) { o.Jq[0][k] *= sr1; o.Jq[1][k] *= sr1; }
  }
  // a failed block contributes zeros everywhere (record, camera blocks)
  if (!o.ok) {
    o.r[0] = 0.0; o.r[1] = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) { o.J[0][k] = 0.0; o.J[1][k] = 0.0; }
    if (P == 2) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { o.Jq[0][k] = 0.0; o.Jq[1][k] = 0.0; }
    }
  }
}

}  // namespace rsba
