// Frame-to-frame motion priors: RsConstVeloPrior / RsConstAccelerationPrior
// (/root/reference/src/rsba/video_bundler_rs_inter.h:55-108, :113-173) as CeresHandler::Add creates them
// (CeresHandler.h:147-185).  With the ratio block constant (opt.ceres.interFrameRatio != 1, :175-177) a prior is a
// 12-residual block over the four pose blocks of frames f and f-1 that is LINEAR in them; with the ratio free (the
// reference's default) the same blocks are linearised at the current ratio and the ratio's own column is formed by
// prior_border_kernel (solver.hip serves it as a 1-wide border of S).  Per pose coordinate i the two residuals
//     r[i]     = s_i * (Ca . x_i),     r[6 + i] = s_i * (Cb . x_i),     x_i = (f.p0[i], f.p1[i], (f-1).p0[i], (f-1).p1[i])
// touch only that coordinate of the four poses (s_i = scale * 0.01 for the rotation rows, scale otherwise), so a
// prior adds 2 x 2 blocks per coordinate to U_f, U_{f-1} and to the (f, f-1) block of the reduced camera system —
// the block-tridiagonal fill SURVEY names.  The shared loss function acts on the 12-vector as a whole (:155,:166).
//
// O(F) work per LM iteration: one thread per frame gathers the prior it heads and the one that refers back to it
// (no atomics, fixed order); the cost and the model cost change are sums of per-wave partials taken in wave order.
#include "obs_math.hpp"
#include "solver_state.hpp"

namespace rsba {

namespace {

__device__ __forceinline__ double wsum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ double prior_q(const DeviceProblem& dp) { return dp.prior_ratio_ptr ? *dp.prior_ratio_ptr : dp.prior_ratio; }

// Jacobian coefficients of the two residual rows with respect to x_i (what Jet arithmetic leaves in the functors)
__device__ __forceinline__ void prior_coefficients(const DeviceProblem& dp, double Ca[4], double Cb[4]) {
  const double q = prior_q(dp), inv = 1.0 / q;
  if (dp.prior_kind == 1) {
    Ca[0] = 1.0; Ca[1] = 0.0; Ca[2] = q; Ca[3] = -(1.0 + q);
    if (q > 2.220446049250313e-16) { Cb[0] = -(1.0 + inv); Cb[1] = 1.0; Cb[2] = 0.0; Cb[3] = inv; }
    else { Cb[0] = -1.0; Cb[1] = 1.0; Cb[2] = 1.0; Cb[3] = -1.0; }
  } else {
    Ca[0] = 0.5; Ca[1] = 0.0; Ca[2] = 0.5 * q; Ca[3] = -0.5 * (1.0 + q);
    Cb[0] = -0.5 * (1.0 + inv); Cb[1] = 0.5; Cb[2] = 0.0; Cb[3] = 0.5 * inv;
  }
}

// the functors' residuals in their own order of operations (T = double path); loss weight rho' and cost rho / 2
struct PriorValue { double r[12]; double weight, cost; };
__device__ __forceinline__ PriorValue prior_value(const DeviceProblem& dp, int f) {
  PriorValue v;
  const double* cur = dp.poses + (size_t)f * 12; const double* prev = cur - 12;
  const double q = prior_q(dp);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double c0 = cur[i], c1 = cur[6 + i], p0 = prev[i], p1 = prev[6 + i];
    double a, b;
    if (dp.prior_kind == 1) {
      a = c0 - (p1 + (p1 - p0) * q);
      const double d = (q > 2.220446049250313e-16) ? (c0 - p1) * (1.0 / q) : (p1 - p0);
      b = c1 - (c0 + d);
    } else {
      const double vt = c0 - p1, v1t = (p1 - p0) * q;
      a = c0 - (p1 + (v1t + (vt - v1t) * 0.5));
      const double vv = c1 - c0, w = (c0 - p1) * (1.0 / q);
      b = c1 - (c0 + (w + (vv - w) * 0.5));
    }
    const double down = i < 3 ? 0.01 : 1.0;
    v.r[i] = a * dp.prior_scale * down; v.r[6 + i] = b * dp.prior_scale * down;
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += v.r[i] * v.r[i];
  double rho[3] = {s, 1.0, 0.0};
  if (dp.huber_a > 0.0) huber_rho(dp.huber_a, s, rho);
  v.weight = rho[1]; v.cost = 0.5 * rho[0];
  return v;
}

// d r / d interFrameRatio of the 12 residuals (the Jacobian column of the ratio block when it is a free parameter)
__device__ __forceinline__ void prior_ratio_column(const DeviceProblem& dp, int f, double dr[12]) {
  const double* cur = dp.poses + (size_t)f * 12; const double* prev = cur - 12;
  const double q = prior_q(dp), iq2 = 1.0 / (q * q);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double c0 = cur[i], p0 = prev[i], p1 = prev[6 + i];
    double da, db;
    if (dp.prior_kind == 1) { da = -(p1 - p0); db = (q > 2.220446049250313e-16) ? (c0 - p1) * iq2 : 0.0; }
    else { da = -0.5 * (p1 - p0); db = 0.5 * (c0 - p1) * iq2; }
    const double si = dp.prior_scale * (i < 3 ? 0.01 : 1.0);
    dr[i] = da * si; dr[6 + i] = db * si;
  }
}

// U_f, g_f and the (f, f-1) cross block: one thread per frame
__global__ __launch_bounds__(64) void prior_blocks_kernel(const DeviceProblem dp, const SolverDev sv, double* __restrict__ ucross) {
  if (lm_not_accepted(sv.ctl)) return;   // (device-side trust region: a rejected candidate is not linearised)
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= dp.F) return;
  const bool heads = dp.prior_of[f] != 0, referred = dp.prior_of[f + 1] != 0;
  if (!heads && !referred) return;
  double Ca[4], Cb[4];
  prior_coefficients(dp, Ca, Cb);
  PriorValue mine, next;
  if (heads) mine = prior_value(dp, f);
  if (referred) next = prior_value(dp, f + 1);
  const double* sc = dp.scale_pose + (size_t)f * 12;
  double* U = sv.U + (size_t)f * 144; double* g = sv.gc + (size_t)f * 12;
  for (int i = 0; i < 6; ++i) {
    const double si = dp.prior_scale * (i < 3 ? 0.01 : 1.0), s2 = si * si;
    double H[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, gv[2] = {0.0, 0.0};
    if (heads)
      for (int p = 0; p < 2; ++p) {
        gv[p] += mine.weight * si * (Ca[p] * mine.r[i] + Cb[p] * mine.r[6 + i]);
        for (int q = 0; q < 2; ++q) H[p][q] += mine.weight * s2 * (Ca[p] * Ca[q] + Cb[p] * Cb[q]);
      }
    if (referred)
      for (int p = 0; p < 2; ++p) {
        gv[p] += next.weight * si * (Ca[2 + p] * next.r[i] + Cb[2 + p] * next.r[6 + i]);
        for (int q = 0; q < 2; ++q) H[p][q] += next.weight * s2 * (Ca[2 + p] * Ca[2 + q] + Cb[2 + p] * Cb[2 + q]);
      }
    for (int p = 0; p < 2; ++p) {
      const int a = 6 * p + i;
      g[a] += gv[p] * sc[a];
      for (int q = 0; q < 2; ++q) { const int b = 6 * q + i; U[a * 12 + b] += H[p][q] * sc[a] * sc[b]; }
    }
    if (heads) {
      const double* scp = sc - 12;   // frame f - 1
      for (int p = 0; p < 2; ++p) for (int q = 0; q < 2; ++q) {
        const int a = 6 * p + i, b = 6 * q + i;
        ucross[(size_t)f * 144 + a * 12 + b] = mine.weight * s2 * (Ca[p] * Ca[2 + q] + Cb[p] * Cb[2 + q]) * sc[a] * scp[b];
      }
    }
  }
}

// The two reductions below run one wave per 64 frames across the chip.  Every wave leaves its partial in
// dp.prior_partial and draws a ticket; the wave that draws the last one adds the partials IN WAVE ORDER (so the sum does
// not depend on which wave that is: deterministic, no floating-point atomics) and re-arms the ticket for the next launch.
// Partials cross workgroups through agent-scope fences around the ticket.
__device__ __forceinline__ bool last_wave_of_grid(const DeviceProblem& dp) {
  __threadfence();                                                  // partials visible before the ticket
  const unsigned t = atomicAdd(dp.prior_ticket, 1u);
  if (t != gridDim.x - 1) return false;
  __threadfence();                                                  // and read only after it
  *dp.prior_ticket = 0u;
  return true;
}

// cost of the prior blocks at dp.poses, added to {cost, fixed cost}; a block whose four poses are all constant (and whose
// ratio block is constant too) is not part of the reduced program and its cost is "fixed" (Ceres: Program::RemoveFixedBlocks)
__global__ __launch_bounds__(64) void prior_cost_kernel(const DeviceProblem dp, double* cost2, int invalid) {
  if (lm_stopped(dp.ctl)) return;        // (device-side trust region: iterations behind a termination fall through — every wave alike: the ticket stays armed)
  const int f = blockIdx.x * 64 + threadIdx.x;
  double c = 0.0, cf = 0.0;
  if (f < dp.F && dp.prior_of[f]) {
    const PriorValue v = prior_value(dp, f);
    bool all_const = dp.prior_free == 0;   // with a free ratio the block always keeps one variable parameter block
    for (int k = 0; k < 24; ++k) all_const = all_const && dp.scale_pose[(size_t)(f - 1) * 12 + k] == 0.0;
    if (all_const) cf = v.cost; else c = v.cost;
  }
  c = wsum64(c); cf = wsum64(cf);
  if (threadIdx.x != 0) return;
  dp.prior_partial[2 * blockIdx.x] = c; dp.prior_partial[2 * blockIdx.x + 1] = cf;
  if (!last_wave_of_grid(dp)) return;
  double a = 0.0, b = 0.0;
  for (unsigned w = 0; w < gridDim.x; ++w) { a += dp.prior_partial[2 * w]; b += dp.prior_partial[2 * w + 1]; }   // fixed order
  cost2[0] += a;
  cost2[1] += b;
  if (invalid) *dp.fail_count += invalid;   // the functors return false for this interFrameRatio
}

// model cost change of the prior blocks for the camera step in sv.step:  -sum m.(r~ + m/2),  m = -J~ y
__global__ __launch_bounds__(64) void prior_model_kernel(const DeviceProblem dp, const SolverDev sv, double* out, double ratio_step, const double* ratio_step_ptr) {
  if (lm_stopped(dp.ctl)) return;
  if (ratio_step_ptr) { const double c = *ratio_step_ptr; ratio_step = isfinite(c) ? c : 0.0; }   // (device-side trust region: the ETA task of the factorisation left s eta there)
  const int f = blockIdx.x * 64 + threadIdx.x;
  double acc = 0.0;
  if (f < dp.F && dp.prior_of[f]) {
    double Ca[4], Cb[4];
    prior_coefficients(dp, Ca, Cb);
    const PriorValue v = prior_value(dp, f);
    const double sw = sqrt(v.weight);
    const double* y = sv.step + (size_t)(f - 1) * 12; const double* sc = dp.scale_pose + (size_t)(f - 1) * 12;   // [prev | cur]
    double dr[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) dr[i] = 0.0;
    if (ratio_step != 0.0) prior_ratio_column(dp, f, dr);            // free ratio: its (scaled) step enters m = -J~ y too
    for (int i = 0; i < 6; ++i) {
      const double si = dp.prior_scale * (i < 3 ? 0.01 : 1.0);
      const double x0 = sc[12 + i] * y[12 + i], x1 = sc[18 + i] * y[18 + i], x2 = sc[i] * y[i], x3 = sc[6 + i] * y[6 + i];
      const double ma = -sw * (si * (Ca[0] * x0 + Ca[1] * x1 + Ca[2] * x2 + Ca[3] * x3) + dr[i] * ratio_step);
      const double mb = -sw * (si * (Cb[0] * x0 + Cb[1] * x1 + Cb[2] * x2 + Cb[3] * x3) + dr[6 + i] * ratio_step);
      acc += ma * (sw * v.r[i] + 0.5 * ma) + mb * (sw * v.r[6 + i] + 0.5 * mb);
    }
  }
  acc = wsum64(acc);
  if (threadIdx.x != 0) return;
  dp.prior_partial[blockIdx.x] = acc;
  if (!last_wave_of_grid(dp)) return;
  double a = 0.0;
  for (unsigned w = 0; w < gridDim.x; ++w) a += dp.prior_partial[w];   // fixed order
  *out += -a;
}

// Free interFrameRatio: the column of the ratio in the normal equations.  border[t] = (J~_x^T J~_ratio) of camera
// coordinate t (camera scales applied, the ratio's own scale is the host's), hg = {J~_ratio^T J~_ratio, J~_ratio^T r~}.
// One thread per frame gathers the prior it heads and the one referring back to it; hg by the last-wave reduction.
__global__ __launch_bounds__(64) void prior_border_kernel(const DeviceProblem dp, const SolverDev sv, double* __restrict__ border, double* __restrict__ hg) {
  if (lm_not_accepted(sv.ctl)) return;   // (device-side trust region: a rejected candidate is not linearised — every wave alike: the ticket stays armed)
  const int f = blockIdx.x * 64 + threadIdx.x;
  double hh = 0.0, gg = 0.0;
  if (f < dp.F) {
    const bool heads = dp.prior_of[f] != 0, referred = dp.prior_of[f + 1] != 0;
    double Ca[4], Cb[4];
    prior_coefficients(dp, Ca, Cb);
    double out[12];
#pragma unroll
    for (int a = 0; a < 12; ++a) out[a] = 0.0;
    if (heads) {
      const PriorValue v = prior_value(dp, f);
      double dr[12];
      prior_ratio_column(dp, f, dr);
      for (int i = 0; i < 6; ++i) {
        const double si = dp.prior_scale * (i < 3 ? 0.01 : 1.0);
        for (int p = 0; p < 2; ++p) out[6 * p + i] += v.weight * si * (Ca[p] * dr[i] + Cb[p] * dr[6 + i]);
        hh += v.weight * (dr[i] * dr[i] + dr[6 + i] * dr[6 + i]);
        gg += v.weight * (dr[i] * v.r[i] + dr[6 + i] * v.r[6 + i]);
      }
    }
    if (referred) {
      const PriorValue v = prior_value(dp, f + 1);
      double dr[12];
      prior_ratio_column(dp, f + 1, dr);
      for (int i = 0; i < 6; ++i) {
        const double si = dp.prior_scale * (i < 3 ? 0.01 : 1.0);
        for (int p = 0; p < 2; ++p) out[6 * p + i] += v.weight * si * (Ca[2 + p] * dr[i] + Cb[2 + p] * dr[6 + i]);
      }
    }
    const double* sc = dp.scale_pose + (size_t)f * 12;
#pragma unroll
    for (int a = 0; a < 12; ++a) border[(size_t)f * 12 + a] = out[a] * sc[a];
  }
  hh = wsum64(hh); gg = wsum64(gg);
  if (threadIdx.x != 0) return;
  dp.prior_partial[2 * blockIdx.x] = hh; dp.prior_partial[2 * blockIdx.x + 1] = gg;
  if (!last_wave_of_grid(dp)) return;
  double a = 0.0, b = 0.0;
  for (unsigned w = 0; w < gridDim.x; ++w) { a += dp.prior_partial[2 * w]; b += dp.prior_partial[2 * w + 1]; }
  hg[0] = a; hg[1] = b;
}

// out[0] = b.u, out[1] = b.v over n entries (one workgroup, fixed order)
__global__ __launch_bounds__(1024) void border_dots_kernel(const double* __restrict__ b, const double* __restrict__ u, const double* __restrict__ v, int64_t n, double* out) {
  __shared__ double s_red[2][16];
  double du = 0.0, dv = 0.0;
  for (int64_t t = threadIdx.x; t < n; t += 1024) { const double bt = b[t]; du += bt * u[t]; dv += bt * v[t]; }
  du = wsum64(du); dv = wsum64(dv);
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = du; s_red[1][threadIdx.x >> 6] = dv; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int w = 0; w < 16; ++w) { a += s_red[0][w]; c += s_red[1][w]; }
    out[0] = a; out[1] = c;
  }
}
// y = u - c v  (c by value, or where the ETA task of the factorisation left it)
__global__ void border_combine_kernel(double* __restrict__ y, const double* __restrict__ u, const double* __restrict__ v, double c, const double* c_ptr, int64_t n) {
#pragma clang fp contract(off)
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c_ptr) c = *c_ptr;
  if (t < n) y[t] = u[t] - c * v[t];
}
__global__ void ratio_init_kernel(double* rt, double ratio, double scale, double lb) { rt[kRtRatio] = ratio; rt[kRtRatioNew] = ratio; rt[kRtRatioEval] = ratio; rt[kRtScale] = scale; rt[kRtLb] = lb; }
// the ratio's scalars the ETA task reads, by value (the loop the host decides) ...
__global__ void ratio_prepare_kernel(double* rt, double diag_term, double gs, double scale) { rt[kRtDiagTerm] = diag_term; rt[kRtGs] = gs; rt[kRtScale] = scale; }
// ... or from the state on the device (the loop that runs without the host): LM diagonal clamp(s^2 h) as LevenbergMarquardtStrategy takes it for
// every column, the damped pivot s^2 h + D / radius, the scaled gradient s g — the host form's operations in the host form's order
__global__ void ratio_prepare_ctl_kernel(double* rt, const double* ctl, double lo, double hi) {
#pragma clang fp contract(off)
  if (ctl[kCtlStatus] != 0.0) return;
  const double sc = rt[kRtScale], h = rt[kRtH], g = rt[kRtG], radius = ctl[kCtlRadius];
  const double diag = fmin(fmax(sc * sc * h, lo), hi);
  rt[kRtDiag] = diag;
  rt[kRtDiagTerm] = sc * sc * h + diag / radius;
  rt[kRtGs] = sc * g;
}
// the ratio's candidate: projected onto its lower bound (ParameterBlock::Plus); what the candidate's prior blocks are evaluated with
__global__ void ratio_candidate_kernel(double* rt, const double* ctl) {
#pragma clang fp contract(off)
  if (ctl[kCtlStatus] != 0.0) return;
  const double ratio = rt[kRtRatio], lb = rt[kRtLb];
  const double rn = fmax(lb, ratio - rt[kRtC]);
  rt[kRtRatioNew] = rn;
  rt[kRtRatioEval] = isfinite(rn) ? rn : ratio;
}

}  // namespace

hipError_t launch_prior_border(const DeviceProblem& dp, const SolverDev& sv, double* border, double* hg, hipStream_t st) {
  hipLaunchKernelGGL(prior_border_kernel, dim3((dp.F + 63) / 64), dim3(64), 0, st, dp, sv, border, hg);
  return hipGetLastError();
}
hipError_t launch_border_dots(const double* b, const double* u, const double* v, int64_t n, double* out2, hipStream_t st) {
  hipLaunchKernelGGL(border_dots_kernel, dim3(1), dim3(1024), 0, st, b, u, v, n, out2);
  return hipGetLastError();
}
hipError_t launch_border_combine(double* y, const double* u, const double* v, double c, int64_t n, hipStream_t st, const double* c_ptr) {
  hipLaunchKernelGGL(border_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, u, v, c, c_ptr, n);
  return hipGetLastError();
}
hipError_t launch_ratio_init(double* rt, double ratio, double scale, double lb, hipStream_t st) {
  hipLaunchKernelGGL(ratio_init_kernel, dim3(1), dim3(1), 0, st, rt, ratio, scale, lb);
  return hipGetLastError();
}
hipError_t launch_ratio_prepare(double* rt, double diag_term, double gs, double scale, hipStream_t st) {
  hipLaunchKernelGGL(ratio_prepare_kernel, dim3(1), dim3(1), 0, st, rt, diag_term, gs, scale);
  return hipGetLastError();
}
hipError_t launch_ratio_prepare_ctl(double* rt, const double* ctl, double lo, double hi, hipStream_t st) {
  hipLaunchKernelGGL(ratio_prepare_ctl_kernel, dim3(1), dim3(1), 0, st, rt, ctl, lo, hi);
  return hipGetLastError();
}
hipError_t launch_ratio_candidate(double* rt, const double* ctl, hipStream_t st) {
  hipLaunchKernelGGL(ratio_candidate_kernel, dim3(1), dim3(1), 0, st, rt, ctl);
  return hipGetLastError();
}

hipError_t launch_prior_blocks(const DeviceProblem& dp, const SolverDev& sv, double* ucross, hipStream_t st) {
  hipLaunchKernelGGL(prior_blocks_kernel, dim3((dp.F + 63) / 64), dim3(64), 0, st, dp, sv, ucross);
  return hipGetLastError();
}
hipError_t launch_prior_cost(const DeviceProblem& dp, double* cost2, int invalid_blocks, hipStream_t st) {
  hipLaunchKernelGGL(prior_cost_kernel, dim3((dp.F + 63) / 64), dim3(64), 0, st, dp, cost2, invalid_blocks);
  return hipGetLastError();
}
hipError_t launch_prior_model(const DeviceProblem& dp, const SolverDev& sv, double* model_cost_change, double ratio_step, hipStream_t st, const double* ratio_step_ptr) {
  hipLaunchKernelGGL(prior_model_kernel, dim3((dp.F + 63) / 64), dim3(64), 0, st, dp, sv, model_cost_change, ratio_step, ratio_step_ptr);
  return hipGetLastError();
}

}  // namespace rsba
