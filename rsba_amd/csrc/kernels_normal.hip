// K2-K9 of SURVEY §2.1: normal-equation blocks, point elimination (Schur complement), back-substitution
// and the scalar reductions of the trust-region loop.  All fp64; all sums are taken in a fixed order
// (no floating-point atomics), so a solve is bit-reproducible from run to run.
//
// These replace, for rsba's BA problems, what Ceres-Solver 1.9's SchurEliminator / SchurComplementSolver
// and TrustRegionMinimizer compute on the CPU (call site
// /root/reference/src/rsba/CeresHandler.h:419).  They are HBM/latency-bound block operations on 12x12,
// 12x3 and 3x3 blocks — deliberately NOT reshaped into MFMA GEMMs: on gfx950 the fp64 MFMA rate equals
// the fp64 VALU rate, and padding 12 -> 16 would waste 44 % of it.
#include "lm_record.hpp"
#include "obs_math.hpp"
#include <type_traits>

#include "solver_state.hpp"
#include "test_hooks.hpp"
#include <cstdlib>

namespace rsba {

namespace {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// Camera-side coordinate t in [0, Fx*CD): the real pose coordinates, then the intrinsics pseudo frames
// (9 coordinates + zero-scaled padding) when the intrinsics are a parameter block.
// Intrinsics block c (shared sess.cam: the only one; per-frame f.cam blocks: one each, CeresHandler.h:260,277) rides as
// pseudo frames F + c * NPF + v; coordinate t behind the poses is coordinate k = v * CD + t % CD of block c (k >= 9: padding).
__device__ __forceinline__ int intr_index(const SolverDev& sv, int64_t u /* t - F*CD */) {
  const int blk = (int)(u / sv.CD), c = blk / sv.NPF, k = (blk % sv.NPF) * sv.CD + (int)(u % sv.CD);
  return k < 9 ? c * 9 + k : -1;
}
__device__ __forceinline__ double* cam_scale_ptr(const DeviceProblem& dp, const SolverDev& sv, int64_t t) {
  const int64_t npose = (int64_t)sv.F * sv.CD;
  if (t < npose) return dp.scale_pose + t;
  const int idx = intr_index(sv, t - npose);
  return idx >= 0 ? dp.scale_intr + idx : nullptr;
}
__device__ __forceinline__ double cam_scale(const DeviceProblem& dp, const SolverDev& sv, int64_t t) {
  const double* p = cam_scale_ptr(dp, sv, t);
  return p ? *p : 0.0;
}
__device__ __forceinline__ size_t u_cross_off(const SolverDev& sv, int v, int f) { return ((size_t)sv.F + (size_t)v * sv.F + f) * sv.CD * sv.CD; }
__device__ __forceinline__ size_t u_self_off(const SolverDev& sv, int c, int v, int w) { return ((size_t)sv.F + (size_t)sv.NPF * sv.F + ((size_t)c * sv.NPF + v) * sv.NPF + w) * sv.CD * sv.CD; }
__device__ __forceinline__ double u_diag(const SolverDev& sv, int64_t t) {
  const int f = (int)(t / sv.CD), a = (int)(t % sv.CD);
  const size_t base = f < sv.F ? (size_t)f * sv.CD * sv.CD : u_self_off(sv, (f - sv.F) / sv.NPF, (f - sv.F) % sv.NPF, (f - sv.F) % sv.NPF);
  return sv.U[base + (size_t)a * sv.CD + a];
}

// ---------------------------------------------------------------------------------------------
// K2a  per-frame camera block  U_f = sum Jc^T Jc,  g_f = sum Jc^T r  and, with intrinsics as a parameter block
// (opt.model.calibrated == false, shared sess.cam), the blocks of J^T J that are NOT block-diagonal:
//   cross: U[F+v][f] rows = intrinsics coordinates of pseudo frame v, cols = pose coordinates of frame f
//   self : per-frame partials of Ji^T Ji (45 unique) and Ji^T r (9), summed over frames by intr_reduce_kernel.
// The products themselves are formed by the evaluation kernel (kernels_eval.hip): every wave leaves, per frame it touches, the one or two
// 16 x 16 blocks that hold everything on and below the diagonal of G = [Ji | Jc | r]^T [Ji | Jc | r] (device_state.hpp: cam_part_blocks /
// cam_part_entry).  Here one workgroup per frame sums its waves' partials in wave order (fixed order: deterministic) and files the entries of G.
// ---------------------------------------------------------------------------------------------
// lm_take_candidate_kernel's copy, by workgroup `block` of a launch that carries it along
__device__ __forceinline__ void take_candidate_block(const DeviceProblem& dp, const SolverDev& sv, int64_t block) {
  const int64_t t = block * 256 + threadIdx.x, npose = (int64_t)dp.F * dp.P * 6, npoint = 3 * (int64_t)dp.M, nintr = sv.NPF > 0 ? (int64_t)dp.NI * 9 : 0;
  if (t < npose) dp.poses[t] = sv.trial_poses[t];
  else if (t < npose + npoint) dp.points[t - npose] = sv.trial_points[t - npose];
  else if (t < npose + npoint + nintr) dp.intr[t - npose - npoint] = sv.trial_intr[t - npose - npoint];
}
template <int CD, bool CAL>
__device__ __forceinline__ void camera_reduce_frame(const DeviceProblem& dp, const SolverDev& sv, const int f) {
  constexpr int NI = CAL ? 0 : 9, NCOL = NI + CD + 1, NBLK = cam_part_blocks(NCOL);   // (device_state.hpp: where the evaluation kernel leaves which entry of G)
  __shared__ double G[NBLK][256];
  const int e = threadIdx.x;
  const int64_t s0 = sv.frame_ptr[f], s1 = sv.frame_ptr[f + 1];
  double sum[NBLK];
#pragma unroll
  for (int q = 0; q < NBLK; ++q) sum[q] = 0.0;
  if (s1 > s0) {
    // where each of the frame's waves left its partial: looked up by a thread per wave first (three dependent loads), so that the sums
    // below — in wave order, as ever — issue their loads back to back instead of behind that chain
    __shared__ int s_seg[256];
    const int rk = dp.frame_rank[f];
    const int64_t w0 = s0 >> 6, w1 = (s1 - 1) >> 6;
    for (int64_t wb = w0; wb <= w1; wb += 256) {
      const int nw = (int)(w1 - wb + 1 < 256 ? w1 - wb + 1 : 256);
      __syncthreads();
      if (e < nw) s_seg[e] = dp.wave_seg_base[wb + e] + rk - dp.frame_rank[dp.obs_frame[(wb + e) << 6]];
      __syncthreads();
      int i = 0;
      for (; i + 4 <= nw; i += 4) {
        double v[4][NBLK];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < NBLK; ++q) v[u][q] = dp.cam_part[((size_t)s_seg[i + u] * NBLK + q) * 256 + e];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < NBLK; ++q) sum[q] += v[u][q];
      }
      for (; i < nw; ++i)
#pragma unroll
        for (int q = 0; q < NBLK; ++q) sum[q] += dp.cam_part[((size_t)s_seg[i] * NBLK + q) * 256 + e];
    }
  }
#pragma unroll
  for (int q = 0; q < NBLK; ++q) G[q][e] = sum[q];
  __syncthreads();
  auto g = [&](int a, int b) {   // entry (a, b) of the symmetric G
    if (a < b) { const int t = a; a = b; b = t; }
    return NBLK == 1 ? G[0][a * 16 + b] : (&G[0][0])[cam_part_entry(a, b)];
  };
  for (int idx = e; idx < CD * CD; idx += 256) sv.U[(size_t)f * CD * CD + idx] = g(NI + idx / CD, NI + idx % CD);
  if (e < CD) sv.gc[(size_t)f * CD + e] = g(NI + CD, NI + e);
  if (!CAL) {
    for (int idx = e; idx < 9 * CD; idx += 256) {
      const int kk = idx / CD, c = idx % CD;
      sv.U[u_cross_off(sv, kk / CD, f) + (size_t)(kk % CD) * CD + c] = g(NI + c, kk);
    }
    if (e < 54) {
      double v;
      if (e >= 45) v = g(NI + CD, e - 45);
      else { int a = 0, rem = e; while (rem >= 9 - a) { rem -= 9 - a; ++a; } v = g(a + rem, a); }
      sv.intr_part[(size_t)f * 54 + e] = v;
    }
  }
}
template <int CD, bool CAL>
__global__ __launch_bounds__(256) void camera_reduce_kernel(const DeviceProblem dp, const SolverDev sv) {
  if (lm_not_accepted(sv.ctl)) return;   // (device-side trust region: a rejected candidate is not linearised)
  if ((int)blockIdx.x >= dp.F) { take_candidate_block(dp, sv, (int64_t)blockIdx.x - dp.F); return; }   // workgroups behind the frames' (launch_camera_blocks with take_candidate) — this kernel reads no parameters
  camera_reduce_frame<CD, CAL>(dp, sv, (int)blockIdx.x);
}

// one workgroup per (intrinsics block c, entry t of the 45 + 9 sums): lanes stride the frames that use the block (in
// frame order), fixed-order wave / workgroup reduction
__global__ __launch_bounds__(256) void intr_reduce_kernel(const DeviceProblem dp, const SolverDev sv) {
  if (lm_not_accepted(sv.ctl)) return;
  __shared__ double s_red[4];
  const int c = blockIdx.x / 54, t = blockIdx.x % 54, tid = threadIdx.x;
  double v = 0.0;
  for (int q = sv.intr_frame_ptr[c] + tid; q < sv.intr_frame_ptr[c + 1]; q += 256) v += sv.intr_part[(size_t)sv.intr_frame_list[q] * 54 + t];
  v = wsum(v);
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  if (tid != 0) return;
  v = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  const int CD = sv.CD;
  if (t >= 45) { sv.gc[((size_t)sv.F + (size_t)c * sv.NPF) * CD + (t - 45)] = v; return; }
  int a = 0, rem = t;
  while (rem >= 9 - a) { rem -= 9 - a; ++a; }
  const int b = a + rem;
  sv.U[u_self_off(sv, c, a / CD, b / CD) + (size_t)(a % CD) * CD + (b % CD)] = v;
  sv.U[u_self_off(sv, c, b / CD, a / CD) + (size_t)(b % CD) * CD + (a % CD)] = v;
}

// where a slot's P record goes (solver_state.hpp: slot_gpos = element offset of its group | position of its frame in the tile << 1 | kind)
__device__ __forceinline__ size_t gpos_group(uint32_t gpos) { return (size_t)(gpos & ~15u); }
__device__ __forceinline__ int gpos_pos(uint32_t gpos) { return (int)((gpos >> 1) & 7u); }
__device__ __forceinline__ bool gpos_factored(uint32_t gpos) { return (gpos & 1u) != 0; }

// virtual observation records of the pseudo frames: one group per (point j, intrinsics block c the point is seen through),
// Q_j,c = sum_{o of j in frames that use c} Ji_o^T (Jp_o L_j^-T)   (9 x 3), cut into the NPF pseudo-frame records of the group
template <int CD>
__global__ __launch_bounds__(256) void virtual_records_kernel(const DeviceProblem dp, const SolverDev sv) {
  if (lm_stopped(sv.ctl)) return;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= sv.nvgroups) return;
  const int j = sv.vgroup_point[g], c = sv.vgroup_intr[g];
  const int REC = 2 + 2 * dp.K, KC = dp.K - 3;
  const double* li = sv.Linv + (size_t)j * 6;
  const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
  double Q[9][3];
#pragma unroll
  for (int k = 0; k < 9; ++k) { Q[k][0] = 0.0; Q[k][1] = 0.0; Q[k][2] = 0.0; }
  for (int64_t s = sv.point_ptr[j]; s < sv.point_ptr[j + 1]; ++s) {
    if (sv.NIB > 1 && dp.frame_intr[sv.slot_frame[s]] != c) continue;
    const double* rec = lm_records(dp, false) + (size_t)s * REC;
    double B[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double p0 = rec[2 + 3 * r], p1 = rec[3 + 3 * r], p2 = rec[4 + 3 * r];
      B[r][0] = p0 * i00; B[r][1] = p0 * i10 + p1 * i11; B[r][2] = p0 * i20 + p1 * i21 + p2 * i22;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double a0 = rec[8 + k], a1 = rec[8 + KC + k];
#pragma unroll
      for (int m = 0; m < 3; ++m) Q[k][m] += a0 * B[0][m] + a1 * B[1][m];
    }
  }
  for (int v = 0; v < sv.NPF; ++v) {
    const uint32_t gpos = sv.slot_gpos[dp.N + g * sv.NPF + v];   // the virtual slot's place: its group (always full form: a pseudo frame's tile), frame position of its tile
    double* out = sv.Pm + gpos_group(gpos) + gpos_pos(gpos) * CD;
#pragma unroll
    for (int rl = 0; rl < CD; ++rl) {
      const int k = v * CD + rl;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double q = 0.0;
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) if (kk == k) q = Q[kk][m];
        out[m * kTile + rl] = q;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K2b  per-point block  V_j = sum Jp^T Jp (6 unique),  g_p,j = sum Jp^T r   (point-major records)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void point_blocks_kernel(const DeviceProblem dp, const SolverDev sv) {
  if (lm_not_accepted(sv.ctl)) return;   // (device-side trust region: a rejected candidate is not linearised)
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= dp.M) return;
  const int REC = 2 + 2 * dp.K;
  double v[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for (int64_t s = sv.point_ptr[j]; s < sv.point_ptr[j + 1]; ++s) {
    const double2* rp = reinterpret_cast<const double2*>(lm_records(dp, false) + (size_t)s * REC);
    const double2 a = rp[0], b = rp[1], c = rp[2], d = rp[3];   // r0 r1 | p00 p01 | p02 p10 | p11 p12
    const double r0 = a.x, r1 = a.y, p0[3] = {b.x, b.y, c.x}, p1[3] = {c.y, d.x, d.y};
    v[0] += p0[0] * p0[0] + p1[0] * p1[0]; v[1] += p0[0] * p0[1] + p1[0] * p1[1]; v[2] += p0[0] * p0[2] + p1[0] * p1[2];
    v[3] += p0[1] * p0[1] + p1[1] * p1[1]; v[4] += p0[1] * p0[2] + p1[1] * p1[2]; v[5] += p0[2] * p0[2] + p1[2] * p1[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] += p0[k] * r0 + p1[k] * r1;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) sv.V[(size_t)j * 6 + k] = v[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) sv.gp[(size_t)j * 3 + k] = g[k];
}

// Ceres 1.9 TrustRegionMinimizer: EstimateScale  scale_i = 1 / (1 + sqrt(|J_i|^2)), once, from the first
// Jacobian.  dp.scale_* holds the 0/1 mask at that moment.
__global__ void jacobi_scale_kernel(const DeviceProblem dp, const SolverDev sv) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, nc = sv.n;
  if (t < nc) {
    double* sp = cam_scale_ptr(dp, sv, t);
    if (sp) *sp *= 1.0 / (1.0 + sqrt(sv.udiag[t]));
  } else if (t < nc + 3 * (int64_t)dp.M) {
    const int64_t u = t - nc; const int j = (int)(u / 3), a = (int)(u % 3);
    const int dg = (a == 0) ? 0 : (a == 1 ? 3 : 5);
    dp.scale_point[u] *= 1.0 / (1.0 + sqrt(sv.V[(size_t)j * 6 + dg]));
  }
}

// LevenbergMarquardtStrategy::ComputeStep: diagonal_ = clamp(|J_i|^2, min_lm_diagonal, max_lm_diagonal)
__global__ void clamp_diagonal_kernel(const DeviceProblem dp, const SolverDev sv, double lo, double hi) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, nc = sv.n;
  if (t < nc) {
    sv.diag_c[t] = fmin(fmax(sv.udiag[t], lo), hi);
  } else if (t < nc + 3 * (int64_t)dp.M) {
    const int64_t u = t - nc; const int j = (int)(u / 3), a = (int)(u % 3);
    const int dg = (a == 0) ? 0 : (a == 1 ? 3 : 5);
    sv.diag_p[u] = fmin(fmax(sv.V[(size_t)j * 6 + dg], lo), hi);
  }
}

// max |g_i| of the UNSCALED gradient (Ceres evaluates the gradient before ScaleColumns): g = g_scaled / scale
// part: 0 = every coordinate, 1 = the points' only, 2 = the cameras' only (several ranks: a rank's own points before the camera exchange —
// their maximum travels in it —, the cameras' from the summed gradient behind it)
__global__ __launch_bounds__(256) void gradient_max_kernel(const DeviceProblem dp, const SolverDev sv, int part) {
  __shared__ double s_red[4];
  double m = 0.0;
  const int64_t nc = sv.n, np = 3 * (int64_t)dp.M;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x + (part == 1 ? nc : 0);
  if (t < (part == 2 ? nc : nc + np)) {
    const double sc = (t < nc) ? cam_scale(dp, sv, t) : dp.scale_point[t - nc];
    const double g = (t < nc) ? sv.gc[t] : sv.gp[t - nc];
    if (sc > 0.0) m = fabs(g / sc);
  }
  m = wmax(m);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) sv.partial[blockIdx.x] = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
}
// (+ `nextra` more values at `extra`: the other ranks' maxima that came with the camera exchange)
__global__ __launch_bounds__(256) void reduce_max_kernel(const double* partial, int n, double* out, const double* extra = nullptr, int nextra = 0) {
  __shared__ double s_red[4];
  double v = 0.0;
  for (int k = threadIdx.x; k < n; k += 256) v = fmax(v, partial[k]);
  for (int k = threadIdx.x; k < nextra; k += 256) v = fmax(v, extra[k]);
  v = wmax(v);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) *out = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
}

__global__ void unscaled_gradient_kernel(const DeviceProblem dp, const SolverDev sv, double* g_pose, double* g_point) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, nc = sv.n;
  if (t < nc) { const double sc = cam_scale(dp, sv, t); g_pose[t] = sc > 0.0 ? sv.gc[t] / sc : 0.0; }
  else if (t < nc + 3 * (int64_t)dp.M) { const int64_t u = t - nc; const double sc = dp.scale_point[u]; g_point[u] = sc > 0.0 ? sv.gp[u] / sc : 0.0; }
}

// ---------------------------------------------------------------------------------------------
// K5a  per point: V' = V + D_p^2 (D^2 = diagonal_/radius), 3x3 Cholesky, L^-1, z = L^-1 g_p
// (SchurEliminator::Eliminate inverts each e-block)
// ---------------------------------------------------------------------------------------------
// CLAMP: clamp_diagonal_kernel's job done on the way (the loop that runs without the host recomputes the diagonal every iteration —
// after a rejected step that is what is there already — and saves the launch): the point's own three entries by its thread, the
// camera side by the first sv.n threads of the grid.
template <bool CLAMP>
__global__ __launch_bounds__(256) void point_factor_kernel(const DeviceProblem dp, const SolverDev sv, double inv_radius, double lo, double hi) {
  if (sv.ctl) inv_radius = 1.0 / sv.ctl[kCtlRadius];   // (device-side trust region: the radius lives in HBM)
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (CLAMP && j < sv.n) sv.diag_c[j] = fmin(fmax(sv.udiag[j], lo), hi);
  if (j >= dp.M) return;
  const double* v = sv.V + (size_t)j * 6;
  double dg[3];
  if (CLAMP) {
    dg[0] = fmin(fmax(v[0], lo), hi); dg[1] = fmin(fmax(v[3], lo), hi); dg[2] = fmin(fmax(v[5], lo), hi);
    sv.diag_p[(size_t)j * 3] = dg[0]; sv.diag_p[(size_t)j * 3 + 1] = dg[1]; sv.diag_p[(size_t)j * 3 + 2] = dg[2];
  } else { dg[0] = sv.diag_p[(size_t)j * 3]; dg[1] = sv.diag_p[(size_t)j * 3 + 1]; dg[2] = sv.diag_p[(size_t)j * 3 + 2]; }
  const double a00 = v[0] + dg[0] * inv_radius, a10 = v[1], a20 = v[2], a11 = v[3] + dg[1] * inv_radius, a21 = v[4], a22 = v[5] + dg[2] * inv_radius;
  const double l00 = sqrt(a00), l10 = a10 / l00, l20 = a20 / l00;
  const double d11 = a11 - l10 * l10, l11 = sqrt(d11), l21 = (a21 - l20 * l10) / l11;
  const double d22 = a22 - l20 * l20 - l21 * l21, l22 = sqrt(d22);
  if (!(a00 > 0.0) || !(d11 > 0.0) || !(d22 > 0.0) || !isfinite(l22)) atomicExch(sv.chol_fail, 1);
  const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
  const double i10 = -l10 * i00 * i11, i21 = -l21 * i11 * i22, i20 = -(l20 * i00 + l21 * i10) * i22;
  double* li = sv.Linv + (size_t)j * 6;
  li[0] = i00; li[1] = i10; li[2] = i11; li[3] = i20; li[4] = i21; li[5] = i22;
  const double* g = sv.gp + (size_t)j * 3;
  double* z = sv.z + (size_t)j * 3;
  z[0] = i00 * g[0]; z[1] = i10 * g[0] + i11 * g[1]; z[2] = i20 * g[0] + i21 * g[1] + i22 * g[2];
}

// K5b  per observation (point-major): P = Jc^T (Jp L^-T)   (CD x 3)
// One wave = 64 consecutive slots.  Records are moved from HBM to LDS with fully coalesced 512-B wave
// accesses (a lane-per-record global access pattern touches 64 cache lines per instruction and ran 4x
// slower); each lane then works on its own record out of LDS (odd pitch: no bank conflicts).
// The results go to the GROUP layout the Schur kernel reads (solver_state.hpp, Pm): coordinate c of the point
// against the CD rows of the slot's frame is a run of CD doubles at row c of the slot's (point, tile) group,
// position (frame % FT) * CD.  Consecutive slots of a point are consecutive frames, so the runs of one
// coordinate line up back to back: the wave stores coordinate by coordinate, each store instruction covering
// (mostly) whole 128-B lines.
constexpr int kProjectChunks = 2;   // consecutive 64-slot chunks per wave of the projection kernel at most (one when the scene is small: project_chunks).  Round 6: 8 until then — swept again on the slim (all-factored, three workgroups per CU) form: C4 project phase 0.096 / 0.096 / 0.101 / 0.112 / 0.113 / 0.150 ms for 1 / 2 / 4 / 8 / 16 / 32
inline int project_chunks(int64_t N) {   // ~2 k waves or more
  static const int forced = [] { const char* e = std::getenv("RSBA_PROJECT_CHUNKS"); const int v = e ? std::atoi(e) : 0; return v >= 1 && v <= 64 ? v : 0; }();   // (tuning aid)
  const int64_t c = N / 64 / 2048;
  return forced ? forced : (int)(c < 1 ? 1 : c > kProjectChunks ? kProjectChunks : c);
}

template <int CD, int KC>
__global__ __launch_bounds__(256) void project_kernel(const DeviceProblem dp, const SolverDev sv) {
  if (lm_stopped(sv.ctl)) return;
  constexpr int REC = 8 + 2 * KC, OUT = CD * 3;
  constexpr int PITCH = (REC > OUT ? REC : OUT) | 1;          // odd
  constexpr int off = KC - CD;                                  // 9 when intrinsics columns precede the pose
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* buf = smem + (size_t)wave * (64 * PITCH + 32);
  uint32_t* s_gpos = reinterpret_cast<uint32_t*>(buf + 64 * PITCH);   // [64] where each slot of the chunk goes (problems that keep records store every group in full form)
  const int64_t sb = ((int64_t)blockIdx.x * 4 + wave) * 64 * kProjectChunks;
  if (sb >= dp.N) return;
  const int64_t se = sb + 64 * kProjectChunks < dp.N ? sb + 64 * kProjectChunks : dp.N;
  // the next chunk's records (and the point of each slot) travel in registers while this one is worked on; indices
  // are clamped rather than predicated so that nothing next to the loads waits for them
  double pre[REC];
  int pre_point = 0; uint32_t pre_gpos = 0;
  auto issue = [&](int64_t c0) {
    const int64_t last = (se - c0) * REC - 1;
    const double* src = lm_records(dp, false) + (size_t)c0 * REC;
#pragma unroll
    for (int k = 0; k < REC; ++k) { const int64_t idx = k * 64 + lane; pre[k] = src[idx < last ? idx : last]; }
    const int64_t sl = c0 + lane < se ? c0 + lane : se - 1;
    pre_point = sv.slot_point[sl]; pre_gpos = sv.slot_gpos[sl];
  };
  issue(sb);
  for (int64_t s0 = sb; s0 < se; s0 += 64) {
    const int64_t nslot = se - s0 < 64 ? se - s0 : 64;
#pragma unroll
    for (int k = 0; k < REC; ++k) { const int idx = k * 64 + lane; buf[(idx / REC) * PITCH + idx % REC] = pre[k]; }
    const double* li = sv.Linv + (size_t)pre_point * 6;
    const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
    s_gpos[lane] = pre_gpos;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (s0 + 64 < se) issue(s0 + 64);
    double out[OUT];
    {
      const double* rec = buf + lane * PITCH;   // lanes past nslot work on stale LDS; their results are not stored
      double B[2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double p0 = rec[2 + 3 * r], p1 = rec[3 + 3 * r], p2 = rec[4 + 3 * r];
        B[r][0] = p0 * i00; B[r][1] = p0 * i10 + p1 * i11; B[r][2] = p0 * i20 + p1 * i21 + p2 * i22;
      }
#pragma unroll
      for (int a = 0; a < CD; ++a) {
        const double c0 = rec[8 + off + a], c1 = rec[8 + KC + off + a];
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k * CD + a] = c0 * B[0][k] + c1 * B[1][k];   // component-major inside the record
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < OUT; ++e) buf[lane * PITCH + e] = out[e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Stores: a lane owns one double (w) of the CD-long runs; kPer = 64 / CD consecutive slots per instruction (their runs of one
    // coordinate sit back to back while the slots stay inside one group), three instructions — one per coordinate, 48 doubles
    // apart: the store's immediate offset — per address computation.  (64 % CD lanes idle.)
    constexpr int kPer = 64 / CD;
    const int my = lane / CD, w = lane % CD;
    if (my < kPer) {
#pragma unroll 2
      for (int i = 0; i * kPer < 64; ++i) {
        const int sl = i * kPer + my;
        if (sl < nslot) {
          const uint32_t gpos = s_gpos[sl];
          double* dst = sv.Pm + (gpos_group(gpos) + (size_t)(gpos_pos(gpos) * CD + w));
          const double* src = buf + sl * PITCH + w;
#pragma unroll
          for (int comp = 0; comp < 3; ++comp) dst[comp * kTile] = src[comp * CD];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---------------------------------------------------------------------------------------------
// K5c  reduced camera system  S = U + D_c^2 - sum_j (sum_a P_aj)(sum_b P_bj)^T ,  rhs = g_c - sum P z
// Work unit = ENTRY (point j, frame tiles I >= J): the point's P records in the FT frames of I and of J, stacked
// into A_j(I) and A_j(J) (48 x 3 each; a frame that does not see the point contributes zero rows) — exactly the
// (point, tile) GROUPS the projection kernel writes: group g is a [3][48] block of Pm, coordinate-major.
// The tile of the pair is the banded SYRK  S_IJ = sum_j A_j(I) A_j(J)^T  — GEMM-shaped, K = 3 per point — and runs
// on v_mfma_f64_16x16x4_f64 (four points fill three MFMA steps of k = 4), operands loaded from HBM/L2 straight
// into the instruction's register layout: lane (r = lane & 15, g = lane >> 4) holds P[row 16 Ib + r][one
// coordinate of one point] for the three 16-row blocks Ib of each side — in the group layout the sixteen lanes of
// a lane group read ONE aligned 128-B line, a wave instruction four of them (round 2 gathered the same sixteen
// values from slot-major records at a 24-B stride: three to four lines per lane group, and the kernel was bound by
// those gathers as much as by the matrix pipe).  No LDS, no barrier in the loop.
// 16 x 16 operand blocks that are all zero for the four points of an MFMA step — frames of the tile that do not see them —
// are found with one wave vote per block and their MFMAs are not issued (a fifth of them at 1k cameras; 0 * x adds nothing,
// so the result is the same to the bit), and a tile paired with itself only forms the blocks on and below the
// diagonal (nothing reads the upper triangle of a diagonal tile of S).
// One chunk of kSchurChunk entries at a time per workgroup (resident workgroups take chunk after chunk: schur_tile_kernel); its four waves take every fourth entry and keep all
// nine 16x16 blocks (loads run kDepth groups ahead of the MFMAs in a register ring), the four partial tiles meet
// once in LDS.  The fp64 VALU form of this product was bound by LDS operand reads at 2.0 ms per 1k-camera
// iteration.  Chunks write partial tiles; the merge kernel sums them in chunk order (fixed order, no atomics)
// and adds U, D_c^2, g_c.
// ---------------------------------------------------------------------------------------------
typedef double dbl4 __attribute__((ext_vector_type(4)));

// LDS of the loop (bytes): [2 kSchurChunk] uint32 element offsets of the entries' two groups | [kSchurChunk] uint16 block masks | (diagonal
// pairs) [3 kSchurChunk] doubles z.  The four partial tiles of the epilogue reuse it from the start.
constexpr int kSchurOffBytes = 2 * kSchurChunk * 4, kSchurMskBytes = kSchurChunk * 2;
constexpr unsigned kLowerBlocks = 0x1D9;   // bits 3 I + J with J <= I

// One chunk.  DIAG: a tile paired with itself — only the blocks on and below the diagonal are formed, and its entries carry
// the rhs term P z.  The loop is written for the instruction issue port: fp64 MFMAs run on the vector unit's fp64 datapath,
// so every vector / scalar instruction around them is time the matrix pipe idles (round 2's loop spent 6 scalar and 5 vector
// instructions per MFMA on 64-bit gather addresses, tail selects and per-step votes: the pipe was busy 39 % of the time,
// SQ_VALU_MFMA_BUSY_CYCLES).  Here: entry counts are padded to a multiple of 16 with entries that point at an all-zero group
// (no tail selects), the table holds ready element offsets (one 64-bit add per operand, the blocks ride on the load's immediate
// offset), and which of the nine 16 x 16 blocks of a group of four entries have anything to multiply
// is ONE scalar read of host-computed masks; every MFMA sits behind a scalar test of its block's bit (untaken for a full mask, the common case).
//
// FA / FB: the groups of the I / J side are stored FACTORED (solver_state.hpp: kGroupFactored; round 5).  The 48 camera-side rows of
// a two-pose frame tile are, frame by frame, (1 - tau) q | tau q with q = Jq^T Jp L^-T (6 x 3): such a group holds the 24 "sources"
// q per coordinate and the four tau, 640 B instead of 1152, and the operand rows are formed here — one multiply per operand double.
// The rows of a factored side are taken in an order of this kernel's own (the epilogue puts every result where it belongs, so
// nothing outside sees it): blocks 0 and 1 are the pose-0 and pose-1 rows of sources 0..15 — ONE loaded double feeds both — and block
// 2 the rows of sources 16..23 (lanes 0..7 pose 0, lanes 8..15 pose 1).  Per lane and group of four entries: 6 source doubles + 2 tau
// per side instead of 9 operand doubles.  The column scales (Jacobi scales, masks of fixed coordinates) factor out of the sum over
// the points: they are applied once, where the partial tiles are merged.
struct FactoredLane {   // per-lane constants of the row order above
  int main, left, tau_m, tau_l;   // element offsets inside a group (coordinate 0; + 16 / + 8 per coordinate)
  double a0, b0, a1, b1, a2, b2;  // weight of block b = a_b + b_b * tau
};
__device__ __forceinline__ FactoredLane factored_lane(int r, bool lerp_rot) {
  FactoredLane f;
  const int sl = 16 + (r & 7);
  f.main = r; f.left = 48 + (r & 7); f.tau_m = 72 + r / 6; f.tau_l = 72 + sl / 6;
  const bool rot_m = (r % 6) < 3 && !lerp_rot, rot_l = (sl % 6) < 3 && !lerp_rot;   // rotation rows without interpolateRotation: pose 0 carries them whole, pose 1 nothing (cam.h:303-304)
  f.a0 = 1.0; f.b0 = rot_m ? 0.0 : -1.0;
  f.a1 = 0.0; f.b1 = rot_m ? 0.0 : 1.0;
  const bool p1 = (r >> 3) != 0;
  f.a2 = p1 ? 0.0 : 1.0; f.b2 = rot_l ? 0.0 : (p1 ? 1.0 : -1.0);
  return f;
}
// the tile row of operand position (block Ib, row i of the block) of a factored side
__device__ __forceinline__ int factored_row(int Ib, int i) {
  const int s = Ib < 2 ? i : 16 + (i & 7), p = Ib == 0 ? 0 : Ib == 1 ? 1 : (i >> 3);
  return 12 * (s / 6) + 6 * p + s % 6;
}
template <bool F> struct SchurSide;
template <> struct SchurSide<true> { double qm[3], ql[3], tm, tl; };   // [coordinate]
template <> struct SchurSide<false> { double v[3][3]; };               // [coordinate][block]

// The table cells of a chunk that one thread stages (entries tid and tid + 256), loaded ahead of the chunk: the persistent form of the
// kernel issues these loads for its NEXT chunk before the epilogue of the current one, so that the dependent chain chunk -> entry list ->
// operands is not paid chunk by chunk (one workgroup per chunk spent 16 % of a CU slot's time between the end of one chunk and the
// first MFMA of the next: dispatch, four dependent reads; profiles/r05/schur_persistent.txt).
struct ChunkStage { uint32_t ga[2], gb[2]; unsigned pm[2]; int32_t pt[2]; };
static_assert(kSchurChunk == 512, "two table cells per thread");
__device__ __forceinline__ void stage_chunk(const SolverDev& sv, const int4& info, int tid, ChunkStage& st) {
  const int64_t e0 = (int64_t)(((uint64_t)(uint32_t)info.y << 32) | (uint32_t)info.x);
  const int n = info.z;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = tid + 256 * u;
    st.ga[u] = st.gb[u] = sv.zero_off;      // (behind the chunk's last entry: the all-zero group behind the last one)
    st.pm[u] = 0; st.pt[u] = 0;
    if (k < n) {
      const uint2 gg = *reinterpret_cast<const uint2*>(sv.ent_groups + 2 * (e0 + k));
      st.ga[u] = gg.x & ~15u; st.gb[u] = gg.y & ~15u;   // (the kind bit: the same for every entry of a tile pair — FA, FB)
      st.pm[u] = sv.ent_mask[e0 + k];     // 16 x 16 blocks of the entry with a frame that sees the point on both sides (host)
      if (info.w & 1) st.pt[u] = sv.ent_pt[e0 + k];
    }
  }
}

// -> the workgroup's next chunk (-1: none), `info` / `st` then hold that chunk's
template <bool DIAG, int kDepth, bool FA, bool FB>
__device__ __forceinline__ int schur_chunk(const SolverDev& sv, const double* __restrict__ Pm, const double* __restrict__ zz, int chunk, int4& info, ChunkStage& st, double* smem, int* s_next,
                                           bool persistent, int xcd, int per_xcd) {
  constexpr int TPITCH = kTile + 1;
  constexpr unsigned kFull = DIAG ? kLowerBlocks : 0x1FFu;
  uint32_t* s_off = reinterpret_cast<uint32_t*>(smem);
  uint16_t* s_msk = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(smem) + kSchurOffBytes);   // [(k & 3) * 128 + (k >> 2)]: the four entries of a wave's group are one 8-byte read
  double* s_z = reinterpret_cast<double*>(reinterpret_cast<char*>(smem) + kSchurOffBytes + kSchurMskBytes);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n = info.z, n16 = (n + 15) & ~15;
  long long* tr = sv.schur_trace ? sv.schur_trace + 8 * (size_t)chunk : nullptr;
  if (tr && tid == 0) { tr[0] = blockIdx.x; tr[1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); tr[2] = wall_clock64(); tr[6] = n; }   // HW_REG_HW_ID
  __syncthreads();   // (the previous chunk's epilogue has read its partial tiles: the same LDS)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = tid + 256 * u;
    if (k < n16) {
      uint32_t ga = st.ga[u], gb = st.gb[u];
      if (kTestHooks && sv.schur_variant == 5 && k < n) { ga = (uint32_t)(k & 63) * kGroupFull; gb = (uint32_t)(64 + (k & 63)) * kGroupFull; }   // ablation: operands out of the caches
      s_off[2 * k] = ga; s_off[2 * k + 1] = gb;
      s_msk[(k & 3) * (kSchurChunk / 4) + (k >> 2)] = (uint16_t)(st.pm[u] & kFull);
      if (DIAG) {
        const int32_t pt = st.pt[u];
#pragma unroll
        for (int c = 0; c < 3; ++c) s_z[3 * k + c] = pt < 0 ? zz[(size_t)(pt & 0x7fffffff) * 3 + c] : 0.0;   // top bit: diagonal entry of the point -> rhs term P z
      }
    }
  }
  __syncthreads();
  if (tr && tid == 0) { tr[3] = wall_clock64(); tr[7] = -clock64(); }
  // the workgroup's next chunk: the next one of its XCD's eighth of the list that nobody has taken (asked for now, needed behind the loop)
  unsigned ticket = 0;
  if (persistent && tid == 0) ticket = __hip_atomic_fetch_add(sv.schur_next + 16 * xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // K = 3 per point against 4 per MFMA: four of the wave's entries share three MFMA steps — step t takes coordinate t of the four,
  // lane group g holding entry g of them (one table cell per lane and group of four).  The wave's entries are wave, wave + 4, ..:
  // entry e of its group q is k = wave + 16 q + 4 e.
  const int tab = 8 * (wave + 4 * g), zof = 3 * (wave + 4 * g);
  const FactoredLane fl = factored_lane(r, sv.lerp_rot != 0);
  dbl4 acc[3][3];
#pragma unroll
  for (int I = 0; I < 3; ++I)
#pragma unroll
    for (int J = 0; J < 3; ++J) acc[I][J] = dbl4{0.0, 0.0, 0.0, 0.0};
  double racc[3] = {0.0, 0.0, 0.0};
  struct Group { SchurSide<FA> a; SchurSide<FB> b; double z[3]; };
  Group ring[kDepth];
  const int nq = n16 >> 4;   // groups of four entries per wave
  const char* lds = reinterpret_cast<const char*>(smem);
  auto fetch_side = [&](const double* p, auto& S) {
    using T = typename std::remove_reference<decltype(S)>::type;
    if constexpr (std::is_same<T, SchurSide<true>>::value) {
      S.tm = p[fl.tau_m]; S.tl = p[fl.tau_l];
#pragma unroll
      for (int t = 0; t < 3; ++t) { S.qm[t] = p[fl.main + 16 * t]; S.ql[t] = p[fl.left + 8 * t]; }
    } else {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int Ib = 0; Ib < 3; ++Ib) S.v[t][Ib] = p[t * kTile + 16 * Ib + r];
    }
  };
  auto fetch = [&](int q, Group& G) {   // (q is wave-uniform; past the end the last group is read again instead of branching)
    const int qq = q < nq ? q : nq - 1;
    const uint2 o = *reinterpret_cast<const uint2*>(lds + tab + 128 * qq);
    fetch_side(Pm + (size_t)o.x, G.a);
    fetch_side(Pm + (size_t)o.y, G.b);
    if (DIAG) {
#pragma unroll
      for (int t = 0; t < 3; ++t) G.z[t] = s_z[zof + t + 48 * qq];
    }
  };
  // the three operand doubles of coordinate t (blocks 0, 1, 2)
  auto operands = [&](const auto& S, int t, double w0, double w1, double w2, double out[3]) {
    using T = typename std::remove_const<typename std::remove_reference<decltype(S)>::type>::type;
    if constexpr (std::is_same<T, SchurSide<true>>::value) { out[0] = S.qm[t] * w0; out[1] = S.qm[t] * w1; out[2] = S.ql[t] * w2; }
    else { out[0] = S.v[t][0]; out[1] = S.v[t][1]; out[2] = S.v[t][2]; }
  };
  unsigned issued = 0;
  if (nq > 0) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) fetch(d, ring[d]);
    for (int base = 0; base < nq; base += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        if (base + d < nq) {
          // blocks with something to multiply: OR of the four entries' masks, one 8-byte LDS read at a wave-uniform address
          const uint2 m4 = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(s_msk) + 2 * (wave * (kSchurChunk / 4) + 4 * (base + d)));
          unsigned mv = m4.x | m4.y; mv = (mv | (mv >> 16)) & 0x1FFu;
          unsigned pm = (unsigned)__builtin_amdgcn_readfirstlane((int)mv);
          const Group& G = ring[d];
          // the weights of a factored side's three blocks for this lane's entry: a + b tau (1 - tau and tau; 1 and 0 for rotation rows without interpolateRotation)
          double wa0 = 0, wa1 = 0, wa2 = 0, wb0 = 0, wb1 = 0, wb2 = 0;
          if constexpr (FA) { wa0 = __builtin_fma(fl.b0, G.a.tm, fl.a0); wa1 = __builtin_fma(fl.b1, G.a.tm, fl.a1); wa2 = __builtin_fma(fl.b2, G.a.tl, fl.a2); }
          if constexpr (FB) { wb0 = __builtin_fma(fl.b0, G.b.tm, fl.a0); wb1 = __builtin_fma(fl.b1, G.b.tm, fl.a1); wb2 = __builtin_fma(fl.b2, G.b.tl, fl.a2); }
          if (kTestHooks && sv.schur_variant == 4) pm = 0;   // ablation: loads only (instrumented build)
          // ONE code path for full and partial masks: every MFMA behind a scalar test of its block's bit (a second, branch-free path for the
          // full mask made the register allocator keep two homes for the 72 accumulator registers and copy them over around every group:
          // 3.5 v_mov_b64 per MFMA in the round-5 build, 2 in round 4's)
          const unsigned pm27 = pm * 0x40201u;   // the nine bits once per coordinate: every MFMA tests a bit of its own (one s_bitcmp1 + branch; the same bit three times made the compiler keep the tests as lane masks and turn them over on the vector unit)
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            double a[3], b[3];
            operands(G.a, t, wa0, wa1, wa2, a); operands(G.b, t, wb0, wb1, wb2, b);
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
              for (int J = 0; J < 3; ++J)
                if ((!DIAG || J <= I) && __builtin_expect(((pm27 >> (9 * t + 3 * I + J)) & 1u) != 0u, 1)) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], b[J], acc[I][J], 0, 0, 0);
            if (DIAG) {
#pragma unroll
              for (int I = 0; I < 3; ++I) racc[I] += a[I] * G.z[t];
            }
          }
          issued += 3u * (unsigned)__builtin_popcount(pm);
        }
        fetch(base + d + kDepth, ring[d]);
      }
    }
  }
  if (lane == 0 && issued) atomicAdd(sv.schur_mfma_count, (unsigned long long)issued);   // a statistic (bench.py: issued against useful flops), not a result
  if (wave == 0) {
    const int64_t slots = (int64_t)(gridDim.x >> 3);      // (the first gridDim.x / 8 of every eighth went to the workgroups as they started)
    int64_t i = slots + (unsigned)__builtin_amdgcn_readfirstlane((int)ticket), c = (int64_t)xcd * per_xcd + i;
    int nx = persistent && i < per_xcd && c < sv.nchunk ? (int)c : -1;
    // its own eighth is done: the next chunk of another XCD's (the eighths have the same number of chunks, not of entries).  The other
    // counters are LOOKED at first, all at once — one round trip — and asked only where the look says there is something left: at the end
    // of the launch (and in a launch with a workgroup per chunk) nobody queues seven dependent atomics in front of its last epilogue.
    if (persistent && nx < 0 && per_xcd > slots) {
      const unsigned seen = lane < 8 ? __hip_atomic_load(sv.schur_next + 16 * ((xcd + lane) & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
      for (int k = 1; k < 8 && nx < 0; ++k) {
        const int y = (xcd + k) & 7;
        const int64_t left = (int64_t)slots + (unsigned)__shfl((int)seen, k, 64);
        if (left >= per_xcd || (int64_t)y * per_xcd + left >= sv.nchunk) continue;
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(sv.schur_next + 16 * y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i = slots + (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        c = (int64_t)y * per_xcd + i;
        if (i < per_xcd && c < sv.nchunk) nx = (int)c;
      }
    }
    if (lane == 0) s_next[0] = nx;
  }
  __syncthreads();   // everyone is done with the tables: the same LDS now takes the four partial tiles
  if (tr && tid == 0) { tr[4] = wall_clock64(); tr[7] += clock64(); }   // (shader cycles of the loop: the clock under this load)
  const int next = __builtin_amdgcn_readfirstlane(s_next[0]);
  info = next >= 0 ? sv.chunk_info[next] : int4{0, 0, 0, 0};
  stage_chunk(sv, info, tid, st);   // (in flight under the epilogue; nothing to read behind the last chunk)
  double* buf = smem + wave * (kTile * TPITCH);
  // every result to its place in the tile: a factored side's rows were taken in this kernel's own order (factored_row).  A factored tile
  // paired with itself formed the blocks J <= I of that order: each lands twice, as (row, column) and as (column, row) — the tile comes
  // out symmetric in full (S_ij and S_ji are the same products summed in the same order: the same bits).
#pragma unroll
  for (int I = 0; I < 3; ++I)
#pragma unroll
    for (int J = 0; J < 3; ++J) {
      if (DIAG && FA && J > I) continue;
      const int cb = FB ? factored_row(J, r) : 16 * J + r;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int ra = FA ? factored_row(I, g + 4 * v) : 16 * I + g + 4 * v;
        buf[ra * TPITCH + cb] = acc[I][J][v];
        if (DIAG && FA && J < I) buf[cb * TPITCH + ra] = acc[I][J][v];
      }
    }
  double* rvec = smem + 4 * kTile * TPITCH + wave * kTile;
  if (DIAG) {
#pragma unroll
    for (int I = 0; I < 3; ++I) {
      double x = racc[I];
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
      if (lane < 16) rvec[FA ? factored_row(I, lane) : 16 * I + lane] = x;
    }
  }
  __syncthreads();
  double* part = sv.schur_part + (size_t)chunk * (kTile * kTile + kTile);
  for (int e = tid; e < kTile * kTile; e += 256) {
    const int o = (e / kTile) * TPITCH + e % kTile;
    part[e] = (smem[o] + smem[kTile * TPITCH + o]) + (smem[2 * kTile * TPITCH + o] + smem[3 * kTile * TPITCH + o]);
  }
  if (DIAG && tid < kTile) { const double* v = smem + 4 * kTile * TPITCH; part[kTile * kTile + tid] = (v[tid] + v[kTile + tid]) + (v[2 * kTile + tid] + v[3 * kTile + tid]); }
  if (tr && tid == 0) tr[5] = wall_clock64();
  return next;
}

// kDepth = groups of four entries in flight per wave (18 loads each in full form, 16 factored: vmcnt counts to 63).  Two waves per SIMD (two workgroups
// per CU) fit 256 registers with two groups in flight; three need 300.
// PERSISTENT form (the default): 8 x min(an eighth of the chunk list, the XCD's workgroup slots) workgroups; workgroup b starts with chunk
// b >> 3 of eighth b & 7 (workgroups go round-robin over the 8 XCDs, so XCD x walks the x-th eighth of the chunk list in order and its L2
// sees the repeats: consecutive chunks share records — host: chunk numbering) and then takes the eighth's next untaken chunk (a counter per
// eighth) until there is none.  The last workgroup to leave puts the counters back to zero for the next launch.
// RSBA_SCHUR_VARIANT=2: one workgroup per chunk (rounds 2 - 4).
template <int kDepth, int kWavesPerSimd>
__global__ __launch_bounds__(256, kWavesPerSimd) void schur_tile_kernel(const SolverDev sv, const double* __restrict__ Pm, const double* __restrict__ zz, int persistent) {
  if (lm_stopped(sv.ctl)) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int s_next[2];
  const int per_xcd = (sv.nchunk + 7) / 8, xcd = (int)(blockIdx.x & 7);
  int chunk = sv.schur_linear ? (int)blockIdx.x : xcd * per_xcd + (int)(blockIdx.x >> 3);
  if (chunk >= sv.nchunk || (!sv.schur_linear && (int)(blockIdx.x >> 3) >= per_xcd)) chunk = -1;
  int4 info = int4{0, 0, 0, 0};
  ChunkStage st;
  if (chunk >= 0) { info = sv.chunk_info[chunk]; stage_chunk(sv, info, (int)threadIdx.x, st); }
  while (chunk >= 0) {
    const int f = info.w;
    if (f & 1) { if (f & 2) chunk = schur_chunk<true, kDepth, true, true>(sv, Pm, zz, chunk, info, st, smem, s_next, persistent != 0, xcd, per_xcd); else chunk = schur_chunk<true, kDepth, false, false>(sv, Pm, zz, chunk, info, st, smem, s_next, persistent != 0, xcd, per_xcd); }
    else if ((f & 6) == 6) chunk = schur_chunk<false, kDepth, true, true>(sv, Pm, zz, chunk, info, st, smem, s_next, persistent != 0, xcd, per_xcd);
    else if (f & 4) chunk = schur_chunk<false, kDepth, false, true>(sv, Pm, zz, chunk, info, st, smem, s_next, persistent != 0, xcd, per_xcd);        // an intrinsics pseudo tile (full form) against a frame tile
    else if (f & 2) chunk = schur_chunk<false, kDepth, true, false>(sv, Pm, zz, chunk, info, st, smem, s_next, persistent != 0, xcd, per_xcd);        // (a frame tile against a lower-numbered full-form tile: not produced by the plan today)
    else chunk = schur_chunk<false, kDepth, false, false>(sv, Pm, zz, chunk, info, st, smem, s_next, persistent != 0, xcd, per_xcd);
  }
  if (persistent && threadIdx.x == 0 && __hip_atomic_fetch_add(sv.schur_next + 128, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
    for (int e = 0; e < 8; ++e) __hip_atomic_store(sv.schur_next + 16 * e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sv.schur_next + 128, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// a group of a very long chunk list: partial[first] = sum of the group's partials, in list order (four running sums, chunk index mod 4).  An element
// per thread, kPremergeSplit workgroups per group (round 6; one workgroup per group walked its 2 352 elements in ten rounds of eight dependent
// loads each: 47 us at 4k cameras for 630 groups)
constexpr int kPremergeSplit = (kTile * kTile + kTile + 255) / 256;
__global__ __launch_bounds__(256) void schur_premerge_kernel(const SolverDev sv) {
  const int g0 = sv.pm_ptr[blockIdx.x], g1 = sv.pm_ptr[blockIdx.x + 1];
  constexpr size_t pstride = kTile * kTile + kTile;
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= (int)pstride) return;
  double ps[4] = {0, 0, 0, 0};
  for (int c = g0; c < g1; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) if (c + u < g1) ps[u & 3] += sv.schur_part[(size_t)sv.pm_list[c + u] * pstride + e];
  }
  sv.schur_part[(size_t)sv.pm_list[g0] * pstride + e] = (ps[0] + ps[1]) + (ps[2] + ps[3]);   // (this thread's own element of the group's first partial: read above, nobody else's)
}

// kMergeSplit workgroups per tile pair (an element per thread: with few tile pairs — 100 cameras have 140 — a workgroup walking its
// tile in nine dependent rounds of loads was 25 us of latency): sum the chunk partials in order, add U / D_c^2 / g_c, identity
// padding, and store into the packed tile slot (transposed when the tile ordering swapped the pair).
// The kernel moves 150 MB at 1k cameras and took 57 us: not bandwidth but a CHAIN of dependent reads per thread — pair -> (I, J, chunk range) ->
// chunk ids -> partials -> [is the side factored -> column scale], [U offset -> U], damping -> store: five round trips, hidden only by
// occupancy (profiles/r06/schur_inline_merge.txt: the same chain inside the Schur kernel, where nothing hides it, cost more than this
// launch).  Round 6: ONE 64-byte descriptor per pair (host: solver.hip, tp_desc) holds everything the first two links used to fetch —
// I, J, the packed slot, the flags AND the first eight chunk ids — and whatever depends on the descriptor alone is requested before the
// partials are summed: descriptor -> {partials, U, scales, damping} -> store.  The sums are formed in the order they always were (eight
// interleaved running sums over the chunk list, chunk index mod 8), to the bit.
constexpr int kMergeSplit = kTile * kTile / 256;
static_assert(kMergeSplit * 256 == kTile * kTile, "an element per thread");
struct PairDesc { int32_t I, J, dst, flags, c0, c1, pad0, pad1, head[8]; };   // flags: bit 0 = store transposed, bit 1 / 2 = I / J side factored; [c0, c1) = the pair's range in tp_chunk_list, head = its first eight ids (-1: none)
static_assert(sizeof(PairDesc) == 64, "one 64-byte line per pair");
// sum over the pair's chunk list of element `off` of the partials (tile elements, then the kTile rhs rows)
__device__ __forceinline__ double merge_sum(const SolverDev& sv, const PairDesc& d, int off) {
  constexpr size_t pstride = kTile * kTile + kTile;
  double ps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < 8; ++u) if (d.head[u] >= 0) ps[u] += sv.schur_part[(size_t)d.head[u] * pstride + off];
  for (int ch = d.c0 + 8; ch < d.c1; ch += 8) {   // (a pair with more than eight chunks: up to kMergeGroup heads of pre-reduced groups)
#pragma unroll
    for (int u = 0; u < 8; ++u) if (ch + u < d.c1) ps[u] += sv.schur_part[(size_t)sv.tp_chunk_list[ch + u] * pstride + off];
  }
  return ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
}
__global__ __launch_bounds__(256) void schur_merge_kernel(const DeviceProblem dp, const SolverDev sv, double inv_radius) {
  const int tp = blockIdx.x, tid = threadIdx.x;
  const PairDesc d = *reinterpret_cast<const PairDesc*>(sv.tp_desc + 16 * (size_t)tp);
  const int I = d.I, J = d.J, CD = sv.CD, FT = sv.FT;
  {
    const int e = blockIdx.y * 256 + tid;
    const int rt = e / kTile, ct = e % kTile;
    const int x = rt / CD, y = ct / CD, r = rt % CD, c = ct % CD;
    const int a = I * FT + x, b = J * FT + y;
    const bool real = a < sv.Fx && b < sv.Fx, on_diag = a == b && r == c;
    // everything that hangs on the descriptor alone, requested BEFORE the partials are waited for (clamped addresses where the value is not used)
    const int64_t add = sv.tp_add[((size_t)tp * FT + x) * FT + y];
    const double sa = (d.flags & 2) ? (a < sv.F ? dp.scale_pose[(size_t)a * CD + r] : 0.0) : 1.0;
    const double sb = (d.flags & 4) ? (b < sv.F ? dp.scale_pose[(size_t)b * CD + c] : 0.0) : 1.0;
    const double lead_a = sv.frame_lead ? sv.frame_lead[a] : (sv.lead != 0 ? 1.0 : 0.0);   // does this rank add the frame's replicated terms (sharded factorisation: the owner of its part)
    const double damp = (real && on_diag) ? sv.diag_c[(size_t)a * CD + r] : 0.0;
    const double u = (real && add >= 0) ? sv.U[add + (size_t)r * CD + c] : 0.0;
    if (sv.ctl) inv_radius = 1.0 / sv.ctl[kCtlRadius];
    double sum = merge_sum(sv, d, e);
    // a factored side left its column scales out of the products (they do not depend on the point): applied here, once per element
    if (d.flags & 2) sum *= sa;
    if (d.flags & 4) sum *= sb;
    const bool lead = lead_a != 0.0;
    double val;
    if (!real) val = (on_diag && lead) ? 1.0 : 0.0;     // padding frames of the last tile
    else {
      val = u - sum;
      if (on_diag && lead) val += damp * inv_radius;
    }
    double* dst = sv.S + (size_t)d.dst * (kTile * kTile);
    if (d.flags & 1) dst[(size_t)ct * kTile + rt] = val; else dst[e] = val;
  }
  if (I == J && blockIdx.y == 0 && tid < kTile) {
    const int a = I * FT + tid / CD;
    const double sa = (d.flags & 2) ? (a < sv.F ? dp.scale_pose[(size_t)I * kTile + tid] : 0.0) : 1.0;
    const double lead_a = sv.frame_lead ? sv.frame_lead[a] : (sv.lead != 0 ? 1.0 : 0.0);
    const double g = a < sv.Fx ? sv.gc[(size_t)I * kTile + tid] : 0.0;
    double sum = merge_sum(sv, d, kTile * kTile + tid);
    if (d.flags & 2) sum *= sa;
    sv.rhs[(size_t)I * kTile + tid] = (a < sv.Fx) ? (lead_a != 0.0 ? g : 0.0) - sum : 0.0;
  }
}

// K7 + K8  point steps and the model cost change in ONE pass over the point-major records:
//   y_p,j = L_j^-T ( z_j - L_j^-1 u_j ),  u_j = sum_o Jp_o^T t_o,  t_o = Jc_o y_c(frame(o)) + Ji_o y_i
// (the back-substitution y_p = L^-T (z - sum_o P_o^T y_c) with P_o = Jc_o^T Jp_o L^-T written out), and
//   model_cost_change = -sum_o m_o.(r_o + m_o / 2),  m_o = -(t_o + Jp_o y_p)      (TrustRegionMinimizer)
// expanded per point so that it needs nothing per observation beyond what the same pass accumulates:
//   sum_o m.(r + m/2) = -sum r.t - y_p.g_p + 1/2 sum |t|^2 + y_p.u + 1/2 y_p^T V y_p      (V, g_p from K2b).
// The records (256 B at 1k cameras) are read once — before, the P records (288 B) and then the Jacobian records
// were each streamed by their own kernel.  One wave owns 64 consecutive points = one contiguous slot range, moves it
// through LDS 64 records at a time with fully coalesced 512-B wave loads (the next 64 already in flight in
// registers), every lane reduces ITS record to 5 numbers, and the lane that owns the point sums its records' numbers
// in slot order (fixed order: deterministic).
template <int CD, int KC>
__global__ __launch_bounds__(256) void point_step_kernel(const DeviceProblem dp, const SolverDev sv) {
  if (lm_stopped(sv.ctl)) return;
  constexpr int REC = 8 + 2 * KC, PITCH = REC | 1, off = KC - CD, NC = 5;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ double s_red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* buf = smem + (size_t)wave * (64 * PITCH + 64 * NC);
  double* cbuf = buf + 64 * PITCH;
  const int64_t j0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  double mc = 0.0;
  if (j0 < dp.M) {   // wave-uniform
    const int jn = (int)(dp.M - j0 < 64 ? dp.M - j0 : 64);
    const bool mine = lane < jn;
    const int64_t j = j0 + (mine ? lane : 0);
    const int64_t lo = mine ? sv.point_ptr[j] : 0, hi = mine ? sv.point_ptr[j + 1] : 0;
    const int64_t sb = sv.point_ptr[j0], se = sv.point_ptr[j0 + jn];
    double acc[NC] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double pre[REC];
    int pre_frame = 0;
    // loads of the chunk at c0; indices are clamped into the wave's range instead of predicated (the values of a
    // clamped lane are never used), so nothing next to the load makes the compiler wait for it
    auto issue = [&](int64_t c0) {
      const int64_t last = (se - c0) * REC - 1;
      const double* src = lm_records(dp, false) + (size_t)c0 * REC;
#pragma unroll
      for (int k = 0; k < REC; ++k) { const int64_t idx = k * 64 + lane; pre[k] = src[idx < last ? idx : last]; }
      pre_frame = sv.slot_frame[c0 + lane < se ? c0 + lane : se - 1];
    };
    if (sb < se) issue(sb);
    for (int64_t c0 = sb; c0 < se; c0 += 64) {
      const int nrec = (int)(se - c0 < 64 ? se - c0 : 64);
#pragma unroll
      for (int k = 0; k < REC; ++k) { const int idx = k * 64 + lane; buf[(idx / REC) * PITCH + idx % REC] = pre[k]; }
      const int frame = pre_frame;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (c0 + 64 < se) issue(c0 + 64);
      if (lane < nrec) {
        const double* rec = buf + lane * PITCH;
        const double* yc = sv.step + (size_t)frame * CD;
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int a = 0; a < CD; ++a) { const double y = yc[a]; t0 += rec[8 + off + a] * y; t1 += rec[8 + KC + off + a] * y; }
        if (off > 0) {
          const double* yi = sv.step + ((size_t)sv.F + (size_t)(sv.NIB > 1 ? dp.frame_intr[frame] : 0) * sv.NPF) * CD;   // step of the frame's intrinsics block: 9 coordinates across its pseudo frames
#pragma unroll
          for (int k = 0; k < off; ++k) { const double y = yi[k]; t0 += rec[8 + k] * y; t1 += rec[8 + KC + k] * y; }
        }
        double* c = cbuf + lane * NC;
        c[0] = rec[2] * t0 + rec[5] * t1; c[1] = rec[3] * t0 + rec[6] * t1; c[2] = rec[4] * t0 + rec[7] * t1;   // Jp^T t
        c[3] = t0 * t0 + t1 * t1;
        c[4] = rec[0] * t0 + rec[1] * t1;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int64_t s_lo = lo > c0 ? lo : c0, s_hi = hi < c0 + nrec ? hi : c0 + nrec;
      for (int64_t sidx = s_lo; sidx < s_hi; ++sidx) {
        const double* c = cbuf + (sidx - c0) * NC;
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[q] += c[q];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (mine) {
      const double* li = sv.Linv + (size_t)j * 6;
      const double* z = sv.z + (size_t)j * 3;
      const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
      const double u0 = acc[0], u1 = acc[1], u2 = acc[2];
      const double w0 = z[0] - i00 * u0, w1 = z[1] - (i10 * u0 + i11 * u1), w2 = z[2] - (i20 * u0 + i21 * u1 + i22 * u2);
      const double y0 = i00 * w0 + i10 * w1 + i20 * w2, y1 = i11 * w1 + i21 * w2, y2 = i22 * w2;
      double* yp = sv.yp + (size_t)j * 3;
      yp[0] = y0; yp[1] = y1; yp[2] = y2;
      const double* v = sv.V + (size_t)j * 6;   // xx xy xz yy yz zz
      const double* g = sv.gp + (size_t)j * 3;
      const double vy0 = v[0] * y0 + v[1] * y1 + v[2] * y2, vy1 = v[1] * y0 + v[3] * y1 + v[4] * y2, vy2 = v[2] * y0 + v[4] * y1 + v[5] * y2;
      mc = -acc[4] - (y0 * g[0] + y1 * g[1] + y2 * g[2]) + 0.5 * acc[3] + (y0 * u0 + y1 * u1 + y2 * u2) + 0.5 * (y0 * vy0 + y1 * vy1 + y2 * vy2);
    }
  }
  mc = wsum(mc);
  if (lane == 0) s_red[wave] = mc;
  __syncthreads();
  if (threadIdx.x == 0) sv.partial[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// ---------------------------------------------------------------------------------------------
// The point-side passes RECOMPUTE every observation's record (lm_record.hpp) instead of reading the point-major copy the
// evaluation kernel used to leave for them — 24 B of observation + cached poses in, not 256 - 400 B of record.  Same arithmetic
// as the record-based kernels above, which remain for problems with SEVERAL intrinsics parameter blocks (per-frame f.cam:
// a point then owns one virtual record group per block it is seen through).
// ---------------------------------------------------------------------------------------------
__global__ void slot_xy_kernel(const DeviceProblem dp, double2* __restrict__ slot_xy) {   // once per plan: observations in slot (point-major) order
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < dp.N) slot_xy[dp.obs_slot[i]] = dp.xy[i];
}

// the record of slot s (clamped by the caller): pose and scales straight from L2 (F x 192 B: resident)
template <bool CAL, int P>
__device__ __forceinline__ void slot_record(const DeviceProblem& dp, const SolverDev& sv, int64_t s, ObsOut<CAL, P>& o, int& frame, int& point) {
  constexpr int CD = 6 * P;
  frame = sv.slot_frame[s]; point = sv.slot_point[s];
  const double2 xy = sv.slot_xy[s];
  double pose[CD], psc[CD];
#pragma unroll
  for (int k = 0; k < CD; ++k) { pose[k] = dp.poses[(size_t)frame * CD + k]; psc[k] = dp.scale_pose[(size_t)frame * CD + k]; }
  double half_rho; bool dropped;
  lm_observation<CAL, P>(dp, frame, point, xy.x, xy.y, pose, psc, o, half_rho, dropped);
}

// K5b without records: P = Jc^T (Jp L^-T) of 64 consecutive slots per wave and step, into the group layout (see project_kernel).
// A slot whose group is stored FACTORED (two-pose frame tiles, solver_state.hpp: kGroupFactored) leaves q = Jq^T (Jp L^-T), 6 x 3 — the
// block with respect to the interpolated pose, loss-corrected, WITHOUT the (1 - tau) / tau weights and without the column scales — and
// tau: the twelve rows (1 - tau) q | tau q are formed by the Schur kernel in registers, the column scales are applied where its partial
// tiles are merged.  Half the bytes written here (the pass is bound by its stores) and read there.
//   group layout: [c][16] sources s = 0..15 of coordinate c | [c][8] sources 16..23 | tau[4];  source s = 6 (frame position in the tile) + pose coordinate
// ALLF: every slot's group is factored (SolverDev::all_real_factored — the rule for two-pose problems; a tile that mixes real and pseudo
// frames is the exception): 19 doubles per slot go through LDS instead of 36 — 40 KB per workgroup instead of 76, and with the register
// budget of three waves per SIMD the pass, which is bound by its fp64 arithmetic, runs three workgroups per CU instead of two.
template <bool CAL, int P, bool ALLF>
__global__ __launch_bounds__(256, ALLF ? 3 : 1) void project_rc_kernel(const DeviceProblem dp, const SolverDev sv, int nch) {
  if (lm_stopped(sv.ctl)) return;   // (device-side trust region: the solve is over, iterations enqueued ahead fall through)
  static_assert(!ALLF || P == 2, "factored groups are a two-pose form");
  constexpr int CD = 6 * P, OUT = CD * 3, PITCH = ALLF ? 19 : (OUT | 1), kPer = 64 / CD, OP = CAL ? 0 : 9;   // OP: the pose columns follow the 9 intrinsics columns
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* buf = smem + (size_t)wave * (64 * PITCH + 32);
  uint32_t* s_gpos = reinterpret_cast<uint32_t*>(buf + 64 * PITCH);
  const int64_t sb = ((int64_t)blockIdx.x * 4 + wave) * 64 * nch;
  if (sb >= dp.N) return;
  const int64_t se = sb + 64 * nch < dp.N ? sb + 64 * nch : dp.N;
  for (int64_t s0 = sb; s0 < se; s0 += 64) {
    const int64_t nslot = se - s0 < 64 ? se - s0 : 64;
    const int64_t s = s0 + lane < se ? s0 + lane : se - 1;   // (lanes past the end repeat the last slot; nothing of theirs is stored)
    ObsOut<CAL, P> o;
    int frame, j;
    slot_record<CAL, P>(dp, sv, s, o, frame, j);
    const double* li = sv.Linv + (size_t)j * 6;
    const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
    double B[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double p0 = o.J[r][OP + CD], p1 = o.J[r][OP + CD + 1], p2 = o.J[r][OP + CD + 2];
      B[r][0] = p0 * i00; B[r][1] = p0 * i10 + p1 * i11; B[r][2] = p0 * i20 + p1 * i21 + p2 * i22;
    }
    const uint32_t gpos = sv.slot_gpos[s];
    if (ALLF || (P == 2 && gpos_factored(gpos))) {
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) buf[lane * PITCH + k * 6 + a] = o.Jq[0][a] * B[0][k] + o.Jq[1][a] * B[1][k];
      buf[lane * PITCH + 18] = o.tau;
    } else {
#pragma unroll
      for (int a = 0; a < CD; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) buf[lane * PITCH + k * CD + a] = o.J[0][OP + a] * B[0][k] + o.J[1][OP + a] * B[1][k];
    }
    s_gpos[lane] = gpos;
    const bool any_factored = ALLF || (P == 2 && __ballot(gpos_factored(gpos)) != 0ull), any_full = !ALLF && __ballot(!(P == 2 && gpos_factored(gpos))) != 0ull;   // (wave-uniform: a wave's slots are almost always of one kind)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (any_factored) {
      // factored slots: six lanes per slot (a pose coordinate each), ten slots per pass; the lane of coordinate 0 also stores tau
      constexpr int kPerF = 10;
      const int my = lane / 6, w = lane % 6;
      if (my < kPerF) {
#pragma unroll 2
        for (int i = 0; i * kPerF < 64; ++i) {
          const int sl = i * kPerF + my;
          if (sl < nslot) {
            const uint32_t gp = s_gpos[sl];
            if (ALLF || gpos_factored(gp)) {
              const int src_no = 6 * gpos_pos(gp) + w;   // source 0..23 of the group
              double* dst = sv.Pm + gpos_group(gp) + (src_no < 16 ? src_no : 48 + (src_no - 16));
              const int stride = src_no < 16 ? 16 : 8;
              const double* src = buf + sl * PITCH + w;
#pragma unroll
              for (int comp = 0; comp < 3; ++comp) dst[comp * stride] = src[comp * 6];
              if (w == 0) sv.Pm[gpos_group(gp) + 72 + gpos_pos(gp)] = buf[sl * PITCH + 18];
            }
          }
        }
      }
    }
    if (any_full) {
      const int my = lane / CD, w = lane % CD;
      if (my < kPer) {
#pragma unroll 2
        for (int i = 0; i * kPer < 64; ++i) {
          const int sl = i * kPer + my;
          if (sl < nslot) {
            const uint32_t gp = s_gpos[sl];
            if (!gpos_factored(gp)) {
              double* dst = sv.Pm + (gpos_group(gp) + (size_t)(gpos_pos(gp) * CD + w));
              const double* src = buf + sl * PITCH + w;
#pragma unroll
              for (int comp = 0; comp < 3; ++comp) dst[comp * kTile] = src[comp * CD];
            }
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// One wave owns kSweepPoints consecutive points = one contiguous slot range and walks it 64 slots at a time: every lane reduces ITS
// slot's recomputed record to NC numbers (per_slot), and the lane that owns a point adds its slots' numbers in slot order (fixed
// order: deterministic) before per_point finishes the point.  -> what per_point returns, summed over the workgroup (fixed order).
// (Sixteen points — a few hundred slots — per wave: the records are computed, not streamed, so the pass wants many waves in flight,
// not long ones; 64 points per wave left the 100-camera scene with 40 workgroups.)
constexpr int kSweepPoints = 16;
// ... and fewer when the problem is small: the waves of a pass all run at once (10 000 points are 625 waves of sixteen on 1 024 SIMDs),
// so what a pass takes is what ONE wave takes — its points' slots, 64 at a time — and a wave with four points is done in a third of
// the time of a wave with sixteen.  sweep_points(M): 16 from 32 k points, 8 from 16 k, 4 below.
inline int sweep_points(int64_t M) {
  static const int forced = [] { const char* e = std::getenv("RSBA_SWEEP_POINTS"); const int v = e ? std::atoi(e) : 0; return v >= 1 && v <= kSweepPoints ? v : 0; }();   // (tuning aid)
  return forced ? forced : M >= 32768 ? 16 : M >= 16384 ? 8 : 4;
}
// The sums are taken by ALL 64 lanes: the wave's (point, component) pairs — 16 x NC — are dealt to the lanes, each adding its pairs'
// numbers over the point's slots in slot order (the same order as ever: same bits) — when only the 16 lanes that own a point did
// this, the other 48 waited through 20 x NC dependent LDS reads and adds per point: most of the sweep's time (the virtual-record sweep
// of a shared intrinsics block, NC = 27, took 0.91 ms at 4k cameras against 0.35 for NC = 9 over the same records).
// NCP: the slots' numbers go through LDS NCP components at a time (NC / NCP passes per 64 slots, the record computed once): the 27 of the
// virtual-record sweep in three passes of nine take 18 KB per workgroup instead of 55 — room beside the projection pass, which runs at the
// same time on its own stream — and the pairs' slot ranges are kept once per pass layout, not per component.
// WAVE_DOUBLES: a wave's share of the dynamic LDS (>= 64 NCP; the fused sweep below keeps a staging area in the same doubles).  per_slot also
// gets the slot's index and the wave's LDS; per_batch(wave's LDS, slots of the batch) runs once the 64 slots of a batch have been through
// per_slot, between two wave barriers, BEFORE the batch's numbers go into the same LDS.
template <bool CAL, int P, int NC, int NCP, int WAVE_DOUBLES, class PerSlot, class PerBatch, class PerPoint>
__device__ __forceinline__ double point_sweep_hooked(const DeviceProblem& dp, const SolverDev& sv, double* smem, int sp, int64_t block, PerSlot per_slot, PerBatch per_batch, PerPoint per_point) {
  static_assert(NC % NCP == 0 && kSweepPoints * NC <= 64 * NCP && WAVE_DOUBLES >= 64 * NCP, "passes of equal width; the final gather fits the buffer");
  constexpr int NPART = NC / NCP, NPAIR = kSweepPoints * NCP, PER = (NPAIR + 63) / 64;   // pairs (point, component of a pass): sized for the most points a wave takes; sp <= kSweepPoints of them this launch
  __shared__ double s_red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* cbuf = smem + (size_t)wave * WAVE_DOUBLES;
  const int64_t j0 = (block * 4 + wave) * sp;
  double ret = 0.0;
  auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  if (j0 < dp.M) {   // wave-uniform
    const int jn = (int)(dp.M - j0 < sp ? dp.M - j0 : sp);
    const bool mine = lane < jn;
    const int64_t j = j0 + (mine ? lane : 0);
    const int64_t lo = mine ? sv.point_ptr[j] : 0, hi = mine ? sv.point_ptr[j + 1] : 0;
    const int64_t sb = sv.point_ptr[j0], se = sv.point_ptr[j0 + jn];
    // this lane's pairs: pair = lane + 64 i -> point pair / NCP of the wave, component pair % NCP of every pass; the point's slot range from its owner lane
    double acc[NPART][PER]; int64_t plo[PER], phi[PER]; int pq[PER], pp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int pair = lane + 64 * i, pj = pair / NCP;
#pragma unroll
      for (int part = 0; part < NPART; ++part) acc[part][i] = 0.0;
      pq[i] = pair % NCP; pp[i] = pj;
      const long long l = __shfl((long long)lo, pj < kSweepPoints ? pj : 0, 64), h = __shfl((long long)hi, pj < kSweepPoints ? pj : 0, 64);
      const bool live = pair < jn * NCP;
      plo[i] = live ? l : 0; phi[i] = live ? h : 0;
    }
    for (int64_t c0 = sb; c0 < se; c0 += 64) {
      const int nrec = (int)(se - c0 < 64 ? se - c0 : 64);
      double c[NC];
      {
        const int64_t s = c0 + lane < se ? c0 + lane : se - 1;
        ObsOut<CAL, P> o;
        int frame, pt;
        slot_record<CAL, P>(dp, sv, s, o, frame, pt);
        per_slot(o, frame, pt, s, cbuf, c);
      }
      per_batch(cbuf, nrec, wave_sync);
#pragma unroll
      for (int part = 0; part < NPART; ++part) {
#pragma unroll
        for (int q = 0; q < NCP; ++q) cbuf[lane * NCP + q] = c[part * NCP + q];
        wave_sync();
        // (four numbers of a pair are READ before the first of them is added — in slot order, as ever: the same bits — with the rows' offsets
        // in the instructions: 2.5 instructions per number instead of the 9 of the plain loop, which was 17 - 29 % of these passes)
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int klo = (int)((plo[i] > c0 ? plo[i] : c0) - c0), khi = (int)((phi[i] < c0 + nrec ? phi[i] : c0 + nrec) - c0);
          const double* col = cbuf + pq[i];
          int k = klo;
          for (; k + 4 <= khi; k += 4) {
            const double* p = col + k * NCP;
            const double v0 = p[0], v1 = p[NCP], v2 = p[2 * NCP], v3 = p[3 * NCP];
            acc[part][i] += v0; acc[part][i] += v1; acc[part][i] += v2; acc[part][i] += v3;
          }
          for (; k < khi; ++k) acc[part][i] += col[k * NCP];
        }
        wave_sync();
      }
    }
    // the sums of a point back to the lane that owns it: [point][NC]
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
      for (int part = 0; part < NPART; ++part) if (lane + 64 * i < NPAIR) cbuf[pp[i] * NC + part * NCP + pq[i]] = acc[part][i];
    wave_sync();
    if (mine) {
      double a[NC];
#pragma unroll
      for (int q = 0; q < NC; ++q) a[q] = cbuf[lane * NC + q];
      ret = per_point(j, a);
    }
  }
  ret = wsum(ret);
  if (lane == 0) s_red[wave] = ret;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
template <bool CAL, int P, int NC, int NCP = NC, class PerSlot, class PerPoint>
__device__ __forceinline__ double point_sweep(const DeviceProblem& dp, const SolverDev& sv, double* smem, int sp, int64_t block, PerSlot per_slot, PerPoint per_point) {
  return point_sweep_hooked<CAL, P, NC, NCP, 64 * NCP>(dp, sv, smem, sp, block,
    [&](const ObsOut<CAL, P>& o, int frame, int pt, int64_t, double*, double* c) { per_slot(o, frame, pt, c); },
    [](double*, int, auto&) {}, per_point);
}

// K2b without records: V_j, g_p,j
template <bool CAL, int P>
__device__ __forceinline__ void point_blocks_sweep(const DeviceProblem& dp, const SolverDev& sv, double* smem, int sp, int64_t block) {
  constexpr int CD = (CAL ? 0 : 9) + 6 * P;   // columns in front of the point's
  point_sweep<CAL, P, 9>(dp, sv, smem, sp, block,
    [&](const ObsOut<CAL, P>& o, int, int, double c[9]) {
      const double r0 = o.r[0], r1 = o.r[1], p0[3] = {o.J[0][CD], o.J[0][CD + 1], o.J[0][CD + 2]}, p1[3] = {o.J[1][CD], o.J[1][CD + 1], o.J[1][CD + 2]};
      c[0] = p0[0] * p0[0] + p1[0] * p1[0]; c[1] = p0[0] * p0[1] + p1[0] * p1[1]; c[2] = p0[0] * p0[2] + p1[0] * p1[2];
      c[3] = p0[1] * p0[1] + p1[1] * p1[1]; c[4] = p0[1] * p0[2] + p1[1] * p1[2]; c[5] = p0[2] * p0[2] + p1[2] * p1[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) c[6 + k] = p0[k] * r0 + p1[k] * r1;
    },
    [&](int64_t j, const double acc[9]) {
#pragma unroll
      for (int k = 0; k < 6; ++k) sv.V[(size_t)j * 6 + k] = acc[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) sv.gp[(size_t)j * 3 + k] = acc[6 + k];
      return 0.0;
    });
}
template <bool CAL, int P>
__global__ __launch_bounds__(256) void point_blocks_rc_kernel(const DeviceProblem dp, const SolverDev sv, int sp) {
  if (lm_not_accepted(sv.ctl)) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  point_blocks_sweep<CAL, P>(dp, sv, smem, sp, blockIdx.x);
}
// The linearisation of an accepted candidate in ONE launch (the loop that runs without the host): the frames' camera blocks, the copy of
// the candidate over x, and the points' blocks side by side — neither reads what the other writes; the point sweeps read the candidate
// where it still lies (dq: dp with the trial buffers for parameters), since the copy over x is in flight beside them.
template <bool CAL, int P>
__global__ __launch_bounds__(256) void linearize_blocks_kernel(const DeviceProblem dp, const DeviceProblem dq, const SolverDev sv, int sp, int ntake) {
  if (lm_not_accepted(sv.ctl)) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x;
  if (b < dp.F) camera_reduce_frame<6 * P, CAL>(dp, sv, b);
  else if (b < dp.F + ntake) take_candidate_block(dp, sv, b - dp.F);
  else point_blocks_sweep<CAL, P>(dq, sv, smem, sp, (int64_t)b - dp.F - ntake);
}

// K7 + K8 without records (see point_step_kernel)
template <bool CAL, int P>
__global__ __launch_bounds__(256) void point_step_rc_kernel(const DeviceProblem dp, const SolverDev sv, int sp) {
  if (lm_stopped(sv.ctl)) return;
  constexpr int CD = 6 * P, OP = CAL ? 0 : 9, OX = OP + CD;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const double mc = point_sweep<CAL, P, 5>(dp, sv, smem, sp, blockIdx.x,
    [&](const ObsOut<CAL, P>& o, int frame, int, double c[5]) {
      const double* yc = sv.step + (size_t)frame * CD;
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int a = 0; a < CD; ++a) { const double y = yc[a]; t0 += o.J[0][OP + a] * y; t1 += o.J[1][OP + a] * y; }
      if (!CAL) {
        const double* yi = sv.step + (size_t)sv.F * CD;   // step of the (one) intrinsics block: 9 coordinates across its pseudo frames
#pragma unroll
        for (int k = 0; k < 9; ++k) { const double y = yi[k]; t0 += o.J[0][k] * y; t1 += o.J[1][k] * y; }
      }
      c[0] = o.J[0][OX] * t0 + o.J[1][OX] * t1; c[1] = o.J[0][OX + 1] * t0 + o.J[1][OX + 1] * t1; c[2] = o.J[0][OX + 2] * t0 + o.J[1][OX + 2] * t1;   // Jp^T t
      c[3] = t0 * t0 + t1 * t1;
      c[4] = o.r[0] * t0 + o.r[1] * t1;
    },
    [&](int64_t j, const double acc[5]) {
      const double* li = sv.Linv + (size_t)j * 6;
      const double* z = sv.z + (size_t)j * 3;
      const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
      const double u0 = acc[0], u1 = acc[1], u2 = acc[2];
      const double w0 = z[0] - i00 * u0, w1 = z[1] - (i10 * u0 + i11 * u1), w2 = z[2] - (i20 * u0 + i21 * u1 + i22 * u2);
      const double y0 = i00 * w0 + i10 * w1 + i20 * w2, y1 = i11 * w1 + i21 * w2, y2 = i22 * w2;
      double* yp = sv.yp + (size_t)j * 3;
      yp[0] = y0; yp[1] = y1; yp[2] = y2;
      const double* v = sv.V + (size_t)j * 6;   // xx xy xz yy yz zz
      const double* g = sv.gp + (size_t)j * 3;
      const double vy0 = v[0] * y0 + v[1] * y1 + v[2] * y2, vy1 = v[1] * y0 + v[3] * y1 + v[4] * y2, vy2 = v[2] * y0 + v[4] * y1 + v[5] * y2;
      return -acc[4] - (y0 * g[0] + y1 * g[1] + y2 * g[2]) + 0.5 * acc[3] + (y0 * u0 + y1 * u1 + y2 * u2) + 0.5 * (y0 * vy0 + y1 * vy1 + y2 * vy2);
    });
  if (threadIdx.x == 0) sv.partial[blockIdx.x] = mc;
}

// the virtual records of the intrinsics pseudo frames without records, ONE intrinsics block (the shared sess.cam): per point
// Q_j = sum_o Ji_o^T (Jp_o L_j^-T) (9 x 3), cut into the NPF pseudo-frame records of the point's virtual slots (see virtual_records_kernel)
template <int P>
__global__ __launch_bounds__(256) void virtual_records_rc_kernel(const DeviceProblem dp, const SolverDev sv, int sp) {
  if (lm_stopped(sv.ctl)) return;
  constexpr int CD = 6 * P, OX = 9 + CD;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  point_sweep<false, P, 27, 9>(dp, sv, smem, sp, blockIdx.x,
    [&](const ObsOut<false, P>& o, int, int j, double c[27]) {
      const double* li = sv.Linv + (size_t)j * 6;
      const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
      double B[2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double p0 = o.J[r][OX], p1 = o.J[r][OX + 1], p2 = o.J[r][OX + 2];
        B[r][0] = p0 * i00; B[r][1] = p0 * i10 + p1 * i11; B[r][2] = p0 * i20 + p1 * i21 + p2 * i22;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int m = 0; m < 3; ++m) c[3 * k + m] = o.J[0][k] * B[0][m] + o.J[1][k] * B[1][m];
    },
    [&](int64_t j, const double acc[27]) {
      const int64_t g = sv.point_vgroup[j];
      if (g < 0) return 0.0;
      for (int v = 0; v < sv.NPF; ++v) {
        const uint32_t gpos = sv.slot_gpos[dp.N + g * sv.NPF + v];   // (a pseudo frame's tile: full form)
        double* out = sv.Pm + gpos_group(gpos) + gpos_pos(gpos) * CD;
#pragma unroll
        for (int rl = 0; rl < CD; ++rl) {
          const int k = v * CD + rl;
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            double q = 0.0;
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) if (kk == k) q = acc[3 * kk + m];
            out[m * kTile + rl] = q;
          }
        }
      }
      return 0.0;
    });
}

// The same sweep writing the slots' (factored) P records as well — project_rc_kernel<false, 2, true> and virtual_records_rc_kernel<2> in ONE
// pass over the observations of a problem with ONE shared intrinsics block whose real frames' tiles are all factored (C5's class): the two
// passes evaluate the same observations against the same point factors, and together they are bound by the vector unit they share.
// A wave's LDS: [64][19] staging of q = Jq^T (Jp L^-T) | tau per slot, then 64 group positions; the sums' nine-component passes reuse its front.
constexpr int kFusedStage = 19, kFusedWaveDoubles = 64 * kFusedStage + 32;
__global__ __launch_bounds__(256) void virtual_project_rc_kernel(const DeviceProblem dp, const SolverDev sv, int sp) {
  if (lm_stopped(sv.ctl)) return;
  constexpr int P = 2, CD = 12, OX = 9 + CD;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  point_sweep_hooked<false, P, 27, 9, kFusedWaveDoubles>(dp, sv, smem, sp, blockIdx.x,
    [&](const ObsOut<false, P>& o, int, int j, int64_t s, double* wl, double c[27]) {
      const double* li = sv.Linv + (size_t)j * 6;
      const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
      double B[2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double p0 = o.J[r][OX], p1 = o.J[r][OX + 1], p2 = o.J[r][OX + 2];
        B[r][0] = p0 * i00; B[r][1] = p0 * i10 + p1 * i11; B[r][2] = p0 * i20 + p1 * i21 + p2 * i22;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int m = 0; m < 3; ++m) c[3 * k + m] = o.J[0][k] * B[0][m] + o.J[1][k] * B[1][m];
      // the slot's factored P record (project_rc_kernel: the same expressions, the same bits)
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) wl[lane * kFusedStage + k * 6 + a] = o.Jq[0][a] * B[0][k] + o.Jq[1][a] * B[1][k];
      wl[lane * kFusedStage + 18] = o.tau;
      reinterpret_cast<uint32_t*>(wl + 64 * kFusedStage)[lane] = sv.slot_gpos[s];
    },
    [&](double* wl, int nrec, auto& wave_sync) {
      wave_sync();
      const uint32_t* s_gpos = reinterpret_cast<const uint32_t*>(wl + 64 * kFusedStage);
      constexpr int kPerF = 10;   // six lanes per slot (a pose coordinate each), ten slots per pass; the lane of coordinate 0 also stores tau
      const int my = lane / 6, w = lane % 6;
      if (my < kPerF) {
#pragma unroll 2
        for (int i = 0; i * kPerF < 64; ++i) {
          const int sl = i * kPerF + my;
          if (sl < nrec) {
            const uint32_t gp = s_gpos[sl];
            const int src_no = 6 * gpos_pos(gp) + w;   // source 0..23 of the group
            double* dst = sv.Pm + gpos_group(gp) + (src_no < 16 ? src_no : 48 + (src_no - 16));
            const int stride = src_no < 16 ? 16 : 8;
            const double* src = wl + sl * kFusedStage + w;
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) dst[comp * stride] = src[comp * 6];
            if (w == 0) sv.Pm[gpos_group(gp) + 72 + gpos_pos(gp)] = wl[sl * kFusedStage + 18];
          }
        }
      }
      wave_sync();
    },
    [&](int64_t j, const double acc[27]) {
      const int64_t g = sv.point_vgroup[j];
      if (g < 0) return 0.0;
      for (int v = 0; v < sv.NPF; ++v) {
        const uint32_t gpos = sv.slot_gpos[dp.N + g * sv.NPF + v];   // (a pseudo frame's tile: full form)
        double* out = sv.Pm + gpos_group(gpos) + gpos_pos(gpos) * CD;
#pragma unroll
        for (int rl = 0; rl < CD; ++rl) {
          const int k = v * CD + rl;
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            double q = 0.0;
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) if (kk == k) q = acc[3 * kk + m];
            out[m * kTile + rl] = q;
          }
        }
      }
      return 0.0;
    });
}
__global__ __launch_bounds__(256) void reduce_sum_kernel(const double* partial, int n, double* out, double sign) {
  __shared__ double s_red[4];
  double v = 0.0;
  for (int k = threadIdx.x; k < n; k += 256) v += partial[k];
  v = wsum(v);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) *out = sign * (s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

// two sums in one launch: workgroup b reduces partial[b * n .. b * n + n) into out0 (b = 0) resp. out1 (b = 1), same order as reduce_sum_kernel
__global__ __launch_bounds__(256) void reduce_sum2_kernel(const double* partial, int n, double* out0, double* out1) {
  __shared__ double s_red[4];
  const double* src = partial + (size_t)blockIdx.x * n;
  double v = 0.0;
  for (int k = threadIdx.x; k < n; k += 256) v += src[k];
  v = wsum(v);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) *(blockIdx.x == 0 ? out0 : out1) = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// reduce_sum_kernel(pa, na, outa, -1) and reduce_sum2_kernel(pb, nb, out0, out1) by one launch of three workgroups
__global__ __launch_bounds__(256) void reduce_sum3_kernel(const double* pa, int na, double* outa, const double* pb, int nb, double* out0, double* out1) {
  __shared__ double s_red[4];
  const double* src = blockIdx.x == 0 ? pa : pb + (size_t)(blockIdx.x - 1) * nb;
  const int n = blockIdx.x == 0 ? na : nb;
  double v = 0.0;
  for (int k = threadIdx.x; k < n; k += 256) v += src[k];
  v = wsum(v);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double sum = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  if (blockIdx.x == 0) *outa = -1.0 * sum; else *(blockIdx.x == 1 ? out0 : out1) = sum;
}

// x_plus_delta = x + scale .* step; |x - x_plus_delta|^2 and |x|^2 over the reduced program's blocks
__global__ __launch_bounds__(256) void candidate_kernel(const DeviceProblem dp, const SolverDev sv, double* __restrict__ part) {
  __shared__ double s_red[2][4];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, nc = sv.n, np = 3 * (int64_t)dp.M;
  double st = 0.0, xx = 0.0;
  if (t < nc + np) {
    const bool cam = t < nc;
    const int64_t npose = (int64_t)sv.F * sv.CD;
    const bool intr = cam && t >= npose;
    const int64_t u = cam ? (intr ? t - npose : t) : t - nc;
    const int ii = intr ? intr_index(sv, u) : 0;     // coordinate of its intrinsics block, -1 = padding of a pseudo frame
    if (intr && ii < 0) { st = 0.0; xx = 0.0; }
    else {
    const double x = cam ? (intr ? dp.intr[ii] : dp.poses[u]) : dp.points[u];
    const double sc = cam ? cam_scale(dp, sv, t) : dp.scale_point[u];
    const double y = cam ? sv.step[t] : sv.yp[u];
    const double in = cam ? (intr ? sv.inprog_intr[u] : sv.inprog_pose[u]) : sv.inprog_point[u];
    const double xn = (sc > 0.0) ? x + (-y * sc) : x;
    if (intr) sv.trial_intr[ii] = xn; else if (cam) sv.trial_poses[u] = xn; else sv.trial_points[u] = xn;
    if (in > 0.0) { const double e = x - xn; st = e * e; xx = x * x; }
    }
  }
  st = wsum(st); xx = wsum(xx);
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = st; s_red[1][threadIdx.x >> 6] = xx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    part[gridDim.x + blockIdx.x] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
  }
}

// sharded solve, after the last iteration: every rank contributes the points it owns (the ones it has observations of),
// buf = [M][3] values | [M] owner count; after the all-reduce the owners' values replace the local copies
__global__ void own_points_kernel(const DeviceProblem dp, const SolverDev sv, double* buf) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= dp.M) return;
  const bool own = sv.point_ptr[j + 1] > sv.point_ptr[j];
  for (int k = 0; k < 3; ++k) buf[3 * j + k] = own ? dp.points[3 * j + k] : 0.0;
  buf[3 * (int64_t)dp.M + j] = own ? 1.0 : 0.0;
}
__global__ void merge_points_kernel(const DeviceProblem dp, const double* buf) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= dp.M || buf[3 * (int64_t)dp.M + j] != 1.0) return;
  for (int k = 0; k < 3; ++k) dp.points[3 * j + k] = buf[3 * j + k];
}

// exchange buffer (1): g_c | diag(U) | cost, fixed cost, failed blocks
// ---- trust-region control on the device ----
// Ceres 1.9's TrustRegionMinimizer loop body behind the linear solve, the same rules in the same
// order as the host form in solver.hip (rsba_solve) — one thread; every operation is an IEEE add / multiply / divide / sqrt /
// compare, so the two forms take bit-identical decisions.  Scalars of the iteration: sv.scalars (ScalarSlot), state: ctl (LmCtlSlot).
__device__ __forceinline__ void lm_push(double* ctl, rsba_iteration* trace, int cap, const rsba_iteration& it) {
  const int n = (int)ctl[kCtlNumTrace];
  if (trace && n < cap) trace[n] = it;
  ctl[kCtlNumTrace] = (double)(n + 1);
}
__device__ __forceinline__ void lm_decide_step(const SolverDev& sv, double* ctl, const LmRules& R, rsba_iteration* trace, int cap) {
#pragma clang fp contract(off)   // every product and sum rounded on its own, as the host form's are (1 - (t t) t would become an fma: one ulp of the radius)
  if (ctl[kCtlStatus] != 0.0) return;
  ctl[kCtlAccept] = 0.0;
  const double* sc = sv.scalars;
  if (sc[kDagSuspect] != 0.0) { ctl[kCtlStatus] = -1.0; return; }   // nothing of this iteration has touched x: the host repeats it on the level schedule
  double radius = ctl[kCtlRadius], decrease = ctl[kCtlDecrease];
  const double cost = ctl[kCtlCost], fixed = ctl[kCtlFixed], gmax = ctl[kCtlGmax];
  const int iteration = (int)ctl[kCtlIteration] + 1;
  ctl[kCtlIteration] = (double)iteration;
  rsba_iteration it;
  it.iteration = iteration; it.step_is_valid = 0; it.step_is_successful = 0; it.reserved = 0;
  it.cost = 0.0; it.cost_change = 0.0; it.gradient_max_norm = 0.0; it.step_norm = 0.0; it.relative_decrease = 0.0; it.trust_region_radius = 0.0; it.model_cost_change = 0.0;
  const double model_cost_change = sc[kModelCostChange];
  const bool cfail = sc[kSolveFailed] != 0.0, nfail = sc[kEvalFailed] != 0.0;
  double step_sq = sc[kStepSq], x_sq = sc[kXSq];
  if (sv.rt) {   // a free interFrameRatio is one more coordinate of x (its candidate: ratio_candidate_kernel)
    const double ratio = sv.rt[kRtRatio], rn = sv.rt[kRtRatioNew];
    step_sq += (ratio - rn) * (ratio - rn); x_sq += ratio * ratio;
  }
  const bool solved = !cfail && isfinite(model_cost_change) && isfinite(step_sq);
  const bool valid = solved && model_cost_change >= 0.0;
  it.model_cost_change = solved ? model_cost_change : 0.0;
  auto done = [&](int term) { it.cost = cost + fixed; it.trust_region_radius = radius; lm_push(ctl, trace, cap, it); ctl[kCtlStatus] = 1.0 + term; };
  if (!valid) {
    const int streak = (int)ctl[kCtlInvalidStreak] + 1;
    ctl[kCtlInvalidStreak] = (double)streak;
    if (streak >= R.max_num_consecutive_invalid_steps) { done(RSBA_FAILURE); return; }
    radius /= decrease; decrease *= 2.0;
    ctl[kCtlUnsuccessful] += 1.0;
    it.gradient_max_norm = gmax;
  } else {
    ctl[kCtlInvalidStreak] = 0.0; it.step_is_valid = 1;
    const double new_cost = nfail ? 1.7976931348623157e308 : (sc[kCost] + 0.0) - fixed;   // (the trial evaluation reports the total in kCost)
    it.step_norm = sqrt(step_sq);
    const double x_norm = sqrt(x_sq);
    if (it.step_norm <= R.parameter_tolerance * (x_norm + R.parameter_tolerance)) { done(RSBA_CONVERGENCE); return; }
    it.cost_change = cost - new_cost;
    if (fabs(it.cost_change) < R.function_tolerance * cost) { done(RSBA_CONVERGENCE); return; }
    it.relative_decrease = it.cost_change / model_cost_change;
    if (it.relative_decrease > R.min_relative_decrease) {
      it.step_is_successful = 1; ctl[kCtlSuccessful] += 1.0;
      const double t = 2.0 * it.relative_decrease - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
      radius = fmin(R.max_trust_region_radius, radius); decrease = 2.0;
      ctl[kCtlRadius] = radius; ctl[kCtlDecrease] = decrease; ctl[kCtlAccept] = 1.0;
      ctl[kCtlRecSel] = 1.0 - ctl[kCtlRecSel];   // (problems that keep records: the candidate's are the current point's from here on — device_state.hpp: lm_records)
      if (sv.rt) sv.rt[kRtRatio] = sv.rt[kRtRatioNew];
      // (the iteration's record is finished by lm_decide_gradient_kernel once the accepted point is linearised)
      ctl[kCtlPending] = it.relative_decrease; ctl[kCtlPending + 1] = it.cost_change; ctl[kCtlPending + 2] = it.step_norm; ctl[kCtlPending + 3] = it.model_cost_change;
      return;
    }
    ctl[kCtlUnsuccessful] += 1.0; it.gradient_max_norm = gmax;
    radius /= decrease; decrease *= 2.0;
  }
  ctl[kCtlRadius] = radius; ctl[kCtlDecrease] = decrease;
  it.cost = cost + fixed; it.trust_region_radius = radius;
  lm_push(ctl, trace, cap, it);
  if (radius < R.min_trust_region_radius) ctl[kCtlStatus] = 1.0 + RSBA_CONVERGENCE;
  else if (iteration >= R.max_num_iterations) ctl[kCtlStatus] = 1.0 + RSBA_NO_CONVERGENCE;
}
__global__ void lm_decide_step_kernel(const SolverDev sv, double* ctl, const LmRules R, rsba_iteration* trace, int cap) { lm_decide_step(sv, ctl, R, trace, cap); }
// after the linearisation of an accepted step: its cost, the gradient test, the iteration's record
__device__ __forceinline__ void lm_decide_gradient(const SolverDev& sv, double* ctl, const LmRules& R, rsba_iteration* trace, int cap) {
#pragma clang fp contract(off)
  if (ctl[kCtlStatus] != 0.0 || ctl[kCtlAccept] == 0.0) return;
  const double* sc = sv.scalars;
  rsba_iteration it;
  it.iteration = (int)ctl[kCtlIteration]; it.step_is_valid = 1; it.step_is_successful = 1; it.reserved = 0;
  it.relative_decrease = ctl[kCtlPending]; it.cost_change = ctl[kCtlPending + 1]; it.step_norm = ctl[kCtlPending + 2]; it.model_cost_change = ctl[kCtlPending + 3];
  const double fixed = ctl[kCtlFixed], radius = ctl[kCtlRadius];
  if (sc[kEvalFailed] != 0.0) {   // the evaluation at the accepted point failed: the host reports it (RSBA_ERR_EVALUATION_FAILED)
    it.cost = 0.0; it.gradient_max_norm = 0.0; it.trust_region_radius = 0.0;
    lm_push(ctl, trace, cap, it);
    ctl[kCtlStatus] = -2.0;
    return;
  }
  double gmax = sc[kGradMax];
  if (sv.rt) { const double ratio = sv.rt[kRtRatio]; gmax = fmax(gmax, fabs(ratio - fmax(sv.rt[kRtLb], ratio - sv.rt[kRtG]))); }   // the ratio's projected gradient (its block is bounded below)
  const double cost = sc[kCost];
  ctl[kCtlCost] = cost; ctl[kCtlGmax] = gmax;
  ctl[kCtlFinalCost] = fmin(ctl[kCtlFinalCost], cost + fixed);
  it.gradient_max_norm = gmax; it.cost = cost + fixed; it.trust_region_radius = radius;
  lm_push(ctl, trace, cap, it);
  if (gmax <= R.gradient_tolerance) ctl[kCtlStatus] = 1.0 + RSBA_CONVERGENCE;
  else if (radius < R.min_trust_region_radius) ctl[kCtlStatus] = 1.0 + RSBA_CONVERGENCE;
  else if (it.iteration >= R.max_num_iterations) ctl[kCtlStatus] = 1.0 + RSBA_NO_CONVERGENCE;
}
__global__ void lm_decide_gradient_kernel(const SolverDev sv, double* ctl, const LmRules R, rsba_iteration* trace, int cap) { lm_decide_gradient(sv, ctl, R, trace, cap); }

// ---- the same steps in fewer launches (the loop that never waits for the host pays ~4 us per launch of a dependent chain; at 100
// cameras that was a seventh of the iteration: profiles/r04/iteration_gaps.txt).  Same arithmetic in the same order as the kernels
// they stand for: the two forms of the trust-region loop still take bit-identical decisions. ----
// reduce_cost_kernel (kernels_eval.hip) + pack_trial_kernel + lm_decide_step_kernel
// (n < 0: the cost was reduced already — and the motion priors' added to it — by kernels of their own)
__global__ __launch_bounds__(256) void lm_verdict_step_kernel(const DeviceProblem dp, const SolverDev sv, double* cost2, int n, double* ctl, const LmRules R, rsba_iteration* trace, int cap) {
  if (ctl[kCtlStatus] != 0.0) return;
  if (n < 0) {
    if (threadIdx.x == 0) {
      sv.scalars[kCost] = cost2[0] + cost2[1]; sv.scalars[kFixedCost] = 0.0; sv.scalars[kEvalFailed] = (double)*dp.fail_count; sv.scalars[kSolveFailed] = (double)*sv.chol_fail;
      lm_decide_step(sv, ctl, R, trace, cap);
    }
    return;
  }
  __shared__ double s_red[3][4];
  double c = 0.0, f = 0.0, nf = 0.0;
  int k = threadIdx.x;
  for (; k + 15 * 256 < n; k += 16 * 256) {   // (as reduce_cost_kernel: sixteen strides' loads together, added in stride order)
    double vc[16], vf[16], vn[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { vc[u] = dp.cost_partial[k + 256 * u]; vf[u] = dp.fixed_partial[k + 256 * u]; vn[u] = dp.fail_partial[k + 256 * u]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) { c += vc[u]; f += vf[u]; nf += vn[u]; }
  }
  for (; k < n; k += 256) { c += dp.cost_partial[k]; f += dp.fixed_partial[k]; nf += dp.fail_partial[k]; }
  c = wsum(c); f = wsum(f); nf = wsum(nf);
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = c; s_red[1][threadIdx.x >> 6] = f; s_red[2][threadIdx.x >> 6] = nf; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double cost = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3], fixed = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
  const int fails = (int)(s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3]);
  cost2[0] = cost; cost2[1] = fixed; *dp.fail_count = fails;
  sv.scalars[kCost] = cost + fixed; sv.scalars[kFixedCost] = 0.0; sv.scalars[kEvalFailed] = (double)fails; sv.scalars[kSolveFailed] = (double)*sv.chol_fail;
  lm_decide_step(sv, ctl, R, trace, cap);
}
// local_linearize_kernel + gradient_max_kernel
__global__ __launch_bounds__(256) void lm_linearize_gradient_kernel(const DeviceProblem dp, const SolverDev sv, const double* cost2) {
  __shared__ double s_red[4];
  const int64_t nc = sv.n, np = 3 * (int64_t)dp.M;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (!lm_not_accepted(sv.ctl)) {
    if (t < nc) sv.udiag[t] = u_diag(sv, t);
    if (t == 0) { sv.scalars[kCost] = cost2[0]; sv.scalars[kFixedCost] = cost2[1]; sv.scalars[kEvalFailed] = (double)*dp.fail_count; }
  }
  double m = 0.0;
  if (t < nc + np) {
    const double sc = (t < nc) ? cam_scale(dp, sv, t) : dp.scale_point[t - nc];
    const double g = (t < nc) ? sv.gc[t] : sv.gp[t - nc];
    if (sc > 0.0) m = fabs(g / sc);
  }
  m = wmax(m);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) sv.partial[blockIdx.x] = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
}
// reduce_max_kernel + lm_decide_gradient_kernel; the state after the iteration goes to the host's slot for it
__global__ __launch_bounds__(256) void lm_verdict_gradient_kernel(const SolverDev sv, int n, double* ctl, const LmRules R, rsba_iteration* trace, int cap, double* snapshot, double seq) {
  __shared__ double s_red[4];
  double v = 0.0;
  for (int k = threadIdx.x; k < n; k += 256) v = fmax(v, sv.partial[k]);
  v = wmax(v);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (n >= 0) sv.scalars[kGradMax] = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));   // (n < 0: several ranks — the maximum came with the camera exchange)
    lm_decide_gradient(sv, ctl, R, trace, cap);
    *sv.chol_fail = 0; sv.scalars[kDagSuspect] = 0.0;   // begin_solve_kernel's job for the NEXT iteration: both flags were read by this iteration's first verdict
  }
  __syncthreads();
  // the state to the host's slot: a word per lane (stores to host memory one after the other cost a bus round trip each), the stamp behind them
  if (threadIdx.x < kCtlSeq) __hip_atomic_store(snapshot + threadIdx.x, ctl[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(snapshot + kCtlSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // the host polls this word: everything above is there when it shows
}

// x = x + delta: the candidate the last decision accepted becomes the current point (the host form swaps the two buffers)
__global__ void lm_take_candidate_kernel(const DeviceProblem dp, const SolverDev sv, int64_t npose, int64_t npoint, int64_t nintr) {
  if (lm_not_accepted(sv.ctl)) return;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < npose) dp.poses[t] = sv.trial_poses[t];
  else if (t < npose + npoint) dp.points[t - npose] = sv.trial_points[t - npose];
  else if (t < npose + npoint + nintr) dp.intr[t - npose - npoint] = sv.trial_intr[t - npose - npoint];
}

__global__ void pack_linearize_kernel(const DeviceProblem dp, const SolverDev sv, const double* cost2, int nslots) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < nslots) sv.xbuf[2 * sv.n + 3 + t] = 0.0;   // the ranks' gradient maxima (launch_gradient_max_points fills this rank's)
  if (t < sv.n) {
    sv.xbuf[t] = sv.gc[t];
    sv.xbuf[sv.n + t] = u_diag(sv, t);
  }
  if (t == 0) { sv.xbuf[2 * sv.n] = cost2[0]; sv.xbuf[2 * sv.n + 1] = cost2[1]; sv.xbuf[2 * sv.n + 2] = (double)*dp.fail_count; }
}
// without an exchange (one rank) the round trip through xbuf is one kernel: udiag = diag(U), the three scalars
__global__ void local_linearize_kernel(const DeviceProblem dp, const SolverDev sv, const double* cost2) {
  if (lm_not_accepted(sv.ctl)) return;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < sv.n) sv.udiag[t] = u_diag(sv, t);
  if (t == 0) { sv.scalars[kCost] = cost2[0]; sv.scalars[kFixedCost] = cost2[1]; sv.scalars[kEvalFailed] = (double)*dp.fail_count; }
}
__global__ void unpack_linearize_kernel(const DeviceProblem dp, const SolverDev sv) {
  if (lm_not_accepted(sv.ctl)) return;   // (device-side trust region on several ranks: the exchange of a rejected candidate's iteration carried nothing new)
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < sv.n) { sv.gc[t] = sv.xbuf[t]; sv.udiag[t] = sv.xbuf[sv.n + t]; }
  if (t == 0) { sv.scalars[kCost] = sv.xbuf[2 * sv.n]; sv.scalars[kFixedCost] = sv.xbuf[2 * sv.n + 1]; sv.scalars[kEvalFailed] = sv.xbuf[2 * sv.n + 2]; }
}
__global__ void pack_trial_kernel(const DeviceProblem dp, const SolverDev sv, const double* cost2) {
  sv.scalars[kCost] = cost2[0] + cost2[1];
  sv.scalars[kFixedCost] = 0.0;
  sv.scalars[kEvalFailed] = (double)*dp.fail_count;
  sv.scalars[kSolveFailed] = (double)*sv.chol_fail;
  if (!sv.lead) sv.scalars[kGradMax] = 0.0;   // several ranks: slots 0 - 11 travel in ONE sum; the maximum (the same on every rank) comes back as the lead rank's
}

inline int nblocks256(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

#define LAUNCH(kernel, grid, block, st, ...)                       \
  do {                                                             \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, st, __VA_ARGS__); \
    hipError_t e_ = hipGetLastError();                             \
    if (e_ != hipSuccess) return e_;                               \
  } while (0)

hipError_t launch_camera_blocks(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st, bool take_candidate, bool padding_is_zero) {
  const size_t CD2 = (size_t)sv.CD * sv.CD;
  const int64_t nparam = (int64_t)dp.F * dp.P * 6 + 3 * (int64_t)dp.M + (sv.NPF > 0 ? 9 * (int64_t)dp.NI : 0);
  const unsigned grid = (unsigned)dp.F + (take_candidate ? (unsigned)((nparam + 255) / 256) : 0u);   // (launch_lm_take_candidate's copy by extra workgroups of the same launch)
  if (sv.NPF > 0 && !padding_is_zero) {   // the padding coordinates of the pseudo frames stay zero (every other entry of these regions is ASSIGNED by the kernels below: once zero, the padding stays zero — the loop that must not touch U after a rejected step says so)
    hipError_t e = hipMemsetAsync(sv.U + (size_t)sv.F * CD2, 0, ((size_t)sv.NPF * sv.F + (size_t)sv.NIB * sv.NPF * sv.NPF) * CD2 * sizeof(double), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(sv.gc + (size_t)sv.F * sv.CD, 0, (size_t)sv.NIB * sv.NPF * sv.CD * sizeof(double), st);
    if (e != hipSuccess) return e;
  }
  if (!dp.cam_part) {   // no observations on this rank: nothing was accumulated
    if (take_candidate) { hipError_t e = launch_lm_take_candidate(dp, sv, st); if (e != hipSuccess) return e; }
    hipError_t e = hipMemsetAsync(sv.U, 0, (size_t)sv.F * CD2 * sizeof(double), st);
    if (e == hipSuccess) e = hipMemsetAsync(sv.gc, 0, (size_t)sv.F * sv.CD * sizeof(double), st);
    if (e == hipSuccess && sv.NPF > 0) e = hipMemsetAsync(sv.intr_part, 0, (size_t)sv.F * 54 * sizeof(double), st);
    return e;
  }
  if (sv.CD == 12) { if (dp.calibrated) LAUNCH((camera_reduce_kernel<12, true>), grid, 256, st, dp, sv); else LAUNCH((camera_reduce_kernel<12, false>), grid, 256, st, dp, sv); }
  else { if (dp.calibrated) LAUNCH((camera_reduce_kernel<6, true>), grid, 256, st, dp, sv); else LAUNCH((camera_reduce_kernel<6, false>), grid, 256, st, dp, sv); }
  return hipSuccess;
}
// camera blocks + the accepted candidate's copy over x + point blocks by one launch (linearize_blocks_kernel); *done = false: not for
// this problem (records kept, or nothing to sweep) — the caller launches them one after the other
hipError_t launch_linearize_blocks(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st, bool* done) {
  *done = false;
  if (!sv.slot_xy || !dp.cam_part || dp.M <= 0 || dp.F <= 0) return hipSuccess;
  DeviceProblem dq = dp;
  dq.poses = sv.trial_poses; dq.points = sv.trial_points;
  if (sv.NPF > 0) dq.intr = sv.trial_intr;
  const int64_t nparam = (int64_t)dp.F * dp.P * 6 + 3 * (int64_t)dp.M + (sv.NPF > 0 ? 9 * (int64_t)dp.NI : 0);
  const int ntake = (int)((nparam + 255) / 256), sp = sweep_points(dp.M), npts = (int)((dp.M + 4 * sp - 1) / (4 * sp));
  const size_t lds = (size_t)4 * 64 * 9 * sizeof(double);
  const dim3 grid((unsigned)(dp.F + ntake + npts));
  if (dp.calibrated) { if (sv.CD == 12) hipLaunchKernelGGL((linearize_blocks_kernel<true, 2>), grid, dim3(256), lds, st, dp, dq, sv, sp, ntake); else hipLaunchKernelGGL((linearize_blocks_kernel<true, 1>), grid, dim3(256), lds, st, dp, dq, sv, sp, ntake); }
  else { if (sv.CD == 12) hipLaunchKernelGGL((linearize_blocks_kernel<false, 2>), grid, dim3(256), lds, st, dp, dq, sv, sp, ntake); else hipLaunchKernelGGL((linearize_blocks_kernel<false, 1>), grid, dim3(256), lds, st, dp, dq, sv, sp, ntake); }
  *done = true;
  return hipGetLastError();
}
hipError_t launch_slot_xy(const DeviceProblem& dp, double2* slot_xy, hipStream_t st) {
  if (dp.N > 0) LAUNCH(slot_xy_kernel, nblocks256(dp.N), 256, st, dp, slot_xy);
  return hipSuccess;
}
hipError_t launch_point_blocks(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  if (sv.slot_xy) {   // calibrated: from the observations themselves
    if (dp.M <= 0) return hipSuccess;
    const size_t lds = (size_t)4 * 64 * 9 * sizeof(double);
    const int sp = sweep_points(dp.M), grid = (int)((dp.M + 4 * sp - 1) / (4 * sp));
    if (dp.calibrated) { if (sv.CD == 12) hipLaunchKernelGGL((point_blocks_rc_kernel<true, 2>), dim3(grid), dim3(256), lds, st, dp, sv, sp); else hipLaunchKernelGGL((point_blocks_rc_kernel<true, 1>), dim3(grid), dim3(256), lds, st, dp, sv, sp); }
    else { if (sv.CD == 12) hipLaunchKernelGGL((point_blocks_rc_kernel<false, 2>), dim3(grid), dim3(256), lds, st, dp, sv, sp); else hipLaunchKernelGGL((point_blocks_rc_kernel<false, 1>), dim3(grid), dim3(256), lds, st, dp, sv, sp); }
    return hipGetLastError();
  }
  LAUNCH(point_blocks_kernel, nblocks256(dp.M), 256, st, dp, sv);
  return hipSuccess;
}
hipError_t launch_jacobi_scale(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  LAUNCH(jacobi_scale_kernel, nblocks256(sv.n + 3 * (int64_t)dp.M), 256, st, dp, sv);
  return hipSuccess;
}
hipError_t launch_clamp_diagonal(const DeviceProblem& dp, const SolverDev& sv, double lo, double hi, hipStream_t st) {
  LAUNCH(clamp_diagonal_kernel, nblocks256(sv.n + 3 * (int64_t)dp.M), 256, st, dp, sv, lo, hi);
  return hipSuccess;
}
hipError_t launch_gradient_max(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  const int nb = nblocks256(sv.n + 3 * (int64_t)dp.M);
  LAUNCH(gradient_max_kernel, nb, 256, st, dp, sv, 0);
  LAUNCH(reduce_max_kernel, 1, 256, st, sv.partial, nb, sv.scalars + kGradMax, nullptr, 0);
  return hipSuccess;
}
// several ranks: max |g| over this rank's own points -> its slot behind the camera exchange's payload (xbuf[2n + 3 + rank]; the other
// ranks' slots are zeroed by launch_pack_linearize, which runs first: a SUM all-reduce then carries every rank's maximum) ...
hipError_t launch_gradient_max_points(const DeviceProblem& dp, const SolverDev& sv, int rank, hipStream_t st) {
  const int nb = std::max(1, nblocks256(3 * (int64_t)dp.M));
  LAUNCH(gradient_max_kernel, nb, 256, st, dp, sv, 1);
  LAUNCH(reduce_max_kernel, 1, 256, st, sv.partial, nb, sv.xbuf + 2 * sv.n + 3 + rank, nullptr, 0);
  return hipSuccess;
}
// ... and behind the exchange: the cameras' maximum from the summed gradient, with the ranks' point maxima -> scalars[kGradMax]
hipError_t launch_gradient_max_cameras(const DeviceProblem& dp, const SolverDev& sv, int world, hipStream_t st) {
  const int nb = std::max(1, nblocks256(sv.n));
  LAUNCH(gradient_max_kernel, nb, 256, st, dp, sv, 2);
  LAUNCH(reduce_max_kernel, 1, 256, st, sv.partial, nb, sv.scalars + kGradMax, sv.xbuf + 2 * sv.n + 3, world);
  return hipSuccess;
}
hipError_t launch_pack_linearize(const DeviceProblem& dp, const SolverDev& sv, const double* cost2, hipStream_t st, int nslots) {
  LAUNCH(pack_linearize_kernel, nblocks256(std::max<int64_t>(sv.n, nslots)), 256, st, dp, sv, cost2, nslots);
  return hipSuccess;
}
__global__ void begin_solve_kernel(const SolverDev sv) { *sv.chol_fail = 0; sv.scalars[kDagSuspect] = 0.0; }
hipError_t launch_begin_solve(const SolverDev& sv, hipStream_t st) {
  LAUNCH(begin_solve_kernel, 1, 1, st, sv);
  return hipSuccess;
}
hipError_t launch_local_linearize(const DeviceProblem& dp, const SolverDev& sv, const double* cost2, hipStream_t st) {
  LAUNCH(local_linearize_kernel, nblocks256(sv.n), 256, st, dp, sv, cost2);
  return hipSuccess;
}
hipError_t launch_unpack_linearize(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  LAUNCH(unpack_linearize_kernel, nblocks256(sv.n), 256, st, dp, sv);
  return hipSuccess;
}
hipError_t launch_pack_trial(const DeviceProblem& dp, const SolverDev& sv, const double* cost2, hipStream_t st) {
  LAUNCH(pack_trial_kernel, 1, 1, st, dp, sv, cost2);
  return hipSuccess;
}
hipError_t launch_unscaled_gradient(const DeviceProblem& dp, const SolverDev& sv, double* g_pose, double* g_point, hipStream_t st) {
  LAUNCH(unscaled_gradient_kernel, nblocks256(sv.n + 3 * (int64_t)dp.M), 256, st, dp, sv, g_pose, g_point);
  return hipSuccess;
}
hipError_t launch_point_factor(const DeviceProblem& dp, const SolverDev& sv, double radius, hipStream_t st, const double* clamp) {
  if (clamp) LAUNCH(point_factor_kernel<true>, nblocks256(std::max<int64_t>(dp.M, sv.n)), 256, st, dp, sv, 1.0 / radius, clamp[0], clamp[1]);
  else LAUNCH(point_factor_kernel<false>, nblocks256(dp.M), 256, st, dp, sv, 1.0 / radius, 0.0, 0.0);
  return hipSuccess;
}
template <int CD, int KC>
static hipError_t launch_project_as(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  constexpr int REC = 8 + 2 * KC, OUT = CD * 3;
  const size_t lds = (size_t)4 * (64 * ((REC > OUT ? REC : OUT) | 1) + 32) * sizeof(double);
  const int grid = (int)((dp.N + 256 * kProjectChunks - 1) / (256 * kProjectChunks));
  hipError_t e = allow_dynamic_lds(project_kernel<CD, KC>, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((project_kernel<CD, KC>), dim3(grid), dim3(256), lds, st, dp, sv);
  return hipGetLastError();
}
// does launch_project do the virtual records too?  (decided with the plan, SolverDev::fused_sweep: one shared intrinsics block, recomputed
// records, two-pose frames all in factored tiles; RSBA_NO_FUSED_SWEEP=1 keeps the two passes apart)
bool project_covers_virtual_records(const DeviceProblem& dp, const SolverDev& sv) { return sv.fused_sweep != 0 && dp.N > 0; }
hipError_t launch_project(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  if (dp.N == 0) return hipSuccess;
  if (project_covers_virtual_records(dp, sv)) {
    const size_t lds = (size_t)4 * kFusedWaveDoubles * sizeof(double);
    const int sp = sweep_points(dp.M), grid = (int)((dp.M + 4 * sp - 1) / (4 * sp));
    hipLaunchKernelGGL(virtual_project_rc_kernel, dim3(grid), dim3(256), lds, st, dp, sv, sp);
    return hipGetLastError();
  }
  const int KC = dp.K - 3;
  if (sv.slot_xy) {
    const int CD = sv.CD;
    const bool allf = CD == 12 && sv.all_real_factored != 0;
    const size_t lds = (size_t)4 * (64 * (allf ? 19 : ((CD * 3) | 1)) + 32) * sizeof(double);
    const int nch = project_chunks(dp.N), grid = (int)((dp.N + 256 * (int64_t)nch - 1) / (256 * (int64_t)nch));
    if (allf) { if (dp.calibrated) hipLaunchKernelGGL((project_rc_kernel<true, 2, true>), dim3(grid), dim3(256), lds, st, dp, sv, nch); else hipLaunchKernelGGL((project_rc_kernel<false, 2, true>), dim3(grid), dim3(256), lds, st, dp, sv, nch); }
    else if (dp.calibrated) { if (CD == 12) hipLaunchKernelGGL((project_rc_kernel<true, 2, false>), dim3(grid), dim3(256), lds, st, dp, sv, nch); else hipLaunchKernelGGL((project_rc_kernel<true, 1, false>), dim3(grid), dim3(256), lds, st, dp, sv, nch); }
    else { if (CD == 12) hipLaunchKernelGGL((project_rc_kernel<false, 2, false>), dim3(grid), dim3(256), lds, st, dp, sv, nch); else hipLaunchKernelGGL((project_rc_kernel<false, 1, false>), dim3(grid), dim3(256), lds, st, dp, sv, nch); }
    return hipGetLastError();
  }
  if (sv.CD == 12 && KC == 12) return launch_project_as<12, 12>(dp, sv, st);
  if (sv.CD == 6 && KC == 6) return launch_project_as<6, 6>(dp, sv, st);
  if (sv.CD == 12 && KC == 21) return launch_project_as<12, 21>(dp, sv, st);
  if (sv.CD == 6 && KC == 15) return launch_project_as<6, 15>(dp, sv, st);
  return hipErrorInvalidValue;
}
// intrinsics as a parameter block: the self block and the intrinsics gradient, summed over the frames
hipError_t launch_intr_blocks(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  if (sv.NPF == 0) return hipSuccess;
  LAUNCH(intr_reduce_kernel, 54 * sv.NIB, 256, st, dp, sv);
  return hipSuccess;
}
hipError_t launch_virtual_records(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  if (sv.NPF == 0) return hipSuccess;
  if (sv.nvgroups == 0) return hipSuccess;
  if (project_covers_virtual_records(dp, sv)) return hipSuccess;   // (launch_project's fused sweep wrote them)
  if (sv.slot_xy) {   // (one intrinsics block: recomputed like the rest)
    const size_t lds = (size_t)4 * 64 * 9 * sizeof(double);   // (nine of the 27 components at a time: point_sweep)
    const int sp = sweep_points(dp.M), grid = (int)((dp.M + 4 * sp - 1) / (4 * sp));
    if (sv.CD == 12) hipLaunchKernelGGL(virtual_records_rc_kernel<2>, dim3(grid), dim3(256), lds, st, dp, sv, sp);
    else hipLaunchKernelGGL(virtual_records_rc_kernel<1>, dim3(grid), dim3(256), lds, st, dp, sv, sp);
    return hipGetLastError();
  }
  if (sv.CD == 12) LAUNCH(virtual_records_kernel<12>, nblocks256(sv.nvgroups), 256, st, dp, sv);
  else LAUNCH(virtual_records_kernel<6>, nblocks256(sv.nvgroups), 256, st, dp, sv);
  return hipSuccess;
}
hipError_t launch_clear_system(const SolverDev& sv, hipStream_t st) {
  return hipMemsetAsync(sv.S, 0, (size_t)sv.nslots * kTile * kTile * sizeof(double), st);
}
hipError_t launch_schur_blocks(const DeviceProblem& dp, const SolverDev& sv, double radius, hipStream_t st) {
  if (sv.nchunk > 0) {
    // dynamic LDS: max(slot table of a chunk, four partial tiles + rhs partials)
    const size_t table = (size_t)kSchurOffBytes + kSchurMskBytes + (size_t)kSchurChunk * 3 * sizeof(double);
    const size_t tiles = (size_t)(4 * kTile * (kTile + 1) + 4 * kTile) * sizeof(double);
    const size_t lds = table > tiles ? table : tiles;
    const int per_xcd = (sv.nchunk + 7) / 8;
    const int persistent = sv.schur_variant != 2 && !sv.schur_linear;
    if (sv.schur_variant == 1) {   // (RSBA_SCHUR_VARIANT=1: three groups in flight, one wave per SIMD; 2: one workgroup per chunk; 4 / 5: ablations of variant 0)
      const dim3 grid(8 * (persistent ? std::min(per_xcd, 32) : per_xcd));
      hipError_t e = allow_dynamic_lds(schur_tile_kernel<3, 1>, lds); if (e != hipSuccess) return e;
      hipLaunchKernelGGL((schur_tile_kernel<3, 1>), grid, dim3(256), lds, st, sv, sv.Pm, sv.z, persistent);
    } else {
      const dim3 grid(8 * (persistent ? std::min(per_xcd, 64) : per_xcd));   // (32 CUs per XCD, two of these workgroups per CU)
      hipError_t e = allow_dynamic_lds(schur_tile_kernel<2, 2>, lds); if (e != hipSuccess) return e;
      hipLaunchKernelGGL((schur_tile_kernel<2, 2>), grid, dim3(256), lds, st, sv, sv.Pm, sv.z, persistent);
    }
    { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; }
  }
  if (sv.npremerge > 0) LAUNCH(schur_premerge_kernel, dim3(sv.npremerge, kPremergeSplit), 256, st, sv);
  LAUNCH(schur_merge_kernel, dim3(sv.ntp, kMergeSplit), 256, st, dp, sv, 1.0 / radius);
  return hipSuccess;
}
static int point_step_blocks(const DeviceProblem& dp, const SolverDev& sv) { const int sp = sweep_points(dp.M); return sv.slot_xy ? (int)((dp.M + 4 * sp - 1) / (4 * sp)) : (int)((dp.M + 255) / 256); }
template <int CD, int KC>
static hipError_t launch_point_step(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  constexpr int REC = 8 + 2 * KC, PITCH = REC | 1;
  const size_t lds = (size_t)4 * (64 * PITCH + 64 * 5) * sizeof(double);
  hipError_t e = allow_dynamic_lds(point_step_kernel<CD, KC>, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((point_step_kernel<CD, KC>), dim3(point_step_blocks(dp, sv)), dim3(256), lds, st, dp, sv);
  return hipGetLastError();
}
// point steps y_p and the per-workgroup partials of the model cost change (sv.partial)
hipError_t launch_back_substitute(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  if (dp.M <= 0) return hipSuccess;
  const int KC = dp.K - 3;
  if (sv.slot_xy) {
    const size_t lds = (size_t)4 * 64 * 5 * sizeof(double);
    const int grid = point_step_blocks(dp, sv);
    const int sp = sweep_points(dp.M);
    if (dp.calibrated) { if (sv.CD == 12) hipLaunchKernelGGL((point_step_rc_kernel<true, 2>), dim3(grid), dim3(256), lds, st, dp, sv, sp); else hipLaunchKernelGGL((point_step_rc_kernel<true, 1>), dim3(grid), dim3(256), lds, st, dp, sv, sp); }
    else { if (sv.CD == 12) hipLaunchKernelGGL((point_step_rc_kernel<false, 2>), dim3(grid), dim3(256), lds, st, dp, sv, sp); else hipLaunchKernelGGL((point_step_rc_kernel<false, 1>), dim3(grid), dim3(256), lds, st, dp, sv, sp); }
    return hipGetLastError();
  }
  if (sv.CD == 12) return KC == 12 ? launch_point_step<12, 12>(dp, sv, st) : launch_point_step<12, 21>(dp, sv, st);
  return KC == 6 ? launch_point_step<6, 6>(dp, sv, st) : launch_point_step<6, 15>(dp, sv, st);
}
// -> scalars[kModelCostChange]; must follow launch_back_substitute directly (it reduces that kernel's partials)
hipError_t launch_model_cost_change(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  LAUNCH(reduce_sum_kernel, 1, 256, st, sv.partial, dp.M > 0 ? point_step_blocks(dp, sv) : 0, sv.scalars + kModelCostChange, -1.0);
  return hipSuccess;
}
hipError_t launch_own_points(const DeviceProblem& dp, const SolverDev& sv, double* buf, hipStream_t st) {
  LAUNCH(own_points_kernel, nblocks256(dp.M), 256, st, dp, sv, buf);
  return hipSuccess;
}
hipError_t launch_merge_points(const DeviceProblem& dp, const double* buf, hipStream_t st) {
  LAUNCH(merge_points_kernel, nblocks256(dp.M), 256, st, dp, buf);
  return hipSuccess;
}
hipError_t launch_lm_verdict_step(const DeviceProblem& dp, const SolverDev& sv, double* cost2, double* ctl, const LmRules& rules, rsba_iteration* trace, int trace_cap, hipStream_t st, bool cost_reduced) {
  LAUNCH(lm_verdict_step_kernel, 1, 256, st, dp, sv, cost2, cost_reduced ? -1 : eval_num_blocks(dp.N), ctl, rules, trace, trace_cap);
  return hipSuccess;
}
hipError_t launch_lm_linearize_gradient(const DeviceProblem& dp, const SolverDev& sv, const double* cost2, hipStream_t st) {
  LAUNCH(lm_linearize_gradient_kernel, nblocks256(sv.n + 3 * (int64_t)dp.M), 256, st, dp, sv, cost2);
  return hipSuccess;
}
hipError_t launch_lm_verdict_gradient(const DeviceProblem& dp, const SolverDev& sv, double* ctl, const LmRules& rules, rsba_iteration* trace, int trace_cap, double* snapshot, double seq, hipStream_t st, bool gradmax_done, int extra_partials) {
  LAUNCH(lm_verdict_gradient_kernel, 1, 256, st, sv, gradmax_done ? -1 : nblocks256(sv.n + 3 * (int64_t)dp.M) + extra_partials, ctl, rules, trace, trace_cap, snapshot, seq);
  return hipSuccess;
}
hipError_t launch_lm_decide_step(const SolverDev& sv, double* ctl, const LmRules& rules, rsba_iteration* trace, int trace_cap, hipStream_t st) {
  LAUNCH(lm_decide_step_kernel, 1, 1, st, sv, ctl, rules, trace, trace_cap);
  return hipSuccess;
}
hipError_t launch_lm_decide_gradient(const SolverDev& sv, double* ctl, const LmRules& rules, rsba_iteration* trace, int trace_cap, hipStream_t st) {
  LAUNCH(lm_decide_gradient_kernel, 1, 1, st, sv, ctl, rules, trace, trace_cap);
  return hipSuccess;
}
hipError_t launch_lm_take_candidate(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  const int64_t npose = (int64_t)dp.F * dp.P * 6, npoint = 3 * (int64_t)dp.M, nintr = sv.NPF > 0 ? 9 * (int64_t)dp.NI : 0;
  LAUNCH(lm_take_candidate_kernel, (unsigned)((npose + npoint + nintr + 255) / 256), 256, st, dp, sv, npose, npoint, nintr);
  return hipSuccess;
}
hipError_t launch_candidate(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  const int nb = nblocks256(sv.n + 3 * (int64_t)dp.M);
  LAUNCH(candidate_kernel, nb, 256, st, dp, sv, sv.partial);
  LAUNCH(reduce_sum2_kernel, 2, 256, st, sv.partial, nb, sv.scalars + kStepSq, sv.scalars + kXSq);   // (one launch: workgroup 0 -> |step|^2, workgroup 1 -> |x|^2)
  return hipSuccess;
}
hipError_t launch_candidate_and_model_cost(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st) {
  const int nb = nblocks256(sv.n + 3 * (int64_t)dp.M);
  LAUNCH(candidate_kernel, nb, 256, st, dp, sv, sv.partial_c);   // (its sums beside the back-substitution's, which are still waiting in sv.partial)
  LAUNCH(reduce_sum3_kernel, 3, 256, st, sv.partial, dp.M > 0 ? point_step_blocks(dp, sv) : 0, sv.scalars + kModelCostChange, sv.partial_c, nb, sv.scalars + kStepSq, sv.scalars + kXSq);
  return hipSuccess;
}

// exchange (2) of a sharded solve moves only the tiles of S that can be non-zero — the tile pairs of the plan; the fill-in tiles of the
// factor's layout (44 % of the packed tiles at 1k cameras) are zero on every rank.  pack: buf[b] = S tile slots[b], then the rhs;
// unpack: the reverse.  One workgroup per tile, the last one takes the rhs.
namespace {
// ---- sharded factorisation (solver.hip: solve_reduced_system): the exchange between its two launches ----
// Every rank has factored the columns of its own part; what its part subtracts from the separators' tiles sits in the partial tiles
// of its UPDATE items (cells of the persistent Cholesky, complete: the launch is over).  Separator tile t of this rank's share:
//   buf[t] = S_t (its partial from its own points; zero for a fill-only tile) - sum of ITS partial tiles, in list order
// and for a diagonal tile the same for its rows of the right-hand side — the forward solve rides along.  The all-reduce of buf over
// the ranks is the separators' system with every part eliminated; top_unpack puts it where the second launch reads S and rhs.
__global__ __launch_bounds__(256) void top_assemble_kernel(const SolverDev sv, const int32_t* __restrict__ slots, const int32_t* __restrict__ info,
                                                           const int32_t* __restrict__ asm_ptr, const int32_t* __restrict__ asm_list, const int32_t* __restrict__ top_tiles,
                                                           int ntop_slots, double* __restrict__ buf) {
  const int t = blockIdx.x, tid = threadIdx.x;
  const int slot = slots[t], has_pair = info[2 * t], rhs_row = info[2 * t + 1];
  const int p0 = asm_ptr[t], p1 = asm_ptr[t + 1];
  const double* S = sv.S + (size_t)slot * (kTile * kTile);
  constexpr size_t pstride = kTile * kTile + kTile;
  for (int e = tid; e < kTile * kTile; e += 256) {
    double v = has_pair ? S[e] : 0.0;
    for (int p = p0; p < p1; ++p) v -= sv.chol_part[(size_t)asm_list[p] * pstride + e];
    buf[(size_t)t * (kTile * kTile) + e] = v;
  }
  if (rhs_row >= 0 && tid < kTile) {
    double v = sv.rhs[(size_t)top_tiles[rhs_row] * kTile + tid];
    for (int p = p0; p < p1; ++p) v -= sv.chol_part[(size_t)asm_list[p] * pstride + kTile * kTile + tid];
    buf[(size_t)ntop_slots * (kTile * kTile) + (size_t)rhs_row * kTile + tid] = v;
  }
}
__global__ __launch_bounds__(256) void top_unpack_kernel(const SolverDev sv, const int32_t* __restrict__ slots, const int32_t* __restrict__ info, const int32_t* __restrict__ top_tiles,
                                                         int ntop_slots, const double* __restrict__ buf) {
  const int t = blockIdx.x, tid = threadIdx.x;
  double* S = sv.S + (size_t)slots[t] * (kTile * kTile);
  for (int e = tid; e < kTile * kTile; e += 256) S[e] = buf[(size_t)t * (kTile * kTile) + e];
  const int rhs_row = info[2 * t + 1];
  if (rhs_row >= 0 && tid < kTile) sv.rhs[(size_t)top_tiles[rhs_row] * kTile + tid] = buf[(size_t)ntop_slots * (kTile * kTile) + (size_t)rhs_row * kTile + tid];
}
// the gather of the camera step: ybuf = y on the rows this rank contributes (its part; rank 0: the separators), zero elsewhere — summed over the ranks
__global__ void step_rows_kernel(const double* __restrict__ yv, const uint8_t* __restrict__ row_mine, int64_t npad, double* __restrict__ ybuf) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < npad) ybuf[i] = row_mine[i / kTile] ? yv[i] : 0.0;
}
__global__ void zero_tiles_kernel(double* __restrict__ S, const int32_t* __restrict__ slots) {
  double* t = S + (size_t)slots[blockIdx.x] * (kTile * kTile);
  for (int e = threadIdx.x; e < kTile * kTile; e += 256) t[e] = 0.0;
}
template <bool UNPACK>
__global__ __launch_bounds__(256) void exchange_pack_kernel(SolverDev sv, const int32_t* __restrict__ slots, int ntiles, double* __restrict__ buf) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b < ntiles) {
    double* t = sv.S + (size_t)slots[b] * (kTile * kTile);
    double* q = buf + (size_t)b * (kTile * kTile);
#pragma unroll
    for (int k = 0; k < (kTile * kTile) / 256; ++k) { if (UNPACK) t[tid + 256 * k] = q[tid + 256 * k]; else q[tid + 256 * k] = t[tid + 256 * k]; }
  } else {
    double* q = buf + (size_t)ntiles * (kTile * kTile);
    for (int64_t e = tid; e < sv.npad; e += 256) { if (UNPACK) sv.rhs[e] = q[e]; else q[e] = sv.rhs[e]; }
  }
}
}  // namespace
hipError_t launch_top_assemble(const SolverDev& sv, const int32_t* slots, const int32_t* info, const int32_t* asm_ptr, const int32_t* asm_list, const int32_t* top_tiles, int ntop_slots, double* buf, hipStream_t st) {
  if (ntop_slots > 0) LAUNCH(top_assemble_kernel, ntop_slots, 256, st, sv, slots, info, asm_ptr, asm_list, top_tiles, ntop_slots, buf);
  return hipSuccess;
}
hipError_t launch_top_unpack(const SolverDev& sv, const int32_t* slots, const int32_t* info, const int32_t* top_tiles, int ntop_slots, const double* buf, hipStream_t st) {
  if (ntop_slots > 0) LAUNCH(top_unpack_kernel, ntop_slots, 256, st, sv, slots, info, top_tiles, ntop_slots, buf);
  return hipSuccess;
}
hipError_t launch_step_rows(const double* yv, const uint8_t* row_mine, int64_t npad, double* ybuf, hipStream_t st) {
  LAUNCH(step_rows_kernel, (unsigned)((npad + 255) / 256), 256, st, yv, row_mine, npad, ybuf);
  return hipSuccess;
}
hipError_t launch_zero_tiles(double* S, const int32_t* slots, int n, hipStream_t st) {
  if (n > 0) LAUNCH(zero_tiles_kernel, n, 256, st, S, slots);
  return hipSuccess;
}

hipError_t launch_exchange_pack(const SolverDev& sv, const int32_t* slots, int ntiles, double* buf, bool unpack, hipStream_t st) {
  if (unpack) hipLaunchKernelGGL(exchange_pack_kernel<true>, dim3(ntiles + 1), dim3(256), 0, st, sv, slots, ntiles, buf);
  else hipLaunchKernelGGL(exchange_pack_kernel<false>, dim3(ntiles + 1), dim3(256), 0, st, sv, slots, ntiles, buf);
  return hipGetLastError();
}

}  // namespace rsba
