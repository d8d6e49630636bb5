// ORACLE — TEST INFRASTRUCTURE ONLY (see rsba_oracle_math.hpp).  PARITY UNPINNED beyond mat_test.cc.
//
// Problem-level restatement: what ceres::Problem / ceres::Solve do with the residual blocks that
// CeresHandler::Add creates (CeresHandler.h:208-301, :335-382) and CeresHandler::solve runs
// (CeresHandler.h:394-426).  Ceres-Solver 1.9.0 is an un-vendored dependency (README.md:3,
// .travis.yml:32-38); its behaviour is restated from its published algorithm.
#include "rsba_oracle.h"
#include "rsba_oracle_math.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace rsba_oracle;

namespace {

struct Layout {
  bool cal; int P, CD, K;
  int off_cam, off_pose, off_point;   // column offsets inside one block row
};
Layout layout_of(const orc_problem* p) {
  Layout L; L.cal = p->calibrated != 0; L.P = p->poses_per_frame; L.CD = 6 * L.P;
  L.off_cam = 0; L.off_pose = L.cal ? 0 : 9; L.off_point = L.off_pose + L.CD; L.K = L.off_point + 3;
  return L;
}
inline int intr_of(const orc_problem* p, int f) { return p->frame_intrinsics ? p->frame_intrinsics[f] : 0; }

// ceres::AutoDiffCostFunction<Functor,2,...>::Evaluate restated for the four functor shapes
// (VideoSfmBaRs.h:53-80, video_bundler_free.h:70-91): one pass with Dual<K>, seeds in block order.
template <bool CAL, int P>
bool block_eval(const orc_problem* p, int64_t i, double* r, double* J /* [2][K] or null */) {
  constexpr int K = (CAL ? 0 : 9) + 6 * P + 3;
  const int f = p->obs_frame[i], j = p->obs_point[i];
  const double* cam = p->intrinsics + 9 * intr_of(p, f);
  const double* pose = p->poses + (size_t)f * 6 * P;
  const double* X = p->points + (size_t)j * 3;
  const double ox = p->obs_xy[2 * i], oy = p->obs_xy[2 * i + 1];
  if (!J) {
    double rr[2];
    bool ok;
    if (P == 2 && !(p->frame_global && p->frame_global[f])) ok = rs_residual<double>(cam, pose, pose + 6, X, ox, oy, p->shutter, p->scanlines, p->interpolate_rotation != 0, rr, !p->no_validate);
    else ok = gs_residual<double>(cam, pose, X, ox, oy, rr, !p->no_validate);   // (a one-pose frame of a two-pose session: CeresHandler.h:266-285)
    if (ok) { r[0] = rr[0]; r[1] = rr[1]; }
    return ok;
  }
  typedef Dual<K> D;
  D dcam[9], dpose[6 * P], dX[3], res[2];
  int col = 0;
  for (int k = 0; k < 9; ++k) dcam[k] = CAL ? D(cam[k]) : D(cam[k], col++);
  for (int k = 0; k < 6 * P; ++k) dpose[k] = D(pose[k], col++);
  for (int k = 0; k < 3; ++k) dX[k] = D(X[k], col++);
  bool ok;
  if (P == 2 && !(p->frame_global && p->frame_global[f])) ok = rs_residual<D>(dcam, dpose, dpose + 6, dX, ox, oy, p->shutter, p->scanlines, p->interpolate_rotation != 0, res, !p->no_validate);
  else ok = gs_residual<D>(dcam, dpose, dX, ox, oy, res, !p->no_validate);
  if (!ok) return false;
  r[0] = res[0].a; r[1] = res[1].a;
  for (int k = 0; k < K; ++k) { J[k] = res[0].v[k]; J[K + k] = res[1].v[k]; }
  return true;
}

// RsConstVeloPrior::operator() (video_bundler_rs_inter.h:63-93) and RsConstAccelerationPrior::operator()
// (:121-158) restated, coordinate by coordinate in the reference's order of operations.  cur0/cur1 = the two poses
// of frame f, prev0/prev1 = those of frame f-1.  Returns the functor's validity flag.
template <class T>
bool motion_prior(int kind, const T& ratio, double scale, const T* cur0, const T* cur1, const T* prev0, const T* prev1, T* res) {
  for (int i = 0; i < 6; ++i) {
    if (kind == 1) {
      // constant velocity: frame start against prev1 + ratio (prev1 - prev0) ...
      T d = prev1[i] - prev0[i];
      d = d * ratio;
      d = prev1[i] + d;
      res[i] = cur0[i] - d;
      // ... and frame end against cur0 + (cur0 - prev1) / ratio (ratio > eps), else cur0 + (prev1 - prev0)
      T e;
      if (ratio > kEps) { e = cur0[i] - prev1[i]; e = e * (T(1.0) / ratio); }
      else e = prev1[i] - prev0[i];
      e = cur0[i] + e;
      res[6 + i] = cur1[i] - e;
    } else {
      // constant acceleration: x' = x_-1 + v_-1 t + a t^2 / 2 with a t^2 = v t - v_-1 t
      T vt = cur0[i] - prev1[i];
      T v1t = (prev1[i] - prev0[i]) * ratio;
      T h = (vt - v1t) * T(0.5);
      h = v1t + h;
      h = prev1[i] + h;
      res[i] = cur0[i] - h;
      T v = cur1[i] - cur0[i];
      T w = (cur0[i] - prev1[i]) * (T(1.0) / ratio);
      T g = (v - w) * T(0.5);
      g = w + g;
      g = cur0[i] + g;
      res[6 + i] = cur1[i] - g;
    }
  }
  for (int i = 0; i < 12; ++i) res[i] = res[i] * T(scale);
  for (int i = 0; i < 3; ++i) { res[i] = res[i] * T(0.01); res[6 + i] = res[6 + i] * T(0.01); }   // down-scaled rotation rows
  return kind == 1 ? !(ratio < 0.0) : !(ratio < kEps);
}

// AutoDiffCostFunction<Rs...Prior,12,1,6,6,6,6>::Evaluate: r[12], J[12][25] over the columns
// [f.poses[0] | f.poses[1] | f-1.poses[0] | f-1.poses[1] | interFrameRatio]; the last column is seeded only when the
// ratio block is a free parameter (ratio_free), and stays zero (null Jacobian of a constant block) otherwise
constexpr int kPC = 25;
bool prior_eval(const orc_problem* p, int k, double* r, double* J) {
  const int f = p->prior_frames[k];
  const double* cur = p->poses + (size_t)f * 12; const double* prev = p->poses + (size_t)(f - 1) * 12;
  if (!J) return motion_prior<double>(p->prior_kind, p->inter_frame_ratio, p->prior_scale, cur, cur + 6, prev, prev + 6, r);
  typedef Dual<kPC> D;
  D x[24], res[12];
  for (int c = 0; c < 12; ++c) { x[c] = D(cur[c], c); x[12 + c] = D(prev[c], 12 + c); }
  const D ratio = p->ratio_free ? D(p->inter_frame_ratio, 24) : D(p->inter_frame_ratio);
  if (!motion_prior<D>(p->prior_kind, ratio, p->prior_scale, x, x + 6, x + 12, x + 18, res)) return false;
  for (int i = 0; i < 12; ++i) { r[i] = res[i].a; for (int c = 0; c < kPC; ++c) J[i * kPC + c] = res[i].v[c]; }
  return true;
}

// GoodPosePrior::operator() (CeresHandler.h:58-64): minus6(pose0, pose), rotation rows times `rotation`, position rows times
// `position`; valid while residuals[0] < 1.  SphericalPrior::operator() (:40-44).
template <class T>
bool good_pose_prior(double rotation, double position, const T* pose0, const T* pose, T* res) {
  for (int i = 0; i < 6; ++i) res[i] = pose0[i] - pose[i];
  for (int i = 0; i < 3; ++i) res[i] = res[i] * T(rotation);
  for (int i = 3; i < 6; ++i) res[i] = res[i] * T(position);
  return res[0] < T(1.0);
}
template <class T>
bool spherical_prior(const T* pose, T* res) {
  res[0] = (pose[0] * pose[0] + pose[1] * pose[1] + pose[2] * pose[2]);
  res[1] = T(1e20) * (T(1.0) - abs(pose[3]) - abs(pose[4]) - abs(pose[5]));
  return res[0] < T(1.0);
}

typedef bool (*block_fn)(const orc_problem*, int64_t, double*, double*);
block_fn pick_block_fn(const Layout& L) {
  if (L.cal) return L.P == 2 ? block_eval<true, 2> : block_eval<true, 1>;
  return L.P == 2 ? block_eval<false, 2> : block_eval<false, 1>;
}

constexpr int64_t kSumChunk = 4096;   // observations per partial sum of the parallel reductions (fixed: results do not depend on the team size)

int threads_of(int n) {
#ifdef _OPENMP
  return n > 0 ? n : omp_get_max_threads();
#else
  (void)n; return 1;
#endif
}

// ---- "the Ceres way": one heap object per residual block with a virtual Evaluate ----------------
struct CostObject {
  virtual ~CostObject() {}
  virtual bool Evaluate(double const* const* params, double* residuals, double** jacobians) const = 0;
};
template <bool CAL, int P>
struct BlockCost : CostObject {
  // video_bundler_free.h:25-29 copies the 9 intrinsics into every functor; VideoSfmBaRs.h:82-83 keeps
  // references to the session / options.
  double ox, oy, cam[9];
  const orc_problem* sess;
  BlockCost(const orc_problem* p, int64_t i) : ox(p->obs_xy[2 * i]), oy(p->obs_xy[2 * i + 1]), sess(p) {
    std::memcpy(cam, p->intrinsics + 9 * intr_of(p, p->obs_frame[i]), sizeof cam);
  }
  bool Evaluate(double const* const* params, double* residuals, double** jac) const override {
    constexpr int K = (CAL ? 0 : 9) + 6 * P + 3;
    typedef Dual<K> D;
    D dcam[9], dpose[6 * P], dX[3], res[2];
    int col = 0, b = 0;
    if (CAL) { for (int k = 0; k < 9; ++k) dcam[k] = D(cam[k]); }
    else { for (int k = 0; k < 9; ++k) dcam[k] = D(params[b][k], col++); ++b; }
    for (int q = 0; q < P; ++q, ++b) for (int k = 0; k < 6; ++k) dpose[6 * q + k] = D(params[b][k], col++);
    for (int k = 0; k < 3; ++k) dX[k] = D(params[b][k], col++);
    bool ok;
    if (P == 2) ok = rs_residual<D>(dcam, dpose, dpose + 6, dX, ox, oy, sess->shutter, sess->scanlines, sess->interpolate_rotation != 0, res);
    else ok = gs_residual<D>(dcam, dpose, dX, ox, oy, res);
    if (!ok) return false;
    residuals[0] = res[0].a; residuals[1] = res[1].a;
    // per-block row-major 2 x Ni arrays
    col = 0; b = 0;
    const int sizes[4] = {CAL ? 0 : 9, 6, P == 2 ? 6 : 0, 3};
    for (int s = 0; s < 4; ++s) {
      if (!sizes[s]) continue;
      if (jac && jac[b]) for (int k = 0; k < sizes[s]; ++k) { jac[b][k] = res[0].v[col + k]; jac[b][sizes[s] + k] = res[1].v[col + k]; }
      col += sizes[s]; ++b;
    }
    return true;
  }
};

// ---- loss-corrected, masked evaluation shared by orc_evaluate / orc_normal_equations / orc_solve --
struct Eval {
  const orc_problem* p; Layout L; int64_t N; int F, M, NI;
  std::vector<double> r, J;           // corrected residuals [N][2], Jacobian [N][2][K]
  std::vector<uint8_t> dropped;       // residual block with every parameter block constant
  std::vector<uint8_t> colmask;       // per parameter column (global numbering) 1 = fixed
  int64_t ncam, nparam;               // camera-side unknowns (poses + intrinsics), all unknowns
  // motion-prior blocks: corrected residuals [NP][12], Jacobian [NP][12][24], dropped flags
  int NP = 0, PC = 24; int64_t iratio = -1; std::vector<double> pr, pJ; std::vector<uint8_t> pdropped;   // PC 25 / iratio >= 0: free interFrameRatio
  // per-pose prior blocks (GoodPosePrior: columns [priorPoses block 6 | pose block 6], 6 residuals; SphericalPrior: [pose
  // block 6], 2 residuals), no loss function.  The priorPoses blocks are camera-side unknowns behind everything else.
  struct Extra { int nres, ncol; int64_t col[12]; double r[6], J[6 * 12]; bool dropped; int kind, which; };
  std::vector<Extra> ex; int64_t ipp = -1;
  // global column of local column c of prior k
  inline int64_t pcol(int k, int c) const { const int f = p->prior_frames[k]; return c == 24 ? iratio : c < 12 ? (int64_t)f * 12 + c : (int64_t)(f - 1) * 12 + (c - 12); }
  Eval(const orc_problem* pp) : p(pp), L(layout_of(pp)) {
    N = p->num_observations; F = p->num_frames; M = p->num_points; NI = p->num_intrinsics;
    ncam = (int64_t)F * L.CD + (L.cal ? 0 : (int64_t)NI * 9);
    NP = (p->prior_kind != 0 && L.P == 2) ? p->num_priors : 0;
    if (NP > 0 && p->ratio_free) { iratio = ncam; ncam += 1; PC = 25; }   // the ratio is one more camera-side unknown
    if (p->num_pose_priors > 0) { ipp = ncam; ncam += 6 * (int64_t)p->num_pose_priors; }
    for (int k = 0; k < p->num_pose_priors; ++k) {
      Extra e{}; e.nres = 6; e.ncol = 12; e.kind = 0; e.which = k;
      for (int c = 0; c < 6; ++c) { e.col[c] = ipp + 6 * (int64_t)k + c; e.col[6 + c] = 6 * (int64_t)p->pose_prior_block[k] + c; }
      ex.push_back(e);
    }
    if (p->has_spherical) {
      Extra e{}; e.nres = 2; e.ncol = 6; e.kind = 1; e.which = p->spherical_pose_block;
      for (int c = 0; c < 6; ++c) e.col[c] = 6 * (int64_t)p->spherical_pose_block + c;
      ex.push_back(e);
    }
    nparam = ncam + (int64_t)M * 3;
    colmask.assign(nparam, 0);
    for (int f = 0; f < F; ++f) for (int q = 0; q < L.P; ++q) {
      const uint8_t m = p->pose_fixed_mask ? p->pose_fixed_mask[f * L.P + q] : 0;
      for (int k = 0; k < 6; ++k) if (m & (1u << k)) colmask[(int64_t)f * L.CD + 6 * q + k] = 1;
    }
    if (!L.cal && p->intrinsics_constant) for (int c = 0; c < NI; ++c) if (p->intrinsics_constant[c])
      for (int k = 0; k < 9; ++k) colmask[(int64_t)F * L.CD + 9 * c + k] = 1;
    if (p->point_constant) for (int j = 0; j < M; ++j) if (p->point_constant[j])
      for (int k = 0; k < 3; ++k) colmask[ncam + 3 * (int64_t)j + k] = 1;
    dropped.assign(N, 0);
    for (int64_t i = 0; i < N; ++i) {
      bool all_const = true;
      for (int k = 0; k < L.K && all_const; ++k) if (!colmask[gcol(i, k)]) all_const = false;
      // a block is constant only if all 6 coordinates are fixed; a partially fixed pose keeps the residual
      dropped[i] = all_const;
    }
    pdropped.assign(NP, 0);
    for (int k = 0; k < NP; ++k) { bool all_const = true; for (int c = 0; c < PC && all_const; ++c) if (!colmask[pcol(k, c)]) all_const = false; pdropped[k] = all_const; }
    for (Extra& e : ex) { bool all_const = true; for (int c = 0; c < e.ncol && all_const; ++c) if (!colmask[e.col[c]]) all_const = false; e.dropped = all_const; }
  }
  // value of global camera-side column a that belongs to a pose or a priorPoses block
  inline double pose_like_value(int64_t a) const { return (ipp >= 0 && a >= ipp) ? p->pose_prior_values[a - ipp] : p->poses[a]; }
  // global column of local column k of observation i
  inline int64_t gcol(int64_t i, int k) const {
    const int f = p->obs_frame[i];
    if (k < L.off_pose) return (int64_t)F * L.CD + 9 * (int64_t)intr_of(p, f) + k;
    if (k < L.off_point) return (int64_t)f * L.CD + (k - L.off_pose);
    return ncam + 3 * (int64_t)p->obs_point[i] + (k - L.off_point);
  }
  // ResidualBlock::Evaluate (Ceres 1.9 residual_block.cc, restated): cost = rho0/2, correct the
  // Jacobian with the UNcorrected residual, then the residual; then drop fixed columns.
  // Returns false if any functor failed.  cost excludes dropped blocks; fixed receives theirs.
  bool run(bool want_jac, double* cost, double* fixed) {
    const int K = L.K;
    r.resize(2 * N); if (want_jac) J.resize((size_t)2 * K * N);
    block_fn fn = pick_block_fn(L);
    double c = 0.0, cf = 0.0; int64_t bad = 0;
    // sums over fixed chunks of observations, the chunk sums added up in order: the same value for any team size
    const int64_t nchunk = (N + kSumChunk - 1) / kSumChunk;
    std::vector<double> ch_c(nchunk, 0.0), ch_cf(nchunk, 0.0); std::vector<int64_t> ch_bad(nchunk, 0);
#pragma omp parallel for schedule(static)
    for (int64_t chunk = 0; chunk < nchunk; ++chunk) {
     double c = 0.0, cf = 0.0; int64_t bad = 0;
     for (int64_t i = chunk * kSumChunk; i < std::min(N, (chunk + 1) * kSumChunk); ++i) {
      double* ri = &r[2 * i]; double* Ji = want_jac ? &J[(size_t)2 * K * i] : nullptr;
      if (!fn(p, i, ri, Ji)) { ++bad; continue; }
      const double s = ri[0] * ri[0] + ri[1] * ri[1];
      double rho[3] = {s, 1.0, 0.0};
      if (p->huber_a > 0.0) huber(p->huber_a, s, rho);
      if (dropped[i]) cf += 0.5 * rho[0]; else c += 0.5 * rho[0];
      if (p->huber_a > 0.0) {
        // Ceres 1.9 corrector.cc (restated)
        const double sr1 = std::sqrt(rho[1]);
        double rscale = sr1, alpha_sq = 0.0;
        if (!(s == 0.0 || rho[2] <= 0.0)) {
          const double Dd = 1.0 + 2.0 * s * rho[2] / rho[1];
          const double alpha = 1.0 - std::sqrt(Dd);
          rscale = sr1 / (1.0 - alpha); alpha_sq = alpha / s;
        }
        if (Ji) {
          if (alpha_sq == 0.0) { for (int k = 0; k < 2 * K; ++k) Ji[k] *= sr1; }
          else for (int k = 0; k < K; ++k) {
            const double rtj = Ji[k] * ri[0] + Ji[K + k] * ri[1];
            Ji[k] = sr1 * (Ji[k] - alpha_sq * ri[0] * rtj);
            Ji[K + k] = sr1 * (Ji[K + k] - alpha_sq * ri[1] * rtj);
          }
        }
        ri[0] *= rscale; ri[1] *= rscale;
      }
      if (Ji) for (int k = 0; k < K; ++k) if (colmask[gcol(i, k)]) { Ji[k] = 0.0; Ji[K + k] = 0.0; }
     }
     ch_c[chunk] = c; ch_cf[chunk] = cf; ch_bad[chunk] = bad;
    }
    for (int64_t chunk = 0; chunk < nchunk; ++chunk) { c += ch_c[chunk]; cf += ch_cf[chunk]; bad += ch_bad[chunk]; }
    pr.resize((size_t)12 * NP); if (want_jac) pJ.resize((size_t)12 * kPC * NP);
    for (int k = 0; k < NP; ++k) {
      double* rk = &pr[(size_t)12 * k]; double* Jk = want_jac ? &pJ[(size_t)12 * kPC * k] : nullptr;
      if (!prior_eval(p, k, rk, Jk)) { ++bad; continue; }
      double s = 0.0; for (int i = 0; i < 12; ++i) s += rk[i] * rk[i];
      double rho[3] = {s, 1.0, 0.0};
      if (p->huber_a > 0.0) huber(p->huber_a, s, rho);
      if (pdropped[k]) cf += 0.5 * rho[0]; else c += 0.5 * rho[0];
      if (p->huber_a > 0.0) {   // the same corrector, 12 rows
        const double sr1 = std::sqrt(rho[1]);
        double rscale = sr1, alpha_sq = 0.0;
        if (!(s == 0.0 || rho[2] <= 0.0)) { const double alpha = 1.0 - std::sqrt(1.0 + 2.0 * s * rho[2] / rho[1]); rscale = sr1 / (1.0 - alpha); alpha_sq = alpha / s; }
        if (Jk) for (int cc = 0; cc < PC; ++cc) {
          double rtj = 0.0; for (int i = 0; i < 12; ++i) rtj += Jk[i * kPC + cc] * rk[i];
          for (int i = 0; i < 12; ++i) Jk[i * kPC + cc] = sr1 * (Jk[i * kPC + cc] - alpha_sq * rk[i] * rtj);
        }
        for (int i = 0; i < 12; ++i) rk[i] *= rscale;
      }
      if (Jk) for (int cc = 0; cc < PC; ++cc) if (colmask[pcol(k, cc)]) for (int i = 0; i < 12; ++i) Jk[i * kPC + cc] = 0.0;
    }
    for (Extra& e : ex) {
      typedef Dual<12> D;
      D x[12], res[6];
      for (int cc = 0; cc < e.ncol; ++cc) x[cc] = D(pose_like_value(e.col[cc]), cc);
      const bool ok = e.kind == 0 ? good_pose_prior<D>(p->pose_prior_rotation, p->pose_prior_position, x, x + 6, res) : spherical_prior<D>(x, res);
      if (!ok) { ++bad; continue; }
      double sq = 0.0;
      for (int i = 0; i < e.nres; ++i) { e.r[i] = res[i].a; sq += e.r[i] * e.r[i]; for (int cc = 0; cc < e.ncol; ++cc) e.J[i * 12 + cc] = colmask[e.col[cc]] ? 0.0 : res[i].v[cc]; }
      if (e.dropped) cf += 0.5 * sq; else c += 0.5 * sq;     // no loss function on these blocks
    }
    if (cost) *cost = c;
    if (fixed) *fixed = cf;
    return bad == 0;
  }
  // J^T r and J^T J contributions of the prior blocks, entry by entry
  template <class FG, class FH> void prior_normal(FG&& g, FH&& h) const {
    for (int k = 0; k < NP; ++k) {
      const double* Jk = &pJ[(size_t)12 * kPC * k]; const double* rk = &pr[(size_t)12 * k];
      for (int a = 0; a < PC; ++a) {
        double ga = 0.0; for (int i = 0; i < 12; ++i) ga += Jk[i * kPC + a] * rk[i];
        g(pcol(k, a), ga);
        for (int b = 0; b < PC; ++b) { double hab = 0.0; for (int i = 0; i < 12; ++i) hab += Jk[i * kPC + a] * Jk[i * kPC + b]; if (hab != 0.0) h(pcol(k, a), pcol(k, b), hab); }
      }
    }
    for (const Extra& e : ex) {
      for (int a = 0; a < e.ncol; ++a) {
        double ga = 0.0; for (int i = 0; i < e.nres; ++i) ga += e.J[i * 12 + a] * e.r[i];
        g(e.col[a], ga);
        for (int b = 0; b < e.ncol; ++b) { double hab = 0.0; for (int i = 0; i < e.nres; ++i) hab += e.J[i * 12 + a] * e.J[i * 12 + b]; if (hab != 0.0) h(e.col[a], e.col[b], hab); }
      }
    }
  }
};

// Envelope (skyline) storage of a symmetric matrix: row a keeps its columns lo[a] .. a.  The reduced camera system of a
// video is banded (a frame shares points with a few dozen neighbours) with, at most, a dense border at the end
// (shared intrinsics, the free interFrameRatio): inside the envelope it is treated as DENSE — the Cholesky factor of a
// matrix never leaves its envelope — so the arithmetic below is the dense row-by-row factorisation with the products
// that are exactly zero left out (same operands, same order: bit-identical to the dense form), at O(n b^2) instead
// of O(n^3).  That is what lets the checker run the 1k- and 4k-camera configurations (SparseCholesky in Ceres: CHOLMOD).
struct EnvMatrix {
  int64_t n = 0;
  std::vector<int64_t> lo, ptr, hi;   // hi[j] = last row whose envelope reaches column j
  std::vector<double> v;
  void shape(const std::vector<int64_t>& lo_) {
    n = (int64_t)lo_.size(); lo = lo_; ptr.assign(n + 1, 0); hi.assign(n, 0);
    for (int64_t a = 0; a < n; ++a) ptr[a + 1] = ptr[a] + (a - lo[a] + 1);
    for (int64_t j = 0; j < n; ++j) hi[j] = j;
    for (int64_t a = 0; a < n; ++a) hi[lo[a]] = std::max(hi[lo[a]], a);
    for (int64_t j = 1; j < n; ++j) hi[j] = std::max(hi[j], hi[j - 1]);
    v.assign((size_t)ptr[n], 0.0);
  }
  inline bool has(int64_t a, int64_t b) const { return b <= a && b >= lo[a]; }
  inline double& at(int64_t a, int64_t b) { return v[(size_t)(ptr[a] + (b - lo[a]))]; }           // b in [lo[a], a]
  inline double get(int64_t a, int64_t b) const { if (b > a) std::swap(a, b); return b >= lo[a] ? v[(size_t)(ptr[a] + (b - lo[a]))] : 0.0; }
  inline const double* row(int64_t a) const { return &v[(size_t)ptr[a]] - lo[a]; }                // row(a)[b] for b in [lo[a], a]
  inline double* row(int64_t a) { return &v[(size_t)ptr[a]] - lo[a]; }
};

// Cholesky A = L L^T in place (lower, inside the envelope), returns false if not positive definite.
// L_ij = (A_ij - sum_{k<j} L_ik L_jk) / L_jj, k ascending — the dense inner-product form.
bool cholesky(EnvMatrix& A) {
  const int64_t n = A.n;
  for (int64_t j = 0; j < n; ++j) {
    double* aj = A.row(j);
    double d = aj[j];
    for (int64_t k = A.lo[j]; k < j; ++k) d -= aj[k] * aj[k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    const double ljj = std::sqrt(d);
    aj[j] = ljj;
    const double inv = 1.0 / ljj;
    const int64_t last = A.hi[j];
#pragma omp parallel for schedule(static) if (last - j > 256)
    for (int64_t i = j + 1; i <= last; ++i) {
      if (A.lo[i] > j) continue;
      double* ai = A.row(i);
      double s = ai[j];
      for (int64_t k = std::max(A.lo[i], A.lo[j]); k < j; ++k) s -= ai[k] * aj[k];
      ai[j] = s * inv;
    }
  }
  return true;
}
void chol_solve(const EnvMatrix& A, std::vector<double>& b) {
  const int64_t n = A.n;
  for (int64_t i = 0; i < n; ++i) { const double* ai = A.row(i); double s = b[i]; for (int64_t k = A.lo[i]; k < i; ++k) s -= ai[k] * b[k]; b[i] = s / ai[i]; }
  for (int64_t i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int64_t k = i + 1; k <= A.hi[i]; ++k) if (A.lo[k] <= i) s -= A.row(k)[i] * b[k];
    b[i] = s / A.row(i)[i];
  }
}
// dense forms (small systems: the covariance of the free coordinates)
bool cholesky(std::vector<double>& A, int64_t n) {
  for (int64_t j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int64_t k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    const double ljj = std::sqrt(d);
    A[j * n + j] = ljj;
    const double inv = 1.0 / ljj;
#pragma omp parallel for schedule(static) if (n - j > 256)
    for (int64_t i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      const double* ai = &A[i * n]; const double* aj = &A[j * n];
      for (int64_t k = 0; k < j; ++k) s -= ai[k] * aj[k];
      A[i * n + j] = s * inv;
    }
  }
  return true;
}
void chol_solve(const std::vector<double>& A, int64_t n, std::vector<double>& b) {
  for (int64_t i = 0; i < n; ++i) { double s = b[i]; for (int64_t k = 0; k < i; ++k) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
  for (int64_t i = n - 1; i >= 0; --i) { double s = b[i]; for (int64_t k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
}
bool inv3_sym(const double V[9], double O[9]) {
  double m[9]; std::memcpy(m, V, sizeof m);
  const double d = det33(m);
  if (!(std::fabs(d) > 0.0) || !std::isfinite(d)) return false;
  const double id = 1.0 / d;
  O[0] = (m[4] * m[8] - m[5] * m[7]) * id; O[1] = (m[2] * m[7] - m[1] * m[8]) * id; O[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  O[3] = (m[5] * m[6] - m[3] * m[8]) * id; O[4] = (m[0] * m[8] - m[2] * m[6]) * id; O[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  O[6] = (m[3] * m[7] - m[4] * m[6]) * id; O[7] = (m[1] * m[6] - m[0] * m[7]) * id; O[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return true;
}

// Point-major adjacency (which observations see point j)
struct PointCsr { std::vector<int64_t> ptr, idx; };
PointCsr point_csr(const orc_problem* p) {
  PointCsr c; const int M = p->num_points; const int64_t N = p->num_observations;
  c.ptr.assign(M + 1, 0); c.idx.resize(N);
  for (int64_t i = 0; i < N; ++i) c.ptr[p->obs_point[i] + 1]++;
  for (int j = 0; j < M; ++j) c.ptr[j + 1] += c.ptr[j];
  std::vector<int64_t> fill(c.ptr.begin(), c.ptr.end() - 1);
  for (int64_t i = 0; i < N; ++i) c.idx[fill[p->obs_point[i]]++] = i;
  return c;
}

// SchurComplementSolver restated (Ceres 1.9 schur_complement_solver.cc / schur_eliminator_impl.h):
// solve (J^T J + D^2) y = J^T r exactly by eliminating the point blocks, Cholesky on the
// reduced camera system, back-substitution.  J is the (already column-scaled) corrected Jacobian.
// reduced camera system S = U + D_c^2 - sum_j W_j (V_j + D_p^2)^-1 W_j^T (lower triangle, envelope storage) and its
// right-hand side, with the inverted point blocks and point gradients the back-substitution needs.
// Every entry of S and rhs is accumulated by ONE thread (the owner of its row: contiguous ranges of frames) in the order
// a serial sweep would use — observations ascending, then priors, damping, points ascending — so the result does not
// depend on the team size.
bool reduced_system(const Eval& E, const PointCsr& pc, const std::vector<double>& J, const std::vector<double>& r,
                    const std::vector<double>& D2, EnvMatrix& S, std::vector<double>& rhs,
                    std::vector<double>& Vinv, std::vector<double>& bp) {
  const Layout& L = E.L; const int K = L.K; const int KC = K - 3;   // camera-side columns of one block
  const int64_t nc = E.ncam, N = E.N; const int M = E.M; const int F = E.F;
  // envelope: a row reaches back to the first column any of its points (or prior blocks) couples it with
  {
    std::vector<int64_t> lo(nc);
    for (int64_t a = 0; a < nc; ++a) lo[a] = a;
    for (int j = 0; j < M; ++j) {
      int64_t cmin = nc;
      for (int64_t t = pc.ptr[j]; t < pc.ptr[j + 1]; ++t) for (int a = 0; a < KC; ++a) cmin = std::min(cmin, E.gcol(pc.idx[t], a));
      for (int64_t t = pc.ptr[j]; t < pc.ptr[j + 1]; ++t) for (int a = 0; a < KC; ++a) { int64_t& l = lo[E.gcol(pc.idx[t], a)]; l = std::min(l, cmin); }
    }
    for (int k = 0; k < E.NP; ++k) {
      int64_t cmin = nc;
      for (int c = 0; c < E.PC; ++c) cmin = std::min(cmin, E.pcol(k, c));
      for (int c = 0; c < E.PC; ++c) { int64_t& l = lo[E.pcol(k, c)]; l = std::min(l, cmin); }
    }
    for (const Eval::Extra& e : E.ex) {
      int64_t cmin = nc;
      for (int c = 0; c < e.ncol; ++c) cmin = std::min(cmin, e.col[c]);
      for (int c = 0; c < e.ncol; ++c) { int64_t& l = lo[e.col[c]]; l = std::min(l, cmin); }
    }
    S.shape(lo);
  }
  rhs.assign(nc, 0.0);
  Vinv.assign((size_t)9 * M, 0.0); bp.assign((size_t)3 * M, 0.0);
  // point blocks first (independent of each other)
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (int j = 0; j < M; ++j) {
    double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int64_t t = pc.ptr[j]; t < pc.ptr[j + 1]; ++t) {
      const int64_t i = pc.idx[t]; const double* Jp0 = &J[(size_t)2 * K * i + L.off_point]; const double* Jp1 = Jp0 + K;
      for (int a = 0; a < 3; ++a) {
        g[a] += Jp0[a] * r[2 * i] + Jp1[a] * r[2 * i + 1];
        for (int b = 0; b < 3; ++b) V[3 * a + b] += Jp0[a] * Jp0[b] + Jp1[a] * Jp1[b];
      }
    }
    for (int a = 0; a < 3; ++a) V[4 * a] += D2[nc + 3 * (int64_t)j + a];
    if (!inv3_sym(V, &Vinv[(size_t)9 * j])) { ++bad; continue; }
    for (int a = 0; a < 3; ++a) bp[3 * (size_t)j + a] = g[a];
  }
  // frame range of every point (the rows its elimination touches), for the owners' quick skip
  std::vector<int32_t> pf0(M, F), pf1(M, -1);
  for (int j = 0; j < M; ++j) for (int64_t t = pc.ptr[j]; t < pc.ptr[j + 1]; ++t) { const int f = E.p->obs_frame[pc.idx[t]]; pf0[j] = std::min(pf0[j], f); pf1[j] = std::max(pf1[j], f); }
  const int64_t npose = (int64_t)F * L.CD;
#pragma omp parallel
  {
#ifdef _OPENMP
    const int T = omp_get_num_threads(), me = omp_get_thread_num();
#else
    const int T = 1, me = 0;
#endif
    const int f_lo = (int)((int64_t)F * me / T), f_hi = (int)((int64_t)F * (me + 1) / T);   // my frames; the rows behind the poses belong to the last thread
    const bool tail = me == T - 1;
    auto mine = [&](int64_t ga) { return ga < npose ? (ga >= (int64_t)f_lo * L.CD && ga < (int64_t)f_hi * L.CD) : tail; };
    // camera-side Hessian and gradient
    for (int64_t i = 0; i < N; ++i) {
      const int f = E.p->obs_frame[i];
      if (!((f >= f_lo && f < f_hi) || (tail && KC > L.CD))) continue;
      const double* Ji = &J[(size_t)2 * K * i];
      for (int a = 0; a < KC; ++a) {
        const int64_t ga = E.gcol(i, a);
        if (!mine(ga)) continue;
        rhs[ga] += Ji[a] * r[2 * i] + Ji[K + a] * r[2 * i + 1];
        for (int b = 0; b < KC; ++b) { const int64_t gb = E.gcol(i, b); if (gb <= ga) S.at(ga, gb) += Ji[a] * Ji[b] + Ji[K + a] * Ji[K + b]; }
      }
    }
    E.prior_normal([&](int64_t a, double v) { if (mine(a)) rhs[a] += v; }, [&](int64_t a, int64_t b, double v) { if (b <= a && mine(a)) S.at(a, b) += v; });
    for (int64_t a = 0; a < nc; ++a) if (mine(a)) S.at(a, a) += D2[a];
    std::vector<double> W, Y;
    for (int j = 0; j < M; ++j) {
      if (!((pf0[j] < f_hi && pf1[j] >= f_lo) || (tail && KC > L.CD))) continue;
      const double* Vi = &Vinv[(size_t)9 * j]; const double* g = &bp[3 * (size_t)j];
      // W_o = Jc_o^T Jp_o (KC x 3);  Y_o = W_o Vinv
      const int64_t n = pc.ptr[j + 1] - pc.ptr[j];
      W.resize((size_t)n * KC * 3); Y.resize((size_t)n * KC * 3);
      for (int64_t t = 0; t < n; ++t) {
        const int64_t i = pc.idx[pc.ptr[j] + t]; const double* Ji = &J[(size_t)2 * K * i];
        for (int a = 0; a < KC; ++a) for (int b = 0; b < 3; ++b)
          W[(t * KC + a) * 3 + b] = Ji[a] * Ji[L.off_point + b] + Ji[K + a] * Ji[K + L.off_point + b];
        for (int a = 0; a < KC; ++a) for (int b = 0; b < 3; ++b)
          Y[(t * KC + a) * 3 + b] = W[(t * KC + a) * 3 + 0] * Vi[b] + W[(t * KC + a) * 3 + 1] * Vi[3 + b] + W[(t * KC + a) * 3 + 2] * Vi[6 + b];
      }
      for (int64_t t1 = 0; t1 < n; ++t1) {
        const int64_t i1 = pc.idx[pc.ptr[j] + t1];
        for (int a = 0; a < KC; ++a) {
          const int64_t ga = E.gcol(i1, a);
          if (!mine(ga)) continue;
          const double* Ya = &Y[(t1 * KC + a) * 3];
          rhs[ga] -= Ya[0] * g[0] + Ya[1] * g[1] + Ya[2] * g[2];
          double* Sa = S.row(ga);
          for (int64_t t2 = 0; t2 < n; ++t2) {
            const int64_t i2 = pc.idx[pc.ptr[j] + t2];
            for (int b = 0; b < KC; ++b) {
              const int64_t gb = E.gcol(i2, b);
              if (gb > ga) continue;
              const double* Wb = &W[(t2 * KC + b) * 3];
              Sa[gb] -= Ya[0] * Wb[0] + Ya[1] * Wb[1] + Ya[2] * Wb[2];
            }
          }
        }
      }
    }
  }
  return bad == 0;
}

bool schur_solve(const Eval& E, const PointCsr& pc, const std::vector<double>& J, const std::vector<double>& r,
                 const std::vector<double>& D2, std::vector<double>& y) {
  const Layout& L = E.L; const int K = L.K; const int KC = K - 3;
  const int64_t nc = E.ncam; const int M = E.M;
  EnvMatrix S; std::vector<double> rhs, Vinv, bp;
  y.assign(E.nparam, 0.0);
  if (!reduced_system(E, pc, J, r, D2, S, rhs, Vinv, bp)) return false;
  if (!cholesky(S)) return false;
  chol_solve(S, rhs);
  for (int64_t a = 0; a < nc; ++a) y[a] = rhs[a];
  // back-substitution  y_p = Vinv (b_p - sum_o W_o^T y_c)
#pragma omp parallel for schedule(static)
  for (int j = 0; j < M; ++j) {
    double t3[3] = {bp[3 * (size_t)j], bp[3 * (size_t)j + 1], bp[3 * (size_t)j + 2]};
    for (int64_t t = pc.ptr[j]; t < pc.ptr[j + 1]; ++t) {
      const int64_t i = pc.idx[t]; const double* Ji = &J[(size_t)2 * K * i];
      // (Jp^T Jc) y_c = Jp^T (Jc y_c)
      double m0 = 0.0, m1 = 0.0;
      for (int a = 0; a < KC; ++a) { const double ya = y[E.gcol(i, a)]; m0 += Ji[a] * ya; m1 += Ji[K + a] * ya; }
      for (int b = 0; b < 3; ++b) t3[b] -= Ji[L.off_point + b] * m0 + Ji[K + L.off_point + b] * m1;
    }
    const double* Vi = &Vinv[(size_t)9 * j];
    for (int a = 0; a < 3; ++a) y[nc + 3 * (int64_t)j + a] = Vi[3 * a] * t3[0] + Vi[3 * a + 1] * t3[1] + Vi[3 * a + 2] * t3[2];
  }
  for (double v : y) if (!std::isfinite(v)) return false;
  return true;
}

}  // namespace

extern "C" {

void orc_default_options(orc_options* o) {
  // Ceres 1.9 Solver::Options defaults; max_num_iterations as CeresHandler.h:405
  o->max_num_iterations = 50; o->jacobi_scaling = 1; o->max_num_consecutive_invalid_steps = 5; o->num_threads = 0;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
}

int32_t orc_jacobian_cols(const orc_problem* p) { return layout_of(p).K; }

int64_t orc_evaluate_blocks(const orc_problem* p, double* residuals, double* jacobians, uint8_t* ok, int32_t num_threads) {
  const Layout L = layout_of(p); block_fn fn = pick_block_fn(L); int64_t bad = 0; const int nt = threads_of(num_threads); (void)nt;
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(nt)
  for (int64_t i = 0; i < p->num_observations; ++i) {
    double scratch[2 * 24];
    const bool good = fn(p, i, residuals + 2 * i, jacobians ? jacobians + (size_t)2 * L.K * i : scratch);
    if (ok) ok[i] = good; if (!good) ++bad;
  }
  return bad;
}

int64_t orc_evaluate_residuals(const orc_problem* p, double* residuals, uint8_t* ok, int32_t num_threads) {
  const Layout L = layout_of(p); block_fn fn = pick_block_fn(L); int64_t bad = 0; const int nt = threads_of(num_threads); (void)nt;
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(nt)
  for (int64_t i = 0; i < p->num_observations; ++i) {
    const bool good = fn(p, i, residuals + 2 * i, nullptr);
    if (ok) ok[i] = good; if (!good) ++bad;
  }
  return bad;
}

int64_t orc_evaluate_blocks_ceres_style(const orc_problem* p, double* residuals, double* jacobians, int32_t num_threads) {
  const Layout L = layout_of(p); const int64_t N = p->num_observations; const int nt = threads_of(num_threads); (void)nt;
  // problem build (CeresHandler.h:249-280: new AutoDiffCostFunction(new Functor) per observation) — untimed by callers
  static thread_local std::vector<std::unique_ptr<CostObject>> blocks; static thread_local const orc_problem* built_for = nullptr;
  if (built_for != p || (int64_t)blocks.size() != N) {
    blocks.clear(); blocks.reserve(N);
    for (int64_t i = 0; i < N; ++i) {
      if (L.cal) { if (L.P == 2) blocks.emplace_back(new BlockCost<true, 2>(p, i)); else blocks.emplace_back(new BlockCost<true, 1>(p, i)); }
      else { if (L.P == 2) blocks.emplace_back(new BlockCost<false, 2>(p, i)); else blocks.emplace_back(new BlockCost<false, 1>(p, i)); }
    }
    built_for = p;
    if (!residuals) return 0;   // build-only call
  }
  int64_t bad = 0;
  std::vector<std::unique_ptr<CostObject>>& B = blocks;
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(nt)
  for (int64_t i = 0; i < N; ++i) {
    const int f = p->obs_frame[i];
    const double* params[4]; double* jac[4]; int b = 0;
    double* Ji = jacobians + (size_t)2 * L.K * i;   // blocks stored back to back, each row-major 2 x Ni
    int off = 0;
    if (!L.cal) { params[b] = p->intrinsics + 9 * intr_of(p, f); jac[b] = Ji + off; off += 18; ++b; }
    for (int q = 0; q < L.P; ++q) { params[b] = p->poses + ((size_t)f * L.P + q) * 6; jac[b] = Ji + off; off += 12; ++b; }
    params[b] = p->points + (size_t)p->obs_point[i] * 3; jac[b] = Ji + off;
    if (!B[i]->Evaluate(params, residuals + 2 * i, jac)) ++bad;
  }
  return bad;
}

int32_t orc_evaluate(const orc_problem* p, double* cost, double* gradient) {
  Eval E(p); double c = 0, cf = 0;
  const bool ok = E.run(gradient != nullptr, &c, &cf);
  if (cost) *cost = c + cf;
  if (gradient) {
    // laid out [F*P*6 | M*3 | NI*9]
    const int64_t npose = (int64_t)E.F * E.L.CD;
    std::vector<double> g(E.nparam, 0.0);
    for (int64_t i = 0; i < E.N; ++i) for (int k = 0; k < E.L.K; ++k)
      g[E.gcol(i, k)] += E.J[(size_t)2 * E.L.K * i + k] * E.r[2 * i] + E.J[(size_t)2 * E.L.K * i + E.L.K + k] * E.r[2 * i + 1];
    E.prior_normal([&](int64_t a, double v) { g[a] += v; }, [](int64_t, int64_t, double) {});
    for (int64_t a = 0; a < npose; ++a) gradient[a] = g[a];
    for (int64_t a = 0; a < 3 * (int64_t)E.M; ++a) gradient[npose + a] = g[E.ncam + a];
    if (!E.L.cal) for (int64_t a = 0; a < 9 * (int64_t)E.NI; ++a) gradient[npose + 3 * (int64_t)E.M + a] = g[npose + a];
    else for (int64_t a = 0; a < 9 * (int64_t)E.NI; ++a) gradient[npose + 3 * (int64_t)E.M + a] = 0.0;
  }
  return ok ? 0 : 1;
}

int32_t orc_normal_equations(const orc_problem* p, double* U, double* gc, double* V, double* gp) {
  Eval E(p); if (!E.L.cal) return 2;
  if (!E.run(true, nullptr, nullptr)) return 1;
  const int CD = E.L.CD, K = E.L.K;
  std::fill(U, U + (size_t)E.F * CD * CD, 0.0); std::fill(gc, gc + (size_t)E.F * CD, 0.0);
  std::fill(V, V + (size_t)E.M * 9, 0.0); std::fill(gp, gp + (size_t)E.M * 3, 0.0);
  for (int64_t i = 0; i < E.N; ++i) {
    const double* Ji = &E.J[(size_t)2 * K * i]; const int f = p->obs_frame[i], j = p->obs_point[i];
    for (int a = 0; a < CD; ++a) {
      gc[(size_t)f * CD + a] += Ji[a] * E.r[2 * i] + Ji[K + a] * E.r[2 * i + 1];
      for (int b = 0; b < CD; ++b) U[((size_t)f * CD + a) * CD + b] += Ji[a] * Ji[b] + Ji[K + a] * Ji[K + b];
    }
    for (int a = 0; a < 3; ++a) {
      gp[(size_t)j * 3 + a] += Ji[CD + a] * E.r[2 * i] + Ji[K + CD + a] * E.r[2 * i + 1];
      for (int b = 0; b < 3; ++b) V[(size_t)j * 9 + 3 * a + b] += Ji[CD + a] * Ji[CD + b] + Ji[K + CD + a] * Ji[K + CD + b];
    }
  }
  // motion priors: their share of the block diagonal (the frame-to-frame cross blocks are not part of this output)
  const int64_t npose_cols = (int64_t)E.F * CD;   // (the priorPoses blocks of GoodPosePrior are not part of this output)
  E.prior_normal([&](int64_t a, double v) { if (a < npose_cols) gc[a] += v; },
                 [&](int64_t a, int64_t b, double v) { if (a < npose_cols && b < npose_cols && a / CD == b / CD) U[(size_t)(a / CD) * CD * CD + (a % CD) * CD + (b % CD)] += v; });
  return 0;
}

// TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy (Ceres 1.9, restated)
int32_t orc_solve(orc_problem* p, const orc_options* opt, orc_summary* sum, orc_iteration* trace, int32_t trace_cap) {
#ifdef _OPENMP
  if (opt->num_threads > 0) omp_set_num_threads(opt->num_threads);
#endif
  Eval E(p); const Layout& L = E.L; const int K = L.K; const int64_t N = E.N, np = E.nparam;
  const PointCsr pc = point_csr(p);
  std::memset(sum, 0, sizeof *sum);
  sum->termination_type = ORC_NO_CONVERGENCE;
  sum->num_residual_blocks = (int32_t)(N + E.NP + (int64_t)E.ex.size());
  int64_t nred = 0; for (int64_t i = 0; i < N; ++i) nred += !E.dropped[i];
  for (int k = 0; k < E.NP; ++k) nred += !E.pdropped[k];
  for (const Eval::Extra& e : E.ex) nred += !e.dropped;
  sum->num_residual_blocks_reduced = (int32_t)nred;
  // which parameter BLOCKS are in the reduced program (non-constant): used for |x| and |step|
  std::vector<uint8_t> in_program(np, 0);
  {
    const int64_t npose_blocks = (int64_t)E.F * L.P;
    for (int64_t b = 0; b < npose_blocks; ++b) {
      bool all = true; for (int k = 0; k < 6; ++k) all = all && E.colmask[6 * b + k];
      if (!all) for (int k = 0; k < 6; ++k) in_program[6 * b + k] = 1;
    }
    if (!L.cal) for (int c = 0; c < E.NI; ++c) if (!E.colmask[(int64_t)E.F * L.CD + 9 * c]) for (int k = 0; k < 9; ++k) in_program[(int64_t)E.F * L.CD + 9 * c + k] = 1;
    for (int j = 0; j < E.M; ++j) if (!E.colmask[E.ncam + 3 * (int64_t)j] && pc.ptr[j + 1] > pc.ptr[j]) for (int k = 0; k < 3; ++k) in_program[E.ncam + 3 * (int64_t)j + k] = 1;
    // pose blocks / intrinsics that no residual touches are not part of the problem
    std::vector<uint8_t> touched(np, 0);
    for (int64_t i = 0; i < N; ++i) for (int k = 0; k < K; ++k) touched[E.gcol(i, k)] = 1;
    if (E.iratio >= 0) for (int k = 0; k < 1; ++k) in_program[E.iratio] = 1;
    for (int k = 0; k < E.NP; ++k) for (int c = 0; c < E.PC; ++c) touched[E.pcol(k, c)] = 1;
    if (E.ipp >= 0) for (int64_t a = E.ipp; a < E.ncam; ++a) in_program[a] = 1;     // the priorPoses blocks are free parameter blocks
    for (const Eval::Extra& e : E.ex) for (int c = 0; c < e.ncol; ++c) touched[e.col[c]] = 1;
    for (int64_t a = 0; a < np; ++a) if (!touched[a]) in_program[a] = 0;
  }
  int64_t nfree = 0; for (int64_t a = 0; a < np; ++a) nfree += (in_program[a] && !E.colmask[a]);
  sum->num_parameters_reduced = (int32_t)nfree;

  auto param_ptr = [&](int64_t a) -> double* {
    const int64_t npose = (int64_t)E.F * L.CD;
    if (a < npose) return p->poses + a;
    if (a == E.iratio) return &p->inter_frame_ratio;
    if (E.ipp >= 0 && a >= E.ipp && a < E.ncam) return p->pose_prior_values + (a - E.ipp);
    if (a < E.ncam) return p->intrinsics + (a - npose);
    return p->points + (a - E.ncam);
  };
  // SetParameterLowerBound(&interFrameRatio, 0, 0.0 | _EPS) (CeresHandler.h:161,172): the candidate is projected onto
  // the bound (ParameterBlock::Plus), the gradient norm is that of the projected gradient.  Ceres' additional
  // projected line search for bounded problems (>= 1.10) is NOT restated.
  const double ratio_lb = p->prior_kind == 2 ? kEps : 0.0;
  auto x_norm_of = [&]() { double s = 0; for (int64_t a = 0; a < np; ++a) if (in_program[a]) { const double v = *param_ptr(a); s += v * v; } return std::sqrt(s); };
  auto gradient_of = [&](std::vector<double>& g) {
    g.assign(np, 0.0);
    for (int64_t i = 0; i < N; ++i) for (int k = 0; k < K; ++k)
      g[E.gcol(i, k)] += E.J[(size_t)2 * K * i + k] * E.r[2 * i] + E.J[(size_t)2 * K * i + K + k] * E.r[2 * i + 1];
    E.prior_normal([&](int64_t a, double v) { g[a] += v; }, [](int64_t, int64_t, double) {});
  };
  auto scale_cols = [&](const std::vector<double>& sc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) for (int k = 0; k < K; ++k) { const double s = sc[E.gcol(i, k)]; E.J[(size_t)2 * K * i + k] *= s; E.J[(size_t)2 * K * i + K + k] *= s; }
    for (int k = 0; k < E.NP; ++k) for (int c = 0; c < E.PC; ++c) { const double s = sc[E.pcol(k, c)]; for (int i = 0; i < 12; ++i) E.pJ[(size_t)12 * kPC * k + i * kPC + c] *= s; }
    for (Eval::Extra& e : E.ex) for (int c = 0; c < e.ncol; ++c) { const double s = sc[e.col[c]]; for (int i = 0; i < e.nres; ++i) e.J[i * 12 + c] *= s; }
  };
  auto col_sq_norms = [&](std::vector<double>& d) {
    d.assign(np, 0.0);
    for (int64_t i = 0; i < N; ++i) for (int k = 0; k < K; ++k) { const double a = E.J[(size_t)2 * K * i + k], b = E.J[(size_t)2 * K * i + K + k]; d[E.gcol(i, k)] += a * a + b * b; }
    for (int k = 0; k < E.NP; ++k) for (int c = 0; c < E.PC; ++c) for (int i = 0; i < 12; ++i) { const double a = E.pJ[(size_t)12 * kPC * k + i * kPC + c]; d[E.pcol(k, c)] += a * a; }
    for (const Eval::Extra& e : E.ex) for (int c = 0; c < e.ncol; ++c) for (int i = 0; i < e.nres; ++i) { const double a = e.J[i * 12 + c]; d[e.col[c]] += a * a; }
  };
  int ntrace = 0;
  auto push = [&](const orc_iteration& it) { if (trace && ntrace < trace_cap) trace[ntrace] = it; ++ntrace; sum->num_iterations = ntrace; };

  double cost = 0, fixed = 0;
  if (!E.run(true, &cost, &fixed)) { sum->termination_type = ORC_FAILURE; return ORC_FAILURE; }
  sum->fixed_cost = fixed; sum->initial_cost = cost + fixed; sum->final_cost = cost + fixed;
  std::vector<double> g; gradient_of(g);
  auto gmax_of = [&](const std::vector<double>& gv) {
    double m = 0;
    for (int64_t a = 0; a < np; ++a) {
      double v = gv[a];
      if (a == E.iratio) { const double x = p->inter_frame_ratio; v = x - std::max(ratio_lb, x - v); }   // projected gradient
      m = std::max(m, std::fabs(v));
    }
    return m;
  };
  double gmax = gmax_of(g);
  double radius = opt->initial_trust_region_radius, decrease_factor = 2.0; bool reuse_diagonal = false;
  orc_iteration it; std::memset(&it, 0, sizeof it);
  it.cost = cost + fixed; it.gradient_max_norm = gmax; it.trust_region_radius = radius;
  double x_norm = x_norm_of();
  if (gmax <= opt->gradient_tolerance) { push(it); sum->termination_type = ORC_CONVERGENCE; return ORC_CONVERGENCE; }
  std::vector<double> scale(np, 1.0);
  if (opt->jacobi_scaling) { std::vector<double> d; col_sq_norms(d); for (int64_t a = 0; a < np; ++a) scale[a] = 1.0 / (1.0 + std::sqrt(d[a])); scale_cols(scale); }
  push(it);

  std::vector<double> diagonal, D2(np), y, x_save(np), r_cur, J_cur, pr_cur;
  int invalid_streak = 0; int iteration = 0;
  while (true) {
    if (iteration >= opt->max_num_iterations) { sum->termination_type = ORC_NO_CONVERGENCE; break; }
    if (!reuse_diagonal) { col_sq_norms(diagonal); for (double& v : diagonal) v = std::min(std::max(v, opt->min_lm_diagonal), opt->max_lm_diagonal); }
    for (int64_t a = 0; a < np; ++a) D2[a] = diagonal[a] / radius;
    const bool solved = schur_solve(E, pc, E.J, E.r, D2, y);
    reuse_diagonal = true;
    ++iteration;
    std::memset(&it, 0, sizeof it); it.iteration = iteration;
    double model_cost_change = 0.0; bool valid = false;
    if (solved) {
      // step = -y ; model_cost_change = -(J step) . (r + J step / 2)
      double acc = 0.0;
      const int64_t nchunk = (N + kSumChunk - 1) / kSumChunk;
      std::vector<double> ch_acc(nchunk, 0.0);
#pragma omp parallel for schedule(static)
      for (int64_t chunk = 0; chunk < nchunk; ++chunk) {
        double a = 0.0;
        for (int64_t i = chunk * kSumChunk; i < std::min(N, (chunk + 1) * kSumChunk); ++i) {
          double m0 = 0, m1 = 0;
          for (int k = 0; k < K; ++k) { const double s = -y[E.gcol(i, k)]; m0 += E.J[(size_t)2 * K * i + k] * s; m1 += E.J[(size_t)2 * K * i + K + k] * s; }
          a += m0 * (E.r[2 * i] + 0.5 * m0) + m1 * (E.r[2 * i + 1] + 0.5 * m1);
        }
        ch_acc[chunk] = a;
      }
      for (int64_t chunk = 0; chunk < nchunk; ++chunk) acc += ch_acc[chunk];
      for (int k = 0; k < E.NP; ++k) for (int i = 0; i < 12; ++i) {
        double m = 0.0;
        for (int c = 0; c < E.PC; ++c) m += E.pJ[(size_t)12 * kPC * k + i * kPC + c] * -y[E.pcol(k, c)];
        acc += m * (E.pr[(size_t)12 * k + i] + 0.5 * m);
      }
      for (const Eval::Extra& e : E.ex) for (int i = 0; i < e.nres; ++i) {
        double m = 0.0;
        for (int c = 0; c < e.ncol; ++c) m += e.J[i * 12 + c] * -y[e.col[c]];
        acc += m * (e.r[i] + 0.5 * m);
      }
      model_cost_change = -acc;
      valid = model_cost_change >= 0.0;   // Ceres: invalid iff model_cost_change < 0
    }
    it.model_cost_change = model_cost_change;
    bool successful = false;
    if (!valid) {
      if (++invalid_streak >= opt->max_num_consecutive_invalid_steps) { sum->termination_type = ORC_FAILURE; it.cost = cost + fixed; it.trust_region_radius = radius; push(it); break; }
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;   // StepIsInvalid == StepRejected(0)
      it.cost = cost + fixed; it.gradient_max_norm = gmax;
    } else {
      invalid_streak = 0; it.step_is_valid = 1;
      // x_plus_delta = x + scale .* step (masked coordinates have step 0)
      double step_sq = 0.0;
      for (int64_t a = 0; a < np; ++a) { x_save[a] = *param_ptr(a); const double d = -y[a] * scale[a]; if (in_program[a] && !E.colmask[a]) { *param_ptr(a) = x_save[a] + d; if (a == E.iratio && *param_ptr(a) < ratio_lb) *param_ptr(a) = ratio_lb; const double e = x_save[a] - *param_ptr(a); step_sq += e * e; } }
      // keep the current linearisation; evaluate residuals only at the candidate
      r_cur.swap(E.r); J_cur.swap(E.J); pr_cur.swap(E.pr);
      const std::vector<Eval::Extra> ex_cur = E.ex;
      double new_cost = 0, new_fixed = 0;
      const bool ev_ok = E.run(false, &new_cost, &new_fixed);
      E.r.swap(r_cur); E.J.swap(J_cur); E.pr.swap(pr_cur);
      E.ex = ex_cur;
      if (!ev_ok) new_cost = std::numeric_limits<double>::max();
      it.step_norm = std::sqrt(step_sq);
      const double step_tol = opt->parameter_tolerance * (x_norm + opt->parameter_tolerance);
      auto restore = [&]() { for (int64_t a = 0; a < np; ++a) *param_ptr(a) = x_save[a]; };
      if (it.step_norm <= step_tol) { restore(); sum->termination_type = ORC_CONVERGENCE; it.cost = cost + fixed; it.trust_region_radius = radius; push(it); break; }
      it.cost_change = cost - new_cost;
      if (std::fabs(it.cost_change) < opt->function_tolerance * cost) { restore(); sum->termination_type = ORC_CONVERGENCE; it.cost = cost + fixed; it.trust_region_radius = radius; push(it); break; }
      it.relative_decrease = it.cost_change / model_cost_change;
      successful = it.relative_decrease > opt->min_relative_decrease;
      if (successful) {
        it.step_is_successful = 1; ++sum->num_successful_steps;
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
        radius = std::min(opt->max_trust_region_radius, radius); decrease_factor = 2.0; reuse_diagonal = false;
        x_norm = x_norm_of();
        if (!E.run(true, &cost, &fixed)) { sum->termination_type = ORC_FAILURE; push(it); break; }
        gradient_of(g); gmax = gmax_of(g);
        it.gradient_max_norm = gmax;
        sum->final_cost = std::min(sum->final_cost, cost + fixed);
        if (gmax <= opt->gradient_tolerance) { sum->termination_type = ORC_CONVERGENCE; it.cost = cost + fixed; it.trust_region_radius = radius; push(it); break; }
        if (opt->jacobi_scaling) scale_cols(scale);
      } else {
        restore(); ++sum->num_unsuccessful_steps; it.gradient_max_norm = gmax;
        radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      }
    }
    if (!valid) ++sum->num_unsuccessful_steps;
    it.cost = cost + fixed; it.trust_region_radius = radius;
    if (radius < opt->min_trust_region_radius) { sum->termination_type = ORC_CONVERGENCE; push(it); break; }
    push(it);
  }
  return sum->termination_type;
}

// ceres::Covariance::Compute + GetCovarianceBlock for the pose block(s) of one frame (reference call site
// VideoSfMHandler.cc:602-621; Ceres 1.9 covariance_impl.cc restated): the (frame, frame) block of (J^T J)^-1 with the
// loss function applied, on the tangent space — constant blocks and the fixed coordinates of a SubsetParameterization
// are removed before the inversion and come back as zero rows / columns.  The point blocks are eliminated first
// (same inverse, by the block-inverse formula).  Returns 0 if J^T J is rank deficient or the evaluation fails.
int32_t orc_pose_covariance(const orc_problem* p, int32_t frame, double* cov) {
  Eval E(p);
  if (!E.run(true, nullptr, nullptr)) return 0;
  const int CD = E.L.CD; const int64_t nc = E.ncam;
  PointCsr pc = point_csr(p);
  std::vector<double> D2(E.nparam, 0.0), rhs, Vinv, bp; EnvMatrix S;
  // fixed point coordinates: keep their (decoupled, otherwise singular) 3x3 blocks invertible
  for (int64_t a = nc; a < E.nparam; ++a) if (E.colmask[a]) D2[a] = 1.0;
  // a point that nobody observes has an all-zero block as well
  for (int j = 0; j < E.M; ++j) if (pc.ptr[j + 1] == pc.ptr[j]) for (int k = 0; k < 3; ++k) D2[nc + 3 * (int64_t)j + k] = 1.0;
  if (!reduced_system(E, pc, E.J, E.r, D2, S, rhs, Vinv, bp)) return 0;
  // free camera-side coordinates
  std::vector<int64_t> freec;
  for (int64_t a = 0; a < nc; ++a) if (!E.colmask[a]) freec.push_back(a);
  const int64_t nf = (int64_t)freec.size();
  std::vector<double> A((size_t)nf * nf);
  for (int64_t a = 0; a < nf; ++a) for (int64_t b = 0; b < nf; ++b) A[a * nf + b] = S.get(freec[a], freec[b]);
  if (!cholesky(A, nf)) return 0;
  for (int a = 0; a < CD * CD; ++a) cov[a] = 0.0;
  for (int k = 0; k < CD; ++k) {
    const int64_t g = (int64_t)frame * CD + k;
    const auto it = std::lower_bound(freec.begin(), freec.end(), g);
    if (it == freec.end() || *it != g) continue;
    std::vector<double> e(nf, 0.0);
    e[it - freec.begin()] = 1.0;
    chol_solve(A, nf, e);
    for (int a = 0; a < CD; ++a) {
      const int64_t ga = (int64_t)frame * CD + a;
      const auto ia = std::lower_bound(freec.begin(), freec.end(), ga);
      if (ia != freec.end() && *ia == ga) cov[a * CD + k] = e[ia - freec.begin()];
    }
  }
  return 1;
}

int32_t orc_pnp_task(const double cam[9], int32_t shutter, const int32_t scanlines[2], const float* object_points, const float* image_points,
                     int32_t n, const int32_t* subset, int32_t m, int32_t drop_coincident, const double init_poses[12], int32_t max_iter, double reprojection_error,
                     double poses_out[12], int32_t* usable, double* final_cost, int32_t* num_inliers, uint8_t* inlier_mask) {
  // solveRSpnp.cpp:283-293: hypotheses with two coincident 3-D points are dropped (float differences, norm in double)
  if (drop_coincident) for (int i = 0; i < m; ++i) for (int j = i + 1; j < m; ++j) {
    const float* a = object_points + 3 * (size_t)subset[i]; const float* b = object_points + 3 * (size_t)subset[j];
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    if (std::sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz) < 1e-10) return 0;
  }
  // solveRSpnp.cpp:127-161: one RsBA<float> block per point over (pose, pose2); float data widened to double
  std::vector<double> poses(init_poses, init_poses + 12), pts((size_t)3 * m), xy((size_t)2 * m), intr(cam, cam + 9);
  std::vector<int32_t> of(m, 0), op(m); std::vector<uint8_t> pconst(m, 1);
  for (int i = 0; i < m; ++i) {
    op[i] = i;
    for (int k = 0; k < 3; ++k) pts[3 * (size_t)i + k] = (double)object_points[3 * (size_t)subset[i] + k];
    for (int k = 0; k < 2; ++k) xy[2 * (size_t)i + k] = (double)image_points[2 * (size_t)subset[i] + k];
  }
  orc_problem P; std::memset(&P, 0, sizeof P);
  P.shutter = shutter; P.scanlines[0] = scanlines[0]; P.scanlines[1] = scanlines[1]; P.interpolate_rotation = 1; P.calibrated = 1;
  P.poses_per_frame = 2; P.num_frames = 1; P.num_points = m; P.num_intrinsics = 1; P.num_observations = m;
  P.poses = poses.data(); P.points = pts.data(); P.intrinsics = intr.data(); P.obs_xy = xy.data(); P.obs_frame = of.data(); P.obs_point = op.data();
  P.point_constant = pconst.data(); P.no_validate = 1; P.inter_frame_ratio = 1.0;
  orc_options o; orc_default_options(&o); o.max_num_iterations = max_iter; o.num_threads = 1;
  orc_summary s;
  const int32_t term = orc_solve(&P, &o, &s, nullptr, 0);
  const bool ok = term != ORC_FAILURE;                       // Summary::IsSolutionUsable
  for (int k = 0; k < 12; ++k) poses_out[k] = ok ? poses[k] : init_poses[k];   // :162-176: the Mats change only then
  if (usable) *usable = ok;
  if (final_cost) *final_cost = s.final_cost;
  // solveRSpnp.cpp:225-258 project3dPoints + :304-310
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    const double X[3] = {(double)object_points[3 * (size_t)i], (double)object_points[3 * (size_t)i + 1], (double)object_points[3 * (size_t)i + 2]};
    const float ix = image_points[2 * (size_t)i], iy = image_points[2 * (size_t)i + 1];
    const double obs[2] = {(double)ix, (double)iy};
    double pose[6], proj[2];
    interpolate_rs<double>(poses_out, poses_out + 6, shutter, scanlines, obs, pose, true);
    bool in = false;
    if (w2i<double>(cam, pose, X, proj, false)) {              // the reference aborts on failure; here: not an inlier
      const float dx = ix - (float)proj[0], dy = iy - (float)proj[1];
      in = std::sqrt((double)dx * dx + (double)dy * dy) < (double)(float)reprojection_error;
    }
    if (inlier_mask) inlier_mask[i] = in;
    cnt += in;
  }
  if (num_inliers) *num_inliers = cnt;
  return 1;
}

void orc_set_num_threads(int32_t n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

void orc_angle_axis_rotate(const double w[3], const double p[3], double out[3]) { angle_axis_rotate(w, p, out); }
void orc_lerp_rotation(const double r0[3], const double r1[3], double tau, double out[3]) { lerp_rotation(r0, r1, tau, out); }
void orc_distort(const double cam[9], const double img[2], double out[2]) { distort(cam, img, out); }
int32_t orc_undistort(const double cam[9], const double img[2], double out[2]) { return undistort(cam, img, out); }
void orc_w2c(const double pose[6], const double X[3], double out[3]) { w2c(pose, X, out); }
void orc_c2w(const double pose[6], const double pt[3], double out[3]) { c2w(pose, pt, out); }
int32_t orc_w2i(const double cam[9], const double pose[6], const double X[3], double out[2], int32_t v) { return w2i(cam, pose, X, out, v != 0); }
int32_t orc_direction_world(const double pose[6], const double X[3], double d[3]) { return direction_world(pose, X, d); }
int32_t orc_c2direction(const double pose[6], const double pt[3], double d[3]) { return c2direction(pose, pt, d); }
int32_t orc_direction_pixel(const double cam[9], const double pose[6], const double xy[2], double d[3]) { return direction_pixel(cam, pose, xy[0], xy[1], d); }
int32_t orc_ray_intersect(const double p2[3], const double d1[3], const double d2[3], double dist[3]) { return ray_intersect(p2, d1, d2, dist); }
int32_t orc_triangulate(const double c1[3], const double d1[3], const double c2[3], const double d2[3], double p[3]) { return triangulate(c1, d1, c2, d2, p); }
int32_t orc_validate(const double cam[9], const double pose[6], const double xy[2], const double X[3], double t) { return validate(cam, pose, xy, X, t); }
int32_t orc_ray_dist(const double cam[9], const double pose[6], const double obs[2], const double cam2[9], const double pose2[6], const double obs2[2], double dist[3]) { return ray_dist(cam, pose, obs, cam2, pose2, obs2, dist); }
double orc_norm3(const double v[3]) { return norm3(v); }
void orc_interpolate_rs(const double p0[6], const double p1[6], int32_t shutter, const int32_t scan[2], const double obs[2], int32_t ir, double out[6]) {
  const int sc[2] = {scan[0], scan[1]}; interpolate_rs(p0, p1, shutter, sc, obs, out, ir != 0); }
void orc_huber(double a, double s, double rho[3]) { huber(a, s, rho); }
int32_t orc_scanline_pose_index(int32_t nposes, int32_t shutter, const double obs[2]) { return scanline_pose_index(nposes, shutter, obs); }
int32_t orc_reproject(const double cam[9], const double* poses, int32_t nposes, int32_t shutter, const int32_t scan[2], int32_t ir, const double X[3], double sq, double obs[2]) {
  const int sc[2] = {scan[0], scan[1]}; return reproject(cam, poses, nposes, shutter, sc, ir != 0, X, sq, obs); }
int32_t orc_validate_obs(const double cam[9], const double* poses, int32_t nposes, int32_t shutter, const int32_t scan[2], int32_t ir, const double X[3], const double obs[2], double sq, double md) {
  const int sc[2] = {scan[0], scan[1]}; return validate_obs(cam, poses, nposes, shutter, sc, ir != 0, X, obs, sq, md); }

}  // extern "C"
