// Per-observation rolling-shutter reprojection residual and its ANALYTIC Jacobian, fp64, gfx950.
//
// Replaces, for the four functor shapes rsba instantiates, the Jet<double,K> pass that
// ceres::AutoDiffCostFunction runs over
//   RsBundleAdjustment::operator()      /root/reference/src/rsba/VideoSfmBaRs.h:25-49
//   ReprojectionError::operator()       /root/reference/src/rsba/video_bundler_free.h:32-65
//   interpolate_rs / interpolate / slerp /root/reference/src/rsba/mat/cam.h:315-349, 293-311, 250-288
//   w2i / w2c / c2i / distort           /root/reference/src/rsba/mat/cam.h:400-419, 354-366, 371-395, 48-72
//   ceres::AngleAxisRotatePoint         Ceres-Solver 1.9.0 rotation.h (third-party)
// Arithmetic follows SURVEY.md Appendix A step by step.  Because tau depends on the observation only,
// the 2x(6+6+3) block is the 2x(6+3) pinhole block scaled by (1-tau) / tau.
#pragma once
#include <hip/hip_runtime.h>

namespace rsba {

constexpr double kDblEps = 2.220446049250313e-16;   // mat/core.h:12 _EPS
constexpr double kMinDepth = 1e-8;                  // mat/cam.h:410

enum Shutter : int { kGlobal = 0, kHorizontal = 1, kVertical = 2 };   // mat/cam.h:37-41

struct Model {
  int shutter;
  int scan0, scan1;
  int interp_rotation;
};

// Output of one observation.  J rows are laid out [cam 9]? [pose0 6] [pose1 6]? [point 3].
template <bool CAL, int P>
struct ObsOut {
  static constexpr int K = (CAL ? 0 : 9) + 6 * P + 3;
  double r[2];
  double J[2][K];
  bool ok;
  // P == 2, WANT_J: the block BEFORE the (1 - tau) / tau scaling — the 2 x 6 Jacobian with respect to the interpolated pose
  // [d/dr | d/dt] — and tau itself: the factored P records of the point elimination (kernels_normal.hip, project_rc_kernel) store
  // Jq^T Jp L^-T once per observation instead of its two scaled copies.  Dead code wherever nobody reads them.
  double Jq[2][6];
  double tau;
};

// R = exp([w]x), p = R q, and D = d(R q)/dw.  Same branch as AngleAxisRotatePoint: Rodrigues when
// |w|^2 > DBL_EPSILON, first-order (p = q + w x q, D = -[q]x) otherwise — the Jet path differentiates
// whichever branch it took, so the analytic form must branch identically.
template <bool WANT_D>
__device__ __forceinline__ void rotate_with_derivative(const double w[3], const double q[3], double p[3],
                                                        double R[3][3], double D[3][3]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double wxq[3] = {w[1] * q[2] - w[2] * q[1], w[2] * q[0] - w[0] * q[2], w[0] * q[1] - w[1] * q[0]};
  if (th2 > kDblEps) {
    const double th = sqrt(th2);
    double s, c;
    sincos(th, &s, &c);
    const double ith2 = 1.0 / th2;
    const double a = s / th;
    const double b = (1.0 - c) * ith2;
    const double wq = w[0] * q[0] + w[1] * q[1] + w[2] * q[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = c * q[i] + a * wxq[i] + b * wq * w[i];
    if (WANT_D) {
      // R = c I + a [w]x + b w w^T
      R[0][0] = c + b * w[0] * w[0];        R[0][1] = -a * w[2] + b * w[0] * w[1]; R[0][2] = a * w[1] + b * w[0] * w[2];
      R[1][0] = a * w[2] + b * w[1] * w[0]; R[1][1] = c + b * w[1] * w[1];         R[1][2] = -a * w[0] + b * w[1] * w[2];
      R[2][0] = -a * w[1] + b * w[2] * w[0]; R[2][1] = a * w[0] + b * w[2] * w[1]; R[2][2] = c + b * w[2] * w[2];
      // D = m w^T - a [q]x + b (w.q) I + b w q^T,  m = -a q + (c-a)/th2 (w x q) + (a-2b)/th2 (w.q) w
      const double ca = (c - a) * ith2, ab = (a - 2.0 * b) * ith2 * wq;
      double m[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) m[i] = -a * q[i] + ca * wxq[i] + ab * w[i];
      const double bwq = b * wq;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) D[i][j] = m[i] * w[j] + b * w[i] * q[j];
      D[0][0] += bwq; D[1][1] += bwq; D[2][2] += bwq;
      D[0][1] += a * q[2];  D[0][2] -= a * q[1];
      D[1][0] -= a * q[2];  D[1][2] += a * q[0];
      D[2][0] += a * q[1];  D[2][1] -= a * q[0];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = q[i] + wxq[i];
    if (WANT_D) {
      R[0][0] = 1.0;   R[0][1] = -w[2]; R[0][2] = w[1];
      R[1][0] = w[2];  R[1][1] = 1.0;   R[1][2] = -w[0];
      R[2][0] = -w[1]; R[2][1] = w[0];  R[2][2] = 1.0;
      D[0][0] = 0.0;   D[0][1] = q[2];  D[0][2] = -q[1];
      D[1][0] = -q[2]; D[1][1] = 0.0;   D[1][2] = q[0];
      D[2][0] = q[1];  D[2][1] = -q[0]; D[2][2] = 0.0;
    }
  }
}

// tau of SURVEY Appendix A step 1.  The functor feeds (observed_x, observed_x) to interpolate_rs
// (VideoSfmBaRs.h:31,45), so tau comes from x for HORIZONTAL *and* VERTICAL — reference quirk, kept.
__device__ __forceinline__ double scanline_tau(const Model& m, double ox) {
  if (m.shutter == kGlobal) return 0.0;
  double tau = (ox - double(m.scan0)) / double(m.scan1 - m.scan0);   // cam.h:326/329, integer subtraction first
  if (tau < 0.0) tau = 0.0;                                          // cam.h:341-345
  if (tau > 1.0) tau = 1.0;
  return tau;
}

// One observation.  pose points at the frame's P consecutive 6-vectors (LDS or global), X at the
// point, cam at the 9 intrinsics.  WANT_J=false is the T=double path (residuals only).
template <bool CAL, int P, bool WANT_J>
__device__ __forceinline__ void eval_observation(const Model& m, const double* __restrict__ cam,
                                                 const double* __restrict__ pose, const double* __restrict__ X,
                                                 double ox, double oy, ObsOut<CAL, P>& o) {
  constexpr int OFF_POSE = CAL ? 0 : 9;
  constexpr int OFF_PT = OFF_POSE + 6 * P;
  double w[3], t[3], tau = 0.0;
  if (P == 2) {
    tau = scanline_tau(m, ox);
    const bool lerp_rot = m.interp_rotation && m.shutter != kGlobal;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // cam.h:266-268 (rotation) and :308-310 (translation): a + (b - a) * tau
      w[i] = lerp_rot ? pose[i] + (pose[6 + i] - pose[i]) * tau : pose[i];
      t[i] = (m.shutter == kGlobal) ? pose[3 + i] : pose[3 + i] + (pose[9 + i] - pose[3 + i]) * tau;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) { w[i] = pose[i]; t[i] = pose[3 + i]; }
  }
  const double q[3] = {X[0] - t[0], X[1] - t[1], X[2] - t[2]};          // cam.h:360-362
  double p[3], R[3][3], D[3][3];
  rotate_with_derivative<WANT_J>(w, q, p, R, D);                         // cam.h:365
  // cam.h:410-412 (validate=true): z < 1e-8 => the functor returns false.  No early exit — the lanes
  // of a wave run in lock-step anyway, and a straight-line body keeps every output in registers; the
  // caller discards (zeroes) the outputs of a failed observation.
  o.ok = !(p[2] < kMinDepth);
  const double fx = cam[0], fy = cam[1], k1 = cam[2], k2 = cam[3], p1 = cam[4], p2 = cam[5], k3 = cam[6];
  const double iz = 1.0 / p[2];
  const double x = p[0] * iz, y = p[1] * iz;                             // cam.h:381-384
  const double r2 = x * x + y * y;                                       // cam.h:65
  const double d = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));                // cam.h:66
  const double xy = x * y;
  const double xd = d * x + (2.0 * p1 * xy + p2 * (r2 + 2.0 * x * x));   // cam.h:70
  const double yd = d * y + (p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * xy);   // cam.h:71
  o.r[0] = fx * xd + cam[7] - ox;                                        // cam.h:390-393, video_bundler_free.h:56-57
  o.r[1] = fy * yd + cam[8] - oy;
  if (!WANT_J) return;

  // d(xd,yd)/d(x,y)
  const double g = k1 + r2 * (2.0 * k2 + 3.0 * k3 * r2);                 // dd/dr2
  const double a00 = fx * (d + 2.0 * x * x * g + 2.0 * p1 * y + 6.0 * p2 * x);
  const double a01 = fx * (2.0 * xy * g + 2.0 * p1 * x + 2.0 * p2 * y);
  const double a10 = fy * (2.0 * xy * g + 2.0 * p1 * x + 2.0 * p2 * y);
  const double a11 = fy * (d + 2.0 * y * y * g + 6.0 * p1 * y + 2.0 * p2 * x);
  // B = A * d(x,y)/dp  (2x3)
  const double B[2][3] = {{a00 * iz, a01 * iz, -(a00 * x + a01 * y) * iz},
                          {a10 * iz, a11 * iz, -(a10 * x + a11 * y) * iz}};
  double Jw[2][3], JX[2][3];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Jw[rr][j] = B[rr][0] * D[0][j] + B[rr][1] * D[1][j] + B[rr][2] * D[2][j];
      JX[rr][j] = B[rr][0] * R[0][j] + B[rr][1] * R[1][j] + B[rr][2] * R[2][j];
    }
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    if (P == 2) {
      const double s0 = 1.0 - tau, s1 = tau;
      const bool lerp_rot = m.interp_rotation && m.shutter != kGlobal;
      o.tau = tau;
#pragma unroll
      for (int j = 0; j < 3; ++j) { o.Jq[rr][j] = Jw[rr][j]; o.Jq[rr][3 + j] = -JX[rr][j]; }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        o.J[rr][OFF_POSE + j] = lerp_rot ? s0 * Jw[rr][j] : Jw[rr][j];       // d/d r0
        o.J[rr][OFF_POSE + 3 + j] = -s0 * JX[rr][j];                         // d/d t0
        o.J[rr][OFF_POSE + 6 + j] = lerp_rot ? s1 * Jw[rr][j] : 0.0;         // d/d r1
        o.J[rr][OFF_POSE + 9 + j] = -s1 * JX[rr][j];                         // d/d t1
      }
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) { o.J[rr][OFF_POSE + j] = Jw[rr][j]; o.J[rr][OFF_POSE + 3 + j] = -JX[rr][j]; }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) o.J[rr][OFF_PT + j] = JX[rr][j];
  }
  if (!CAL) {
    const double r4 = r2 * r2, r6 = r4 * r2;
    o.J[0][0] = xd;            o.J[1][0] = 0.0;                   // fx
    o.J[0][1] = 0.0;           o.J[1][1] = yd;                    // fy
    o.J[0][2] = fx * x * r2;   o.J[1][2] = fy * y * r2;           // k1
    o.J[0][3] = fx * x * r4;   o.J[1][3] = fy * y * r4;           // k2
    o.J[0][4] = fx * 2.0 * xy; o.J[1][4] = fy * (r2 + 2.0 * y * y);   // p1
    o.J[0][5] = fx * (r2 + 2.0 * x * x); o.J[1][5] = fy * 2.0 * xy;   // p2
    o.J[0][6] = fx * x * r6;   o.J[1][6] = fy * y * r6;           // k3
    o.J[0][7] = 1.0;           o.J[1][7] = 0.0;                   // cx
    o.J[0][8] = 0.0;           o.J[1][8] = 1.0;                   // cy
  }
}

// Ceres-Solver 1.9.0 HuberLoss::Evaluate (third-party)
__device__ __forceinline__ void huber_rho(double a, double s, double rho[3]) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = fmax(2.2250738585072014e-308, a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

}  // namespace rsba
