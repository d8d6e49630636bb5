// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU fp64 restatement of the arithmetic on henrique/rsba's bundle-adjustment hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
//
// PARITY STATUS: the geometry helpers below are pinned by the reference's own known-answer
// tests (src/rsba/test/mat_test.cc:23-313, re-expressed in tests/test_oracle_kat.py).  The
// reference holds NO test for interpolate_rs / RsBundleAdjustment / Jacobians / Huber / the
// Ceres solve, and Ceres-Solver 1.9.0 (the un-vendored dependency that owns AngleAxisRotatePoint,
// Jet autodiff, the loss corrector and the LM loop) is not in /root/reference and not installed:
// for those parts this oracle is "PARITY UNPINNED" by the reference and is instead cross-checked
// against an independent high-precision implementation (tests/golden/make_golden.py).
//
// Every function cites the reference file:line it follows.  All reference paths are relative to
// /root/reference/src/rsba/.
#pragma once
#include <cmath>
#include <cstddef>
#include <limits>

namespace rsba_oracle {

// mat/core.h:12  (_EPS = __DBL_EPSILON__)
constexpr double kEps = std::numeric_limits<double>::epsilon();

// mat/cam.h:19-34 block sizes and intrinsics order {fx,fy,k1,k2,p1,p2,k3,cx,cy}
enum { kPose = 6, kPoint = 3, kCam = 9 };
enum CamIdx { FX = 0, FY = 1, K1 = 2, K2 = 3, P1 = 4, P2 = 5, K3 = 6, CX = 7, CY = 8 };
// mat/cam.h:37-41
enum Shutter { GLOBAL = 0, HORIZONTAL = 1, VERTICAL = 2 };

// ---------------------------------------------------------------------------------------------
// Forward-mode dual number of fixed width N: the stand-in for ceres::Jet<double,N>
// (Ceres-Solver 1.9.0 include/ceres/jet.h; third-party, restated from its published definition:
// value + N infinitesimal parts, comparisons act on the value only — SURVEY Appendix C.2).
// ---------------------------------------------------------------------------------------------
template <int N>
struct Dual {
  double a;
  double v[N];
  Dual() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Dual(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT implicit like Jet
  Dual(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};

template <int N> inline Dual<N> operator+(const Dual<N>& x, const Dual<N>& y) {
  Dual<N> r; r.a = x.a + y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& x, const Dual<N>& y) {
  Dual<N> r; r.a = x.a - y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& x) {
  Dual<N> r; r.a = -x.a; for (int i = 0; i < N; ++i) r.v[i] = -x.v[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& x, const Dual<N>& y) {
  Dual<N> r; r.a = x.a * y.a; for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& x, const Dual<N>& y) {
  // jet.h: d(x/y) = (dx - (x/y) dy) / y
  Dual<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv;
  for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& x, double s) { Dual<N> r = x; r.a += s; return r; }
template <int N> inline Dual<N> operator+(double s, const Dual<N>& x) { Dual<N> r = x; r.a += s; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& x, double s) { Dual<N> r = x; r.a -= s; return r; }
template <int N> inline Dual<N> operator-(double s, const Dual<N>& x) { Dual<N> r = -x; r.a += s; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& x, double s) {
  Dual<N> r; r.a = x.a * s; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * s; return r; }
template <int N> inline Dual<N> operator*(double s, const Dual<N>& x) { return x * s; }
template <int N> inline Dual<N> operator/(const Dual<N>& x, double s) { return x * (1.0 / s); }
template <int N> inline Dual<N> operator/(double s, const Dual<N>& y) {
  Dual<N> r; r.a = s / y.a; const double m = -s / (y.a * y.a);
  for (int i = 0; i < N; ++i) r.v[i] = m * y.v[i]; return r; }
template <int N> inline Dual<N>& operator+=(Dual<N>& x, const Dual<N>& y) { x = x + y; return x; }
template <int N> inline Dual<N>& operator*=(Dual<N>& x, const Dual<N>& y) { x = x * y; return x; }
template <int N> inline bool operator<(const Dual<N>& x, const Dual<N>& y) { return x.a < y.a; }
template <int N> inline bool operator>(const Dual<N>& x, const Dual<N>& y) { return x.a > y.a; }
template <int N> inline bool operator<(const Dual<N>& x, double y) { return x.a < y; }
template <int N> inline bool operator>(const Dual<N>& x, double y) { return x.a > y; }
template <int N> inline Dual<N> sqrt(const Dual<N>& x) {
  Dual<N> r; r.a = std::sqrt(x.a); const double m = 1.0 / (2.0 * r.a);
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * m; return r; }
template <int N> inline Dual<N> sin(const Dual<N>& x) {
  Dual<N> r; r.a = std::sin(x.a); const double m = std::cos(x.a);
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * m; return r; }
template <int N> inline Dual<N> cos(const Dual<N>& x) {
  Dual<N> r; r.a = std::cos(x.a); const double m = -std::sin(x.a);
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * m; return r; }
template <int N> inline Dual<N> abs(const Dual<N>& x) { return x.a < 0.0 ? -x : x; }
inline double value_of(double x) { return x; }
template <int N> inline double value_of(const Dual<N>& x) { return x.a; }

using std::sqrt; using std::sin; using std::cos; using std::abs;

// ---------------------------------------------------------------------------------------------
// Ceres-Solver 1.9.0 include/ceres/rotation.h AngleAxisRotatePoint (third-party, restated —
// SURVEY Appendix C.1).  Rodrigues when theta^2 > DBL_EPSILON, first-order p + w x p otherwise.
// Every output component is formed from temporaries, so result may alias pt (mat/cam.h:365 and
// mat_test.cc:52 call it in place).
// ---------------------------------------------------------------------------------------------
template <class T>
inline void angle_axis_rotate(const T w[3], const T p[3], T out[3]) {
  const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > T(kEps)) {
    const T th = sqrt(th2);
    const T c = cos(th), s = sin(th);
    const T ith = T(1.0) / th;
    const T k[3] = {w[0] * ith, w[1] * ith, w[2] * ith};
    const T kxp[3] = {k[1] * p[2] - k[2] * p[1], k[2] * p[0] - k[0] * p[2], k[0] * p[1] - k[1] * p[0]};
    const T kdp = (k[0] * p[0] + k[1] * p[1] + k[2] * p[2]) * (T(1.0) - c);
    const T o0 = p[0] * c + kxp[0] * s + k[0] * kdp;
    const T o1 = p[1] * c + kxp[1] * s + k[1] * kdp;
    const T o2 = p[2] * c + kxp[2] * s + k[2] * kdp;
    out[0] = o0; out[1] = o1; out[2] = o2;
  } else {
    const T wxp[3] = {w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]};
    const T o0 = p[0] + wxp[0], o1 = p[1] + wxp[1], o2 = p[2] + wxp[2];
    out[0] = o0; out[1] = o1; out[2] = o2;
  }
}

// mat/core.h:164-167 norm3, :155-158 3-arg norm
template <class T> inline T norm3(const T v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
template <class T> inline T norm_xyz(const T& x, const T& y, const T& z) { return sqrt(x * x + y * y + z * z); }
// mat/core.h:170-177 normalize3
template <class T> inline bool normalize3(const T v[3], T out[3]) {
  const T n = norm3(v);
  if (n < T(kEps)) return false;
  const T inv = T(1.0) / n;
  out[0] = v[0] * inv; out[1] = v[1] * inv; out[2] = v[2] * inv;
  return true;
}
// mat/core.h:188-190 dist3, :181-183 dist2
template <class T> inline T dist3(const T a[3], const T b[3]) { return norm_xyz(T(a[0] - b[0]), T(a[1] - b[1]), T(a[2] - b[2])); }
inline double dist2(const double a[2], const double b[2]) { return std::sqrt((a[0]-b[0])*(a[0]-b[0]) + (a[1]-b[1])*(a[1]-b[1])); }

// mat/core.h:20-48 det33 / inv33 (row-major 3x3; singular if |det| < eps)
inline double det33(const double m[9]) {
  return m[0] * (m[4] * m[8] - m[7] * m[5]) - m[1] * (m[3] * m[8] - m[6] * m[5]) + m[2] * (m[3] * m[7] - m[6] * m[4]);
}
inline bool inv33(const double m[9], double o[9]) {
  const double d = det33(m);
  if (d < kEps && d > -kEps) return false;
  o[0] = (m[4] * m[8] - m[7] * m[5]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
  o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[3] * m[2] - m[0] * m[5]) / d;
  o[6] = (m[3] * m[7] - m[6] * m[4]) / d; o[7] = (m[6] * m[1] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[3] * m[1]) / d;
  return true;
}

// mat/cam.h:48-72 distort: Brown model, d = 1 + r2(k1 + r2(k2 + r2 k3)), tangential p1,p2
template <class T>
inline void distort(const T cam[kCam], const T img[2], T out[2]) {
  const T x = img[0], y = img[1];
  const T r2 = x * x + y * y;
  const T d = T(1.0) + r2 * (cam[K1] + r2 * (cam[K2] + (r2 * cam[K3])));
  const T two(2.0);
  const T xy = x * y;
  const T o0 = (d * x) + (two * cam[P1] * xy + cam[P2] * (r2 + two * x * x));
  const T o1 = (d * y) + (cam[P1] * (r2 + two * y * y) + two * cam[P2] * xy);
  out[0] = o0; out[1] = o1;
}

// mat/cam.h:77-112 undistort: fixed-point iteration p_u -= distort(p_u) - p_n, <=200 iterations,
// break when the error exceeds |p_n| (diverging), valid once the error < |p_n| * 0.001 / fx.
inline bool undistort(const double cam[kCam], const double img[2], double out[2]) {
  const double pn[2] = {img[0], img[1]};
  double pu[2] = {img[0], img[1]};
  const double nrm = std::sqrt(pn[0] * pn[0] + pn[1] * pn[1]);
  const double tol = nrm * 0.001 / cam[FX];
  bool valid = false;
  for (int it = 0; it < 200; ++it) {
    double pd[2];
    distort(cam, pu, pd);
    const double e[2] = {pd[0] - pn[0], pd[1] - pn[1]};
    const double dist = std::sqrt(e[0] * e[0] + e[1] * e[1]);
    if (dist > nrm) break;
    pu[0] -= e[0]; pu[1] -= e[1];
    if (dist < tol) { valid = true; break; }
  }
  out[0] = pu[0]; out[1] = pu[1];
  return valid;
}

// mat/cam.h:250-288 "slerp" — the live code is LINEAR interpolation of the angle-axis vectors
template <class T>
inline void lerp_rotation(const T r0[3], const T r1[3], const T& tau, T out[3]) {
  out[0] = r0[0] + (r1[0] - r0[0]) * tau;
  out[1] = r0[1] + (r1[1] - r0[1]) * tau;
  out[2] = r0[2] + (r1[2] - r0[2]) * tau;
}

// mat/cam.h:293-311 interpolate: rotation lerped (or copied from pose0), translation lerped
template <class T>
inline void interpolate(const T p0[kPose], const T p1[kPose], const T& tau, T out[kPose], bool interp_rotation) {
  if (interp_rotation) lerp_rotation(p0, p1, tau, out);
  else { out[0] = p0[0]; out[1] = p0[1]; out[2] = p0[2]; }
  out[3] = p0[3] + (p1[3] - p0[3]) * tau;
  out[4] = p0[4] + (p1[4] - p0[4]) * tau;
  out[5] = p0[5] + (p1[5] - p0[5]) * tau;
}

// mat/cam.h:315-349 interpolate_rs: GLOBAL copies pose0; VERTICAL uses obs[1], anything else obs[0];
// tau = (obs - scan0) / T(scan1 - scan0) (integer subtraction first), clamped to [0,1].
template <class T>
inline void interpolate_rs(const T p0[kPose], const T p1[kPose], int shutter, const int scan[2], const T obs[2],
                           T out[kPose], bool interp_rotation) {
  if (shutter == GLOBAL) { for (int i = 0; i < kPose; ++i) out[i] = p0[i]; return; }
  T tau;
  if (shutter == VERTICAL) tau = (obs[1] - T(double(scan[0]))) / T(double(scan[1] - scan[0]));
  else tau = (obs[0] - T(double(scan[0]))) / T(double(scan[1] - scan[0]));
  if (tau < T(0.0)) tau = T(0.0);
  if (tau > T(1.0)) tau = T(1.0);
  interpolate(p0, p1, tau, out, interp_rotation);
}

// mat/cam.h:354-366 w2c: pt = R(pose[0..2]) * (point - pose[3..5]), rotation applied in place
template <class T>
inline void w2c(const T pose[kPose], const T X[3], T pt[3]) {
  pt[0] = X[0] - pose[3]; pt[1] = X[1] - pose[4]; pt[2] = X[2] - pose[5];
  angle_axis_rotate(pose, pt, pt);
}

// mat/cam.h:116-126 c2w: p = R(-r) pt + c
template <class T>
inline void c2w(const T pose[kPose], const T pt[3], T p[3]) {
  const T inv[3] = {-pose[0], -pose[1], -pose[2]};
  angle_axis_rotate(inv, pt, p);
  p[0] = p[0] + pose[3]; p[1] = p[1] + pose[4]; p[2] = p[2] + pose[5];
}

// mat/cam.h:371-395 c2i: reject |z| < eps; dehomogenise; distort; scale by focal; add centre
template <class T>
inline bool c2i(const T cam[kCam], const T pt[3], T proj[2]) {
  if (pt[2] < T(kEps) && pt[2] > T(-kEps)) return false;
  const T img[2] = {pt[0] / pt[2], pt[1] / pt[2]};
  distort(cam, img, proj);
  proj[0] = proj[0] * cam[FX]; proj[1] = proj[1] * cam[FY];
  proj[0] = proj[0] + cam[CX]; proj[1] = proj[1] + cam[CY];
  return true;
}

// mat/cam.h:400-419 w2i: z < 1e-8 fails when validating, else a numerically-zero z becomes eps
template <class T>
inline bool w2i(const T cam[kCam], const T pose[kPose], const T X[3], T proj[2], bool validate = true) {
  T pt[3];
  w2c(pose, X, pt);
  if (pt[2] < T(1e-8)) {
    if (validate) return false;
    if (pt[2] < T(kEps) && pt[2] > T(-kEps)) pt[2] = T(kEps);
  }
  return c2i(cam, pt, proj);
}

// mat/cam.h:424-439 reprojection_error, :444-457 validate (double only in the reference)
inline bool reprojection_error(const double cam[kCam], const double pose[kPose], const double xy[2], const double X[3], double& sq) {
  double proj[2];
  if (!w2i(cam, pose, X, proj)) return false;
  const double dx = proj[0] - xy[0], dy = proj[1] - xy[1];
  sq = dx * dx + dy * dy;
  return true;
}
inline bool validate(const double cam[kCam], const double pose[kPose], const double xy[2], const double X[3], double sq_threshold) {
  double sq;
  return reprojection_error(cam, pose, xy, X, sq) && sq < sq_threshold;
}

// mat/cam.h:130-138 c2direction, :142-150 direction (world), :154-176 direction (from pixel)
inline bool c2direction(const double pose[kPose], const double pt[3], double d[3]) {
  const double inv[3] = {-pose[0], -pose[1], -pose[2]};
  angle_axis_rotate(inv, pt, d);
  return normalize3(d, d);
}
inline bool direction_world(const double pose[kPose], const double X[3], double d[3]) {
  d[0] = X[0] - pose[3]; d[1] = X[1] - pose[4]; d[2] = X[2] - pose[5];
  return normalize3(d, d);
}
inline bool direction_pixel(const double cam[kCam], const double pose[kPose], double x0, double y0, double d[3], bool check = true) {
  if (cam[FX] < kEps) return false;
  if (cam[FY] < kEps) return false;
  d[0] = (x0 - cam[CX]) / cam[FX];
  d[1] = (y0 - cam[CY]) / cam[FY];
  d[2] = 1.0;
  if (!undistort(cam, d, d) && check) return false;   // undistort only touches d[0..1]
  return c2direction(pose, d, d);
}

// mat/cam.h:462-498 ray_intersect(p2,d1,d2): c = d1 x d2 normalised; solve [d1 d2 c] via inv33
inline bool ray_intersect(const double p2[3], const double d1[3], const double d2[3], double dist[3]) {
  double c[3] = {d1[1] * d2[2] - d1[2] * d2[1], d1[2] * d2[0] - d1[0] * d2[2], d1[0] * d2[1] - d1[1] * d2[0]};
  const double n = norm3(c);
  if (n < 3 * kEps) return false;
  const double inv = 1.0 / n;
  c[0] *= inv; c[1] *= inv; c[2] *= inv;
  const double m[9] = {d1[0], d2[0], c[0], d1[1], d2[1], c[1], d1[2], d2[2], c[2]};
  double mi[9];
  if (!inv33(m, mi)) return false;
  dist[0] = mi[0] * p2[0] + mi[1] * p2[1] + mi[2] * p2[2];
  dist[1] = -(mi[3] * p2[0] + mi[4] * p2[1] + mi[5] * p2[2]);
  dist[2] = mi[6] * p2[0] + mi[7] * p2[1] + mi[8] * p2[2];
  return true;
}

// mat/cam.h:188-231 triangulate: A = (I - a a^T) + (I - b b^T); x = A^-1 ((I-aa^T)c1 + (I-bb^T)c2)
// (the reference aborts when det(A) < eps; here that is reported as failure)
inline bool triangulate(const double c1[3], const double a[3], const double c2[3], const double b[3], double p[3]) {
  double A[9], Pa[9], Pb[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    const double I = (i == j) ? 1.0 : 0.0;
    Pa[3 * i + j] = I - a[i] * a[j];
    Pb[3 * i + j] = I - b[i] * b[j];
    A[3 * i + j] = Pa[3 * i + j] + Pb[3 * i + j];
  }
  if (det33(A) < kEps) return false;
  double y[3], Ai[9];
  for (int i = 0; i < 3; ++i)
    y[i] = Pa[3 * i] * c1[0] + Pa[3 * i + 1] * c1[1] + Pa[3 * i + 2] * c1[2] + Pb[3 * i] * c2[0] + Pb[3 * i + 1] * c2[1] + Pb[3 * i + 2] * c2[2];
  if (!inv33(A, Ai)) return false;
  for (int i = 0; i < 3; ++i) p[i] = Ai[3 * i] * y[0] + Ai[3 * i + 1] * y[1] + Ai[3 * i + 2] * y[2];
  return true;
}

// mat/cam.h:521-576 rayDist
inline bool ray_dist(const double cam[kCam], const double pose[kPose], const double obs[2],
                     const double cam2[kCam], const double pose2[kPose], const double obs2[2], double dist[3]) {
  double d1[3], d2[3];
  if (!direction_pixel(cam, pose, obs[0], obs[1], d1)) return false;
  if (!direction_pixel(cam2, pose2, obs2[0], obs2[1], d2)) return false;
  const double p2[3] = {pose2[3] - pose[3], pose2[4] - pose[4], pose2[5] - pose[5]};
  double len[3];
  if (!ray_intersect(p2, d1, d2, len)) { dist[0] = p2[0]; dist[1] = p2[1]; dist[2] = p2[2]; return true; }
  const double l = len[0], k = len[1];
  dist[0] = l * d1[0] - (p2[0] + k * d2[0]);
  dist[1] = l * d1[1] - (p2[1] + k * d2[1]);
  dist[2] = l * d1[2] - (p2[2] + k * d2[2]);
  return true;
}

// ---------------------------------------------------------------------------------------------
// Cost functors
// ---------------------------------------------------------------------------------------------
// video_bundler_free.h:44-65 ReprojectionError::operator()(camera,pose,point): w2i with validate,
// residual = projection - observation; the 5-px branch returns true on both arms.
template <class T>
inline bool gs_residual(const T cam[kCam], const T pose[kPose], const T X[3], double ox, double oy, T res[2], bool validate = true) {
  T proj[2];
  if (!w2i(cam, pose, X, proj, validate)) return false;
  res[0] = proj[0] - T(ox);
  res[1] = proj[1] - T(oy);
  return true;
}

// VideoSfmBaRs.h:25-49 RsBundleAdjustment::operator(): obs[2] = {observed_x, observed_x} — BOTH
// entries are x (reference quirk, kept), pose = interpolate_rs(...), then the GS residual.
template <class T>
inline bool rs_residual(const T cam[kCam], const T p0[kPose], const T p1[kPose], const T X[3], double ox, double oy,
                        int shutter, const int scan[2], bool interp_rotation, T res[2], bool validate = true) {
  T pose[kPose];
  const T obs[2] = {T(ox), T(ox)};
  interpolate_rs(p0, p1, shutter, scan, obs, pose, interp_rotation);
  return gs_residual(cam, pose, X, ox, oy, res, validate);   // validate = false: solveRSpnp.cpp:67 (RsBA)
}

// struct/VideoSfM.cc:83-97 = :118-132, the `default:` branch of both getPose overloads — a frame with MORE than two poses
// carries one per scan line ("fullDoF") and an observation uses the pose of its rounded, clamped line:
//   line = obs[0] if sess.rs == HORIZONTAL else obs[1]   (so VERTICAL *and* GLOBAL read y);
//   line < 0 -> 0;  line > size - 1 -> size - 1 (size_t converted to double);  poses[round(line)]  (halves away from zero)
inline int scanline_pose_index(int nposes, int shutter, const double obs[2]) {
  double line = (shutter == HORIZONTAL) ? obs[0] : obs[1];
  if (line < 0) line = 0;
  else if (line > (double)(nposes - 1)) line = (double)(nposes - 1);
  return (int)std::round(line);
}

// struct/VideoSfM.cc:103-133 getPose (copying overload): the non-autodiff twin
// used by reproject/validate; unlike the functor it feeds the TRUE (x,y) to interpolate_rs.
inline void frame_pose_at(const double* poses, int nposes, int shutter, const int scan[2], const double obs[2],
                          bool interp_rotation, double out[kPose]) {
  if (nposes == 1) { for (int i = 0; i < kPose; ++i) out[i] = poses[i]; return; }
  if (nposes > 2) { const double* q = poses + (size_t)kPose * scanline_pose_index(nposes, shutter, obs); for (int i = 0; i < kPose; ++i) out[i] = q[i]; return; }
  interpolate_rs(poses, poses + kPose, shutter, scan, obs, out, interp_rotation);
}

// struct/VideoSfM.cc:139-155 reproject: fixed point on tau starting from the principal point;
// at most 49 w2i evaluations; stop when the projection moves < 1e-3 px (squared 1e-6).
inline bool reproject(const double cam[kCam], const double* poses, int nposes, int shutter, const int scan[2],
                      bool interp_rotation, const double X[3], double sq_threshold, double obs[2]) {
  double pose[kPose];
  double prev[2], proj[2] = {cam[CX], cam[CY]};
  int limit = 50;
  do {
    if (--limit < 1) return false;
    prev[0] = proj[0]; prev[1] = proj[1];
    frame_pose_at(poses, nposes, shutter, scan, proj, interp_rotation, pose);
    if (!w2i(cam, pose, X, proj, true)) return false;
    prev[0] -= proj[0]; prev[1] -= proj[1];
  } while (nposes > 1 && prev[0] * prev[0] + prev[1] * prev[1] > 1e-6);
  obs[0] = proj[0]; obs[1] = proj[1];
  return validate(cam, pose, obs, X, sq_threshold);
}

// struct/VideoSfM.cc:159-169 validate(sess,f,opt,pt,obs): distance and reprojection gates
inline bool validate_obs(const double cam[kCam], const double* poses, int nposes, int shutter, const int scan[2],
                         bool interp_rotation, const double X[3], const double obs[2], double sq_threshold, double min_dist) {
  double pose[kPose];
  frame_pose_at(poses, nposes, shutter, scan, obs, interp_rotation, pose);
  const double d[3] = {pose[3] - X[0], pose[4] - X[1], pose[5] - X[2]};
  return norm3(d) >= min_dist && validate(cam, pose, obs, X, sq_threshold);
}

// Ceres-Solver 1.9.0 loss_function.cc HuberLoss::Evaluate (third-party, restated — SURVEY C.3):
// b = a^2; s <= b: rho = (s,1,0); else r = sqrt(s): rho = (2 a r - b, max(DBL_MIN, a/r), -rho1/(2 s)).
inline void huber(double a, double s, double rho[3]) {
  const double b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = std::fmax(std::numeric_limits<double>::min(), a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

}  // namespace rsba_oracle
