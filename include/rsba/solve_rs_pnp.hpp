// Host-side mirror of rsba's rolling-shutter PnP entry points, over librsba_amd:
//   vision::solveRsPnP          /root/reference/src/rsba/solveRSpnp.cpp:100-192  (solveRSpnp.h:14-26)
//   vision::solveRsPnPRansac    /root/reference/src/rsba/solveRSpnp.cpp:413-524  (solveRSpnp.h:29-46)
// Same names, argument order and meaning, with plain arrays where the reference takes cv::Mat (OpenCV is not a
// dependency here): object points [n][3] float, image points [n][2] float, the 9 sfm intrinsics instead of
// cameraMatrix + distCoeffs (what sfmCam() makes of them), rvec / tvec / rvec2 / tvec2 as double[3] in OpenCV's
// convention (x_cam = R(rvec) X + tvec).  What the reference runs sequentially — one ceres::Solve per random subset
// (pnpTask, :265-335) — is ONE device launch here (rsba_pnp_tasks); the subsets come from the same generator
// (cv::RNG's multiply-with-carry, restated) walked in the same order, and the winner is chosen by replaying the
// reference's sequential rule (first strictly better hypothesis wins, stop once minInliersCount is reached) over the
// batch's inlier counts, so the result is the one the single-threaded reference would return for these subsets.
// The global-shutter initialisation the reference takes from OpenCV when all four vectors are zero (cv::solvePnP, :111-117;
// cv::solvePnPRansac, :437-449) is provided natively, in OpenCV's own shape (SOLVEPNP_ITERATIVE = a direct linear transform
// followed by a Levenberg-Marquardt refinement of the reprojection error): the DLT of the undistorted, normalised image points is
// glue on this side (a 12 x 12 symmetric eigenproblem); the refinement, and for the RANSAC form the refinement and inlier count of
// every sampled hypothesis, run on the device through the same rsba_pnp_tasks with shutter GLOBAL — one launch.  OpenCV's sampling
// order cannot be replayed without OpenCV, so this half is functionally, not bitwise, the reference's.
// Everything numeric that decides a result is computed by librsba_amd; a missing device throws std::runtime_error.
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "session.hpp"

extern "C" {
#include "../rsba_amd.h"
}

namespace rsba_amd {

namespace pnp_detail {

// ceres::AngleAxisRotatePoint for the two rvec/tvec <-> (rotation, camera centre) conversions either side of a call
// (solveRSpnp.cpp:119-128, :166-176): glue on two 3-vectors, not part of the solve.
inline void rotate(const double w[3], const double p[3], double out[3]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double wxp[3] = {w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]};
  if (th2 > 2.220446049250313e-16) {
    const double th = std::sqrt(th2), c = std::cos(th), s = std::sin(th), it = 1.0 / th;
    const double k[3] = {w[0] * it, w[1] * it, w[2] * it};
    const double kxp[3] = {wxp[0] * it, wxp[1] * it, wxp[2] * it};
    const double kp = (k[0] * p[0] + k[1] * p[1] + k[2] * p[2]) * (1.0 - c);
    for (int i = 0; i < 3; ++i) out[i] = p[i] * c + kxp[i] * s + k[i] * kp;
  } else {
    for (int i = 0; i < 3; ++i) out[i] = p[i] + wxp[i];
  }
}
// pose = (rvec, -R(rvec)^T tvec)
inline void to_pose(const double rvec[3], const double tvec[3], double pose[6]) {
  const double rinv[3] = {-rvec[0], -rvec[1], -rvec[2]}, nt[3] = {-tvec[0], -tvec[1], -tvec[2]};
  for (int i = 0; i < 3; ++i) pose[i] = rvec[i];
  rotate(rinv, nt, pose + 3);
}
// tvec = -R(rvec) centre
inline void from_pose(const double pose[6], double rvec[3], double tvec[3]) {
  double t[3];
  rotate(pose, pose + 3, t);
  for (int i = 0; i < 3; ++i) { rvec[i] = pose[i]; tvec[i] = -t[i]; }
}
inline void check(int32_t st) {
  if (st != RSBA_OK) throw std::runtime_error(std::string(rsba_status_string(st)) + ": " + rsba_last_error());
}

// cv::RNG (OpenCV core, restated): multiply-with-carry, uniform(a, b) = a + next() % (b - a)
struct Rng {
  uint64_t state;
  explicit Rng(uint64_t s = 0xffffffffULL) : state(s ? s : 0xffffffffULL) {}
  unsigned next() { state = (uint64_t)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32); return (unsigned)state; }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// ---- global-shutter initialisation (what cv::solvePnP(ITERATIVE) starts from): direct linear transform ----
// sfm intrinsics (mat/cam.h:33): fx, fy, k1, k2, p1, p2, k3, cx, cy
inline void normalised_point(const double cam[NUM_CAM_PARAMS], double u, double v, double out[2]) {
  const double pn[2] = {(u - cam[7]) / cam[0], (v - cam[8]) / cam[1]};
  double pu[2] = {pn[0], pn[1]};
  for (int it = 0; it < 20; ++it) {   // the fixed-point undistortion of mat/cam.h:77-112, a few steps of it
    const double x = pu[0], y = pu[1], r2 = x * x + y * y, d = 1.0 + r2 * (cam[2] + r2 * (cam[3] + r2 * cam[6])), xy = x * y;
    const double dx = d * x + 2.0 * cam[4] * xy + cam[5] * (r2 + 2.0 * x * x), dy = d * y + cam[4] * (r2 + 2.0 * y * y) + 2.0 * cam[5] * xy;
    pu[0] -= dx - pn[0]; pu[1] -= dy - pn[1];
  }
  out[0] = pu[0]; out[1] = pu[1];
}
// eigenvector of the smallest eigenvalue of a symmetric 12 x 12 matrix (cyclic Jacobi)
inline void smallest_eigenvector12(double a[12][12], double vec[12], double gap[3] = nullptr) {
  double v[12][12] = {};
  for (int i = 0; i < 12; ++i) v[i][i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < 12; ++i) for (int j = i + 1; j < 12; ++j) off += a[i][j] * a[i][j];
    if (off < 1e-30) break;
    for (int p_ = 0; p_ < 12; ++p_) for (int q = p_ + 1; q < 12; ++q) {
      if (std::fabs(a[p_][q]) < 1e-300) continue;
      const double theta = (a[q][q] - a[p_][p_]) / (2.0 * a[p_][q]);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0)), c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
      for (int k = 0; k < 12; ++k) { const double akp = a[k][p_], akq = a[k][q]; a[k][p_] = c * akp - sn * akq; a[k][q] = sn * akp + c * akq; }
      for (int k = 0; k < 12; ++k) { const double apk = a[p_][k], aqk = a[q][k]; a[p_][k] = c * apk - sn * aqk; a[q][k] = sn * apk + c * aqk; }
      for (int k = 0; k < 12; ++k) { const double vkp = v[k][p_], vkq = v[k][q]; v[k][p_] = c * vkp - sn * vkq; v[k][q] = sn * vkp + c * vkq; }
    }
  }
  int best = 0;
  for (int i = 1; i < 12; ++i) if (a[i][i] < a[best][best]) best = i;
  for (int k = 0; k < 12; ++k) vec[k] = v[k][best];
  if (gap) {   // the two smallest eigenvalues and the largest: is the null space ONE direction?
    double second = -1.0, largest = 0.0;
    for (int i = 0; i < 12; ++i) { if (i != best && (second < 0.0 || a[i][i] < second)) second = a[i][i]; largest = std::fmax(largest, a[i][i]); }
    gap[0] = a[best][best]; gap[1] = second; gap[2] = largest;
  }
}
// eigen decomposition of a symmetric 3 x 3 matrix (cyclic Jacobi): values ascending, vec[k] = unit eigenvector of val[k]
inline void eigen3(const double sym[6] /* xx xy xz yy yz zz */, double val[3], double vec[3][3]) {
  double a[3][3] = {{sym[0], sym[1], sym[2]}, {sym[1], sym[3], sym[4]}, {sym[2], sym[4], sym[5]}}, v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 40; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p_ = 0; p_ < 3; ++p_) for (int q = p_ + 1; q < 3; ++q) {
      if (std::fabs(a[p_][q]) < 1e-300) continue;
      const double theta = (a[q][q] - a[p_][p_]) / (2.0 * a[p_][q]);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0)), c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
      for (int k = 0; k < 3; ++k) { const double akp = a[k][p_], akq = a[k][q]; a[k][p_] = c * akp - sn * akq; a[k][q] = sn * akp + c * akq; }
      for (int k = 0; k < 3; ++k) { const double apk = a[p_][k], aqk = a[q][k]; a[p_][k] = c * apk - sn * aqk; a[q][k] = sn * apk + c * aqk; }
      for (int k = 0; k < 3; ++k) { const double vkp = v[k][p_], vkq = v[k][q]; v[k][p_] = c * vkp - sn * vkq; v[k][q] = sn * vkp + c * vkq; }
    }
  }
  int order[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j) if (a[order[j]][order[j]] < a[order[i]][order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
  for (int k = 0; k < 3; ++k) { val[k] = a[order[k]][order[k]]; for (int r = 0; r < 3; ++r) vec[k][r] = v[r][order[k]]; }
}
// nearest rotation to a (nearly orthogonal) 3 x 3 matrix, R <- (R + R^-T) / 2; false: singular or a reflection
inline bool nearest_rotation(double R[9]) {
  for (int it = 0; it < 30; ++it) {
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (!(std::fabs(det) > 1e-12)) return false;
    const double it_[9] = {(R[4] * R[8] - R[5] * R[7]) / det, (R[5] * R[6] - R[3] * R[8]) / det, (R[3] * R[7] - R[4] * R[6]) / det,
                           (R[2] * R[7] - R[1] * R[8]) / det, (R[0] * R[8] - R[2] * R[6]) / det, (R[1] * R[6] - R[0] * R[7]) / det,
                           (R[1] * R[5] - R[2] * R[4]) / det, (R[2] * R[3] - R[0] * R[5]) / det, (R[0] * R[4] - R[1] * R[3]) / det};   // R^-T
    for (int k = 0; k < 9; ++k) R[k] = 0.5 * (R[k] + it_[k]);
  }
  return R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]) > 0;
}
// (rotation matrix world->camera, tvec) -> rsba's 6-vector
inline bool pose_from_rt(const double R[9], const double tvec[3], double pose[6]) {
  double rvec[3];
  const double tr = R[0] + R[4] + R[8], cs = std::fmin(1.0, std::fmax(-1.0, 0.5 * (tr - 1.0))), th = std::acos(cs);
  const double ax[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const double sn = 0.5 * std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (sn > 1e-8) for (int k = 0; k < 3; ++k) rvec[k] = ax[k] * (th / (2.0 * sn));
  else if (cs > 0) for (int k = 0; k < 3; ++k) rvec[k] = 0.5 * ax[k];
  else {   // a half turn: the axis from the diagonal
    const double d[3] = {std::sqrt(std::fmax(0.0, 0.5 * (R[0] + 1.0))), std::sqrt(std::fmax(0.0, 0.5 * (R[4] + 1.0))), std::sqrt(std::fmax(0.0, 0.5 * (R[8] + 1.0)))};
    rvec[0] = th * d[0]; rvec[1] = th * d[1] * (R[1] + R[3] >= 0 ? 1.0 : -1.0); rvec[2] = th * d[2] * (R[2] + R[6] >= 0 ? 1.0 : -1.0);
  }
  to_pose(rvec, tvec, pose);
  for (int k = 0; k < 6; ++k) if (!std::isfinite(pose[k])) return false;
  return true;
}
// A PLANAR target (a wall, a checkerboard): the 12-unknown DLT has a two-dimensional null space there, and cv::solvePnP — what the
// reference calls, solveRSpnp.cpp:111-117 — starts from the plane's homography instead (its cvFindExtrinsicCameraParams2 takes that branch
// when the two smallest singular values of the centred points' scatter are in a ratio below 1e-3).  Here: the points in the plane's own
// frame (e1, e2, n = eigenvectors of the scatter, centred at c, divided by `scale`), the homography (x, y, 1) -> normalised image point
// from the 9-unknown DLT (image points Hartley-normalised), then [r1 r2 t] ~ H: r1, r2 the normalised first columns, r3 = r1 x r2,
// t = h3 over the mean of the two column norms, the nearest rotation to [r1 r2 r3] — and back to world coordinates.
inline bool planar_pose(const float* opoints, const double* normalised, const int32_t* idx, int m, const double c[3], double scale, const double axes[3][3] /* e1 e2 n (ascending eigenvalues reversed) */,
                        double pose[6]) {
  if (m < 4) return false;
  double e1[3] = {axes[0][0], axes[0][1], axes[0][2]}, e2[3] = {axes[1][0], axes[1][1], axes[1][2]};
  const double nrm[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};   // right-handed whatever the eigenvectors' signs
  double cu = 0.0, cv = 0.0, su = 0.0;
  for (int i = 0; i < m; ++i) { cu += normalised[2 * (size_t)idx[i]] / m; cv += normalised[2 * (size_t)idx[i] + 1] / m; }
  for (int i = 0; i < m; ++i) su += std::hypot(normalised[2 * (size_t)idx[i]] - cu, normalised[2 * (size_t)idx[i] + 1] - cv) / m;
  if (!(su > 0.0)) return false;
  su /= 1.4142135623730951;
  double ata[12][12] = {};
  for (int i = 0; i < m; ++i) {
    const double d[3] = {(opoints[3 * (size_t)idx[i]] - c[0]) / scale, (opoints[3 * (size_t)idx[i] + 1] - c[1]) / scale, (opoints[3 * (size_t)idx[i] + 2] - c[2]) / scale};
    const double x = d[0] * e1[0] + d[1] * e1[1] + d[2] * e1[2], y = d[0] * e2[0] + d[1] * e2[1] + d[2] * e2[2];
    const double u = (normalised[2 * (size_t)idx[i]] - cu) / su, v = (normalised[2 * (size_t)idx[i] + 1] - cv) / su;
    const double r0[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, -u}, r1[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, -v};
    for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) ata[a][b] += r0[a] * r0[b] + r1[a] * r1[b];
  }
  double big = 0.0;
  for (int a = 0; a < 9; ++a) big += ata[a][a];
  if (!(big > 0.0)) return false;
  for (int a = 9; a < 12; ++a) ata[a][a] = 2.0 * big;   // (the 12 x 12 solver with three idle unknowns far from the null space)
  double hv[12], gap[3];
  smallest_eigenvector12(ata, hv, gap);
  if (!(gap[1] > 1e-9 * gap[2]) || !(gap[0] < 0.05 * gap[1])) return false;   // collinear image or object points: no single homography
  // undo the image normalisation: H = [[su 0 cu] [0 su cv] [0 0 1]] Hn
  double H[9];
  for (int k = 0; k < 3; ++k) { H[k] = su * hv[k] + cu * hv[6 + k]; H[3 + k] = su * hv[3 + k] + cv * hv[6 + k]; H[6 + k] = hv[6 + k]; }
  if (H[8] < 0) for (double& x : H) x = -x;   // the centroid (x = y = 0) in front of the camera
  const double n1 = std::sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]), n2 = std::sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
  if (!(n1 > 1e-300) || !(n2 > 1e-300) || !std::isfinite(n1 + n2)) return false;
  const double r1[3] = {H[0] / n1, H[3] / n1, H[6] / n1}, r2[3] = {H[1] / n2, H[4] / n2, H[7] / n2};
  const double r3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  const double lam = 0.5 * (n1 + n2), tp[3] = {H[2] / lam, H[5] / lam, H[8] / lam};
  double Rp[9] = {r1[0], r2[0], r3[0], r1[1], r2[1], r3[1], r1[2], r2[2], r3[2]};
  if (!nearest_rotation(Rp)) return false;
  // x_cam ~ Rp [e1 e2 n]^T (X - c) / scale + tp
  double R[9], tvec[3];
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) R[3 * r + k] = Rp[3 * r] * e1[k] + Rp[3 * r + 1] * e2[k] + Rp[3 * r + 2] * nrm[k];
  for (int r = 0; r < 3; ++r) tvec[r] = scale * tp[r] - (R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2]);
  return pose_from_rt(R, tvec, pose);
}
// rsba pose (angle-axis world->camera, camera centre) from >= 6 correspondences idx[0..m): false when the points are degenerate
inline bool dlt_pose(const float* opoints, const double* normalised, const int32_t* idx, int m, double pose[6]) {
  if (m < 6) return false;
  double c[3] = {0, 0, 0}, scale = 0.0;
  for (int i = 0; i < m; ++i) for (int k = 0; k < 3; ++k) c[k] += opoints[3 * (size_t)idx[i] + k] / m;
  for (int i = 0; i < m; ++i) { double d2 = 0; for (int k = 0; k < 3; ++k) { const double d = opoints[3 * (size_t)idx[i] + k] - c[k]; d2 += d * d; } scale += std::sqrt(d2) / m; }
  if (!(scale > 0.0)) return false;
  // The shape of the cloud, from the eigenvalues l0 <= l1 <= l2 of the centred points' 3 x 3 scatter: collinear points determine no pose
  // (declined: the caller starts from the zero pose, as the reference does whenever the GS initialisation fails); coplanar points —
  // l0 < 1e-3 l1, cv::solvePnP's own test — go through the plane's homography (planar_pose), everything else through the 12-unknown DLT.
  {
    double sc[6] = {0, 0, 0, 0, 0, 0};   // xx xy xz yy yz zz
    for (int i = 0; i < m; ++i) {
      const double d[3] = {(opoints[3 * (size_t)idx[i]] - c[0]) / scale, (opoints[3 * (size_t)idx[i] + 1] - c[1]) / scale, (opoints[3 * (size_t)idx[i] + 2] - c[2]) / scale};
      sc[0] += d[0] * d[0]; sc[1] += d[0] * d[1]; sc[2] += d[0] * d[2]; sc[3] += d[1] * d[1]; sc[4] += d[1] * d[2]; sc[5] += d[2] * d[2];
    }
    double val[3], vec[3][3];
    eigen3(sc, val, vec);
    if (!(val[1] > 1e-4 * val[2])) return false;   // (a strip a hundred times longer than wide counts as a line)
    if (val[0] < 1e-3 * val[1]) { const double axes[3][3] = {{vec[2][0], vec[2][1], vec[2][2]}, {vec[1][0], vec[1][1], vec[1][2]}, {vec[0][0], vec[0][1], vec[0][2]}}; return planar_pose(opoints, normalised, idx, m, c, scale, axes, pose); }
  }
  double ata[12][12] = {};
  for (int i = 0; i < m; ++i) {
    const double X[4] = {(opoints[3 * (size_t)idx[i]] - c[0]) / scale, (opoints[3 * (size_t)idx[i] + 1] - c[1]) / scale, (opoints[3 * (size_t)idx[i] + 2] - c[2]) / scale, 1.0};
    const double u = normalised[2 * (size_t)idx[i]], v = normalised[2 * (size_t)idx[i] + 1];
    double r0[12] = {}, r1[12] = {};
    for (int k = 0; k < 4; ++k) { r0[k] = X[k]; r0[8 + k] = -u * X[k]; r1[4 + k] = X[k]; r1[8 + k] = -v * X[k]; }
    for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) ata[a][b] += r0[a] * r0[b] + r1[a] * r1[b];
  }
  double pvec[12], gap[3];
  smallest_eigenvector12(ata, pvec, gap);
  if (!(gap[1] > 1e-9 * gap[2]) || !(gap[0] < 0.05 * gap[1])) return false;   // a second (near-)null direction, or no null direction to speak of: an ambiguous DLT is not an initialisation
  double M[9], t[3];
  for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) M[3 * r + k] = pvec[4 * r + k]; t[r] = pvec[4 * r + 3]; }
  // the points must end up in front of the camera (z of the centroid = t[2] in the shifted frame)
  if (t[2] < 0) { for (double& x : M) x = -x; for (double& x : t) x = -x; }
  double lambda = 0.0;
  for (int r = 0; r < 3; ++r) lambda += std::sqrt(M[3 * r] * M[3 * r] + M[3 * r + 1] * M[3 * r + 1] + M[3 * r + 2] * M[3 * r + 2]) / 3.0;
  if (!(lambda > 1e-300) || !std::isfinite(lambda)) return false;
  double R[9];
  for (int k = 0; k < 9; ++k) R[k] = M[k] / lambda;
  if (!nearest_rotation(R)) return false;
  // x_cam ~ R (X - c) / scale + t / lambda  =>  tvec = scale t / lambda - R c (up to the common factor 1 / scale, which the projection ignores)
  double tvec[3];
  for (int r = 0; r < 3; ++r) tvec[r] = scale * t[r] / lambda - (R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2]);
  return pose_from_rt(R, tvec, pose);
}
inline std::vector<double> normalised_points(const double cam[NUM_CAM_PARAMS], const float* ipoints, int n) {
  std::vector<double> out(2 * (size_t)n);
  for (int i = 0; i < n; ++i) normalised_point(cam, ipoints[2 * (size_t)i], ipoints[2 * (size_t)i + 1], &out[2 * (size_t)i]);
  return out;
}

}  // namespace pnp_detail

// cv::solvePnP(..., SOLVEPNP_ITERATIVE) as the reference uses it at solveRSpnp.cpp:111-117: DLT over all points, then a
// Levenberg-Marquardt refinement of the global-shutter reprojection error (on the device).  pose = rsba's 6-vector.
inline bool solveGsPnP(const float* opoints, const float* ipoints, int n, const double cam[NUM_CAM_PARAMS], double pose[6], int device = 0) {
  const std::vector<double> nrm = pnp_detail::normalised_points(cam, ipoints, n);
  std::vector<int32_t> all((size_t)n);
  for (int i = 0; i < n; ++i) all[(size_t)i] = i;
  double init[12], out[12], cost = 0.0; uint8_t status = 0; int32_t inl = 0;
  if (!pnp_detail::dlt_pose(opoints, nrm.data(), all.data(), n, init)) return false;
  for (int k = 0; k < 6; ++k) init[6 + k] = init[k];
  const int32_t sl[2] = {0, 1};
  pnp_detail::check(rsba_pnp_tasks(device, cam, (int32_t)GLOBAL, sl, opoints, ipoints, n, all.data(), n, 1, init, 0, 20, 0, 0.0f, out, &status, &cost, &inl));
  for (int k = 0; k < 6; ++k) pose[k] = status == 1 ? out[k] : init[k];
  return true;
}

// cv::solvePnPRansac as the reference uses it at solveRSpnp.cpp:437-449: minimal subsets, a pose per subset, the pose with the most
// points inside reprojectionError wins.  The subsets' DLT poses are refined and scored on the device in ONE launch.  Returns the
// number of inliers of the winner (0: none found).
inline int solveGsPnPRansac(const float* opoints, const float* ipoints, int n, const double cam[NUM_CAM_PARAMS], double pose[6], int iterationsCount,
                            float reprojectionError, int min_points_count = 6, uint64_t rng_state = 0x9e3779b97f4a7c15ULL, int device = 0) {
  const int m = min_points_count < 6 ? 6 : min_points_count;
  if (n < m || iterationsCount <= 0) return 0;
  const std::vector<double> nrm = pnp_detail::normalised_points(cam, ipoints, n);
  pnp_detail::Rng gen(rng_state);
  std::vector<int32_t> subsets; std::vector<double> inits;
  std::vector<int32_t> pick((size_t)m);
  for (int it = 0; it < iterationsCount; ++it) {
    for (int k = 0; k < m;) {   // m distinct indices
      const int c = gen.uniform(0, n); bool dup = false;
      for (int j = 0; j < k; ++j) dup = dup || pick[(size_t)j] == c;
      if (!dup) pick[(size_t)k++] = c;
    }
    double p[6];
    if (!pnp_detail::dlt_pose(opoints, nrm.data(), pick.data(), m, p)) continue;
    subsets.insert(subsets.end(), pick.begin(), pick.end());
    inits.insert(inits.end(), p, p + 6); inits.insert(inits.end(), p, p + 6);
  }
  const int tasks = (int)(inits.size() / 12);
  if (tasks == 0) return 0;
  std::vector<double> poses((size_t)tasks * 12), cost((size_t)tasks);
  std::vector<uint8_t> status((size_t)tasks);
  std::vector<int32_t> count((size_t)tasks);
  const int32_t sl[2] = {0, 1};
  pnp_detail::check(rsba_pnp_tasks(device, cam, (int32_t)GLOBAL, sl, opoints, ipoints, n, subsets.data(), m, tasks, inits.data(), 12, 5, 0, reprojectionError,
                                   poses.data(), status.data(), cost.data(), count.data()));
  int best = -1;
  for (int t = 0; t < tasks; ++t) if (status[(size_t)t] != 0 && (best < 0 || count[(size_t)t] > count[(size_t)best])) best = t;
  if (best < 0) return 0;
  for (int k = 0; k < 6; ++k) pose[k] = poses[(size_t)best * 12 + k];
  return count[(size_t)best];
}

// solveRSpnp.cpp:100-192.  Returns summary.IsSolutionUsable(); the four vectors are updated only then.
inline bool solveRsPnP(const float* opoints, const float* ipoints, int n, const double cam[NUM_CAM_PARAMS], double rvec[3], double tvec[3],
                       double rvec2[3], double tvec2[3], const SHUTTER shutter, const int scanlines[2], int device = 0) {
  double l1 = 0.0;
  for (int i = 0; i < 3; ++i) l1 += std::fabs(rvec[i]) + std::fabs(tvec[i]) + std::fabs(rvec2[i]) + std::fabs(tvec2[i]);
  double init[12], out[12], cost = 0.0; uint8_t status = 0; int32_t inl = 0;
  pnp_detail::to_pose(rvec, tvec, init);
  pnp_detail::to_pose(rvec2, tvec2, init + 6);
  if (l1 == 0.0 && solveGsPnP(opoints, ipoints, n, cam, init, device))   // GS Init (:111-117): both poses start from the global-shutter one
    for (int k = 0; k < 6; ++k) init[6 + k] = init[k];
  std::vector<int32_t> all((size_t)n);
  for (int i = 0; i < n; ++i) all[(size_t)i] = i;
  const int32_t sl[2] = {scanlines[0], scanlines[1]};
  pnp_detail::check(rsba_pnp_tasks(device, cam, (int32_t)shutter, sl, opoints, ipoints, n, all.data(), n, 1, init, 0, 10, 0, 0.0f, out, &status, &cost, &inl));
  if (status != 1) return false;
  pnp_detail::from_pose(out, rvec, tvec);
  pnp_detail::from_pose(out + 6, rvec2, tvec2);
  return true;
}

// solveRSpnp.cpp:413-524.  inliers (may be null) receives the indices of the winning hypothesis' inliers.
inline void solveRsPnPRansac(const float* opoints, const float* ipoints, int n, const double cam[NUM_CAM_PARAMS], double rvec[3], double tvec[3],
                             double rvec2[3], double tvec2[3], const SHUTTER shutter, const int scanlines[2], int iterationsCount = 100,
                             float reprojectionError = 8.0f, int minInliersCount = 100, std::vector<int>* inliers = nullptr,
                             int min_points_count = 6, uint64_t rng_state = 0xffffffffULL, int device = 0) {
  double l1 = 0.0;
  for (int i = 0; i < 3; ++i) l1 += std::fabs(rvec[i]) + std::fabs(tvec[i]) + std::fabs(rvec2[i]) + std::fabs(tvec2[i]);
  double init[12];
  pnp_detail::to_pose(rvec, tvec, init);
  pnp_detail::to_pose(rvec2, tvec2, init + 6);
  if (l1 == 0.0) {   // GS Init (:437-449): global-shutter RANSAC at twice the threshold, taken when it finds more than four inliers
    double gs[6];
    if (solveGsPnPRansac(opoints, ipoints, n, cam, gs, iterationsCount, reprojectionError * 2, min_points_count, rng_state ^ 0x9e3779b97f4a7c15ULL, device) > 4)
      for (int k = 0; k < 6; ++k) init[k] = init[6 + k] = gs[k];
  }
  if (minInliersCount <= 0) minInliersCount = n;                                        // :453-454
  const int32_t sl[2] = {scanlines[0], scanlines[1]};
  double best_pose[12];
  for (int k = 0; k < 12; ++k) best_pose[k] = init[k];
  int best_count = 0; bool have_best = false;
  if (n >= min_points_count && iterationsCount > 0) {                                   // :473-478
    // the subsets PnPSolver::operator() draws: one mask, shuffled cumulatively by generateVar (:334-337, :381-391)
    const int m = min_points_count;
    std::vector<char> mask((size_t)n, 0);
    for (int i = 0; i < m; ++i) mask[(size_t)i] = 1;
    pnp_detail::Rng gen(rng_state);
    std::vector<int32_t> subsets((size_t)iterationsCount * m);
    for (int it = 0; it < iterationsCount; ++it) {
      for (int i = 0; i < n; ++i) { const int i1 = gen.uniform(0, n), i2 = gen.uniform(0, n); const char c = mask[(size_t)i1]; mask[(size_t)i1] = mask[(size_t)i2]; mask[(size_t)i2] = c; }
      int col = 0;
      for (int i = 0; i < n; ++i) if (mask[(size_t)i]) subsets[(size_t)it * m + col++] = i;
    }
    std::vector<double> poses((size_t)iterationsCount * 12), cost((size_t)iterationsCount);
    std::vector<uint8_t> status((size_t)iterationsCount);
    std::vector<int32_t> count((size_t)iterationsCount);
    pnp_detail::check(rsba_pnp_tasks(device, cam, (int32_t)shutter, sl, opoints, ipoints, n, subsets.data(), m, iterationsCount, init, 0, 10, 1,
                                     reprojectionError, poses.data(), status.data(), cost.data(), count.data()));
    for (int it = 0; it < iterationsCount; ++it) {                                      // pnpTask's bookkeeping, in order (:312-326, :339-346)
      if (status[(size_t)it] != 0 && count[(size_t)it] > best_count) {
        best_count = count[(size_t)it]; have_best = true;
        for (int k = 0; k < 12; ++k) best_pose[k] = poses[(size_t)it * 12 + k];
      }
      if (best_count >= minInliersCount) break;
    }
  }
  if (have_best && best_count >= min_points_count) {                                    // :480-512
    std::vector<uint8_t> flags((size_t)n);
    pnp_detail::check(rsba_pnp_inliers(device, cam, (int32_t)shutter, sl, opoints, ipoints, n, best_pose, reprojectionError, flags.data()));
    std::vector<int32_t> idx;
    for (int i = 0; i < n; ++i) if (flags[(size_t)i]) idx.push_back(i);
    // final solveRsPnP over the inliers from the winning poses (:496-502)
    double refined[12], c = 0.0; uint8_t st = 0; int32_t cnt = 0;
    pnp_detail::check(rsba_pnp_tasks(device, cam, (int32_t)shutter, sl, opoints, ipoints, n, idx.data(), (int32_t)idx.size(), 1, best_pose, 0, 10, 0,
                                     reprojectionError, refined, &st, &c, &cnt));
    const double* fin = st == 1 ? refined : best_pose;
    pnp_detail::from_pose(fin, rvec, tvec);
    pnp_detail::from_pose(fin + 6, rvec2, tvec2);
    if (inliers) inliers->assign(idx.begin(), idx.end());
  } else {                                                                              // :513-521
    for (int i = 0; i < 3; ++i) { rvec[i] = tvec[i] = rvec2[i] = tvec2[i] = 0.0; }
    if (inliers) inliers->clear();
  }
}

}  // namespace rsba_amd
