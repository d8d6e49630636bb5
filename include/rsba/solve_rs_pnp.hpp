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
// Not provided: the global-shutter initialisation through cv::solvePnP / cv::solvePnPRansac when all four vectors
// are zero (:111-117, :437-449) — the caller passes an initial guess (std::invalid_argument otherwise).
// Everything numeric is computed by librsba_amd; a missing device throws std::runtime_error.
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

}  // namespace pnp_detail

// solveRSpnp.cpp:100-192.  Returns summary.IsSolutionUsable(); the four vectors are updated only then.
inline bool solveRsPnP(const float* opoints, const float* ipoints, int n, const double cam[NUM_CAM_PARAMS], double rvec[3], double tvec[3],
                       double rvec2[3], double tvec2[3], const SHUTTER shutter, const int scanlines[2], int device = 0) {
  double l1 = 0.0;
  for (int i = 0; i < 3; ++i) l1 += std::fabs(rvec[i]) + std::fabs(tvec[i]) + std::fabs(rvec2[i]) + std::fabs(tvec2[i]);
  if (l1 == 0.0) throw std::invalid_argument("solveRsPnP: the global-shutter initialisation (cv::solvePnP, solveRSpnp.cpp:111-117) is not provided: pass an initial guess");
  double init[12], out[12], cost = 0.0; uint8_t status = 0; int32_t inl = 0;
  pnp_detail::to_pose(rvec, tvec, init);
  pnp_detail::to_pose(rvec2, tvec2, init + 6);
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
  if (l1 == 0.0) throw std::invalid_argument("solveRsPnPRansac: the global-shutter initialisation (cv::solvePnPRansac, solveRSpnp.cpp:437-449) is not provided: pass an initial guess");
  if (minInliersCount <= 0) minInliersCount = n;                                        // :453-454
  const int32_t sl[2] = {scanlines[0], scanlines[1]};
  double init[12];
  pnp_detail::to_pose(rvec, tvec, init);
  pnp_detail::to_pose(rvec2, tvec2, init + 6);
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
