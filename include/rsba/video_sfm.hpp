// Host-side mirror of the reprojection / validation helpers either side of the solve:
//   vision::sfm::reproject(sess, f, opt, pt, obs)   /root/reference/src/rsba/struct/VideoSfM.cc:139-155
//   vision::sfm::validate (sess, f, opt, pt, obs)   /root/reference/src/rsba/struct/VideoSfM.cc:159-169
// Same names and argument meaning; the batched overloads are what a caller that loops over a frame's
// observations (CeresHandler.h:239-243, VideoSfMHandler.cc evalTracks) should use — one device launch per
// frame instead of one call per observation.  Everything is computed by librsba_amd (rsba_validate_frame /
// rsba_reproject_frame: no handle, a per-thread device arena is reused between calls); a missing device throws
// std::runtime_error.
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "session.hpp"

extern "C" {
#include "../rsba_amd.h"
}

namespace rsba_amd {

namespace detail {

// the arguments of one frame, flattened for rsba_validate_frame / rsba_reproject_frame
struct FrameArgs {
  std::vector<double> poses, points, xy;
  const double* cam;
  int32_t scan[2];
  FrameArgs(const Session& sess, const Frame& f, const std::vector<const double*>& pts, const std::vector<std::array<double, 2>>& obs) {
    if (f.poses.empty()) throw std::runtime_error("empty frame");   // struct/VideoSfM.cc:105 (more than two poses: one per scan line, picked per item on the device as getPose does, :118-132)
    for (const auto& p : f.poses) poses.insert(poses.end(), p.begin(), p.end());
    cam = (f.__isset.cam ? f.cam : sess.cam).data();
    scan[0] = sess.scanlines[0]; scan[1] = sess.scanlines[1];
    const size_t n = pts.size();
    points.resize(n * 3); xy.resize(obs.size() * 2);
    for (size_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) points[3 * i + k] = pts[i][k];
    for (size_t i = 0; i < obs.size(); ++i) { xy[2 * i] = obs[i][0]; xy[2 * i + 1] = obs[i][1]; }
  }
};

}  // namespace detail

// batched validate: flags[i] = validate(sess, f, opt, pts[i], obs[i]); one upload, one launch, one download per call
inline std::vector<uint8_t> validate(const Session& sess, const Frame& f, const SfmOptions& opt, const std::vector<const double*>& pts,
                                     const std::vector<std::array<double, 2>>& obs, int device = 0) {
  std::vector<uint8_t> flags(pts.size(), 0);
  if (pts.empty()) return flags;
  if (obs.size() != pts.size()) throw std::runtime_error("validate: one observation per point");
  const detail::FrameArgs a(sess, f, pts, obs);
  if (rsba_validate_frame(device, a.cam, a.poses.data(), (int32_t)f.poses.size(), sess.rs, a.scan, opt.model.interpolateRotation, a.points.data(), a.xy.data(),
                          (int64_t)pts.size(), opt.tracks.sqrdThreshold, (double)opt.tracks.minDistanceToCamera, flags.data()) != RSBA_OK)
    throw std::runtime_error(std::string("rsba_amd: ") + rsba_last_error());
  return flags;
}

// struct/VideoSfM.cc:159-169
inline bool validate(const Session& sess, const Frame& f, const SfmOptions& opt, const double pt[3], const double obs[2]) {
  return validate(sess, f, opt, std::vector<const double*>{pt}, std::vector<std::array<double, 2>>{{obs[0], obs[1]}})[0] != 0;
}

// batched reproject: obs[i] <- projection of pts[i] into f at its own scan-line time; flags[i] = return value
inline std::vector<uint8_t> reproject(const Session& sess, const Frame& f, const SfmOptions& opt, const std::vector<const double*>& pts,
                                      std::vector<std::array<double, 2>>& obs, int device = 0) {
  std::vector<uint8_t> flags(pts.size(), 0);
  obs.resize(pts.size());
  if (pts.empty()) return flags;
  const detail::FrameArgs a(sess, f, pts, {});
  if (rsba_reproject_frame(device, a.cam, a.poses.data(), (int32_t)f.poses.size(), sess.rs, a.scan, opt.model.interpolateRotation, a.points.data(),
                           (int64_t)pts.size(), &obs[0][0], flags.data()) != RSBA_OK)
    throw std::runtime_error(std::string("rsba_amd: ") + rsba_last_error());
  return flags;
}

// struct/VideoSfM.cc:139-155
inline bool reproject(const Session& sess, const Frame& f, const SfmOptions& opt, const double pt[3], double obs[2]) {
  std::vector<std::array<double, 2>> o;
  const bool ok = reproject(sess, f, opt, std::vector<const double*>{pt}, o)[0] != 0;
  obs[0] = o[0][0]; obs[1] = o[0][1];
  return ok;
}

}  // namespace rsba_amd
