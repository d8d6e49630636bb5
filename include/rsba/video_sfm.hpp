// Host-side mirror of the reprojection / validation helpers either side of the solve:
//   vision::sfm::reproject(sess, f, opt, pt, obs)   /root/reference/src/rsba/struct/VideoSfM.cc:139-155
//   vision::sfm::validate (sess, f, opt, pt, obs)   /root/reference/src/rsba/struct/VideoSfM.cc:159-169
// Same names and argument meaning; the batched overloads are what a caller that loops over a frame's
// observations (CeresHandler.h:239-243, VideoSfMHandler.cc evalTracks) should use — one device launch per
// frame instead of one call per observation.  Everything is computed by librsba_amd (rsba_validate_observations
// / rsba_reproject); a missing device throws std::runtime_error.
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

// one-frame device problem over n (point, observation) pairs
class FrameProblem {
 public:
  FrameProblem(const Session& sess, const Frame& f, const SfmOptions& opt, const std::vector<const double*>& pts,
               const std::vector<std::array<double, 2>>& obs) {
    const size_t n = pts.size();
    if (f.poses.empty() || f.poses.size() > 2) throw std::runtime_error("validate/reproject: frames with 1 or 2 poses only");
    for (const auto& p : f.poses) poses_.insert(poses_.end(), p.begin(), p.end());
    const std::vector<double>& cam = f.__isset.cam ? f.cam : sess.cam;
    cam_.assign(cam.begin(), cam.end());
    points_.resize(n * 3); xy_.resize(n * 2); frame_.assign(n, 0); point_.resize(n);
    for (size_t i = 0; i < n; ++i) {
      for (int k = 0; k < 3; ++k) points_[3 * i + k] = pts[i][k];
      if (!obs.empty()) { xy_[2 * i] = obs[i][0]; xy_[2 * i + 1] = obs[i][1]; }
      point_[i] = (int32_t)i;
    }
    rsba_problem_desc d = {};
    d.shutter = sess.rs; d.scanlines[0] = sess.scanlines[0]; d.scanlines[1] = sess.scanlines[1];
    d.interpolate_rotation = opt.model.interpolateRotation; d.calibrated = 1;
    d.poses_per_frame = (int32_t)f.poses.size(); d.num_frames = 1; d.num_points = (int32_t)n; d.num_intrinsics = 1;
    d.num_observations = (int64_t)n;
    d.poses = poses_.data(); d.points = points_.data(); d.intrinsics = cam_.data();
    d.obs_xy = xy_.data(); d.obs_frame = frame_.data(); d.obs_point = point_.data();
    if (rsba_create(&d, 0, &h_) != RSBA_OK) throw std::runtime_error(std::string("rsba_amd: ") + rsba_last_error());
  }
  ~FrameProblem() { if (h_) rsba_destroy(h_); }
  FrameProblem(const FrameProblem&) = delete;
  FrameProblem& operator=(const FrameProblem&) = delete;
  rsba_handle* handle() const { return h_; }
  const std::vector<int32_t>& frames() const { return frame_; }
  const std::vector<int32_t>& points() const { return point_; }

 private:
  std::vector<double> poses_, cam_, points_, xy_;
  std::vector<int32_t> frame_, point_;
  rsba_handle* h_ = nullptr;
};

}  // namespace detail

// batched validate: flags[i] = validate(sess, f, opt, pts[i], obs[i])
inline std::vector<uint8_t> validate(const Session& sess, const Frame& f, const SfmOptions& opt, const std::vector<const double*>& pts,
                                     const std::vector<std::array<double, 2>>& obs) {
  std::vector<uint8_t> flags(pts.size(), 0);
  if (pts.empty()) return flags;
  detail::FrameProblem fp(sess, f, opt, pts, obs);
  if (rsba_validate_observations(fp.handle(), opt.tracks.sqrdThreshold, (double)opt.tracks.minDistanceToCamera, flags.data()) != RSBA_OK)
    throw std::runtime_error(std::string("rsba_amd: ") + rsba_last_error());
  return flags;
}

// struct/VideoSfM.cc:159-169
inline bool validate(const Session& sess, const Frame& f, const SfmOptions& opt, const double pt[3], const double obs[2]) {
  return validate(sess, f, opt, std::vector<const double*>{pt}, std::vector<std::array<double, 2>>{{obs[0], obs[1]}})[0] != 0;
}

// batched reproject: obs[i] <- projection of pts[i] into f at its own scan-line time; flags[i] = return value
inline std::vector<uint8_t> reproject(const Session& sess, const Frame& f, const SfmOptions& opt, const std::vector<const double*>& pts,
                                      std::vector<std::array<double, 2>>& obs) {
  std::vector<uint8_t> flags(pts.size(), 0);
  obs.resize(pts.size());
  if (pts.empty()) return flags;
  detail::FrameProblem fp(sess, f, opt, pts, {});
  if (rsba_reproject(fp.handle(), fp.frames().data(), fp.points().data(), (int64_t)pts.size(), &obs[0][0], flags.data()) != RSBA_OK)
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
