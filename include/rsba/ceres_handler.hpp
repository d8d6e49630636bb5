// Host-side mirror of rsba's problem assembly and BA entry point, over the facade:
//   CeresHandler::CeresHandler / Add / solve   /root/reference/src/rsba/CeresHandler.h:76-427
//   VideoSfMHandler::BA                        /root/reference/src/rsba/VideoSfMHandler.cc:574-631
// Same names, argument meaning and error behaviour for the per-observation path (SURVEY Appendix D
// steps 1, 3-4: pose initialisation of frames without poses, the observation loop incl. the match-based track lookup).
// Not built (out-of-scope costs): the structure-less ray costs — reaching them throws std::runtime_error.  The per-pose priors
// (SphericalPrior :127-130, GoodPosePrior :188-204) and the motion priors (:147-186) are built, the latter with a known
// opt.ceres.interFrameRatio (!= 1: constant block) and with the free, lower-bounded ratio of the default.  opt.debug.calcCovariances (VideoSfMHandler.cc:599-621) is built.  revalidateReprojections (:239-243) runs as one
// batched device validation per frame (video_sfm.hpp).
#pragma once
#include <cmath>
#include <iomanip>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <thread>

#include "motion_priors.hpp"
#include "pose_priors.hpp"
#include "reprojection_costs.hpp"
#include "video_sfm.hpp"

namespace rsba_amd {

// struct/VideoSfM.cc:75-99 getPose (pointer-returning overload), 1-pose and "one pose per scan-line" cases
inline double* getPose(const Session& sess, Frame& f, const SfmOptions&, const double obs[2]) {
  switch (f.poses.size()) {
    case 0: throw std::runtime_error("empty frame");
    case 1: return f.poses[0].data();
    case 2: throw std::runtime_error("not possible to get a pose reference on linear RS");
    default: {
      double line = (sess.rs == HORIZONTAL) ? obs[0] : obs[1];
      if (line < 0) line = 0; else if (line > f.poses.size() - 1) line = (double)(f.poses.size() - 1);
      return f.poses[(size_t)std::round(line)].data();
    }
  }
}

class CeresHandler {
 public:
  ceres::Problem problem;
  SfmOptions opt;
  ceres::LossFunction* lossFunction = nullptr;
  size_t startFrame;

  // CeresHandler.h:85-90: one shared HuberLoss when opt.ceres.huberLoss > 0
  CeresHandler(const SfmOptions& o, size_t start = 0) : problem(), opt(o), startFrame(start) {
    if (opt.ceres.huberLoss > 0) lossFunction = new ceres::HuberLoss(opt.ceres.huberLoss);
  }

  // CeresHandler.h:94-390, per-observation path
  void Add(const size_t frameKey, Session& sess, bool uninitialized = false) {
    Frame& f = sess.frames[frameKey];
    const int formerParamNum = problem.NumParameterBlocks();
    if (!f.__isset.poses) {                                            // :99-144 initialise the camera frame
      if (frameKey > 0) {
        Frame& f_1 = sess.frames[frameKey - 1];
        f.poses = f_1.poses;                                            // last pose as reference
        if (frameKey > 1) {                                             // extrapolate the linear velocity
          Frame& f_2 = sess.frames[frameKey - 2];
          for (size_t pi = 0; pi < f.poses.size(); ++pi)
            for (int k = 0; k < NUM_POSE_PARAMS; ++k) f.poses[pi][k] = f_1.poses[pi][k] + (f_1.poses[pi][k] - f_2.poses[pi][k]);
        } else {
          bool originFrame = true;
          for (auto& pose : f_1.poses) if (originFrame) for (double p : pose) if (p != 0) originFrame = false;
          for (auto& pose : f.poses) { pose[3] += 1e-4; pose[4] += 1e-4; pose[5] += 1e-4; }
          if (frameKey == 1 && originFrame)                             // :127-130
            problem.AddResidualBlock(SphericalPrior::Create(), nullptr, f.poses[0].data());
        }
      } else {                                                          // frameKey == 0: zeros
        f.poses.assign(opt.model.rolling_shutter ? 2 : 1, std::vector<double>(NUM_POSE_PARAMS, 0.0));
      }
      uninitialized = true;
      f.__isset.poses = true;
    }
    if (frameKey >= opt.ceres.fixFirstNCameras) {
      if (frameKey > 0 && (opt.ceres.constFrameVelocity != 0 || opt.ceres.constFrameAcceleration != 0)) {   // :148-186
        Frame& f_1 = sess.frames[frameKey - 1];
        if (f.poses.size() == 2 && f_1.poses.size() == 2) {
          if (opt.ceres.constFrameAcceleration != 0) {
            problem.AddResidualBlock(RsConstAccelerationPrior::Create(opt.ceres.constFrameAcceleration), lossFunction, &opt.ceres.interFrameRatio,
                                     f.poses[0].data(), f.poses[1].data(), f_1.poses[0].data(), f_1.poses[1].data());
            problem.SetParameterLowerBound(&opt.ceres.interFrameRatio, 0, std::numeric_limits<double>::epsilon());   // _EPS
          } else {
            problem.AddResidualBlock(RsConstVeloPrior::Create(opt.ceres.constFrameVelocity), lossFunction, &opt.ceres.interFrameRatio,
                                     f.poses[0].data(), f.poses[1].data(), f_1.poses[0].data(), f_1.poses[1].data());
            problem.SetParameterLowerBound(&opt.ceres.interFrameRatio, 0, 0.0);
          }
          // :175-177.  With the option left at 1 the ratio stays a free, lower-bounded parameter block that Solve
          // optimises (and writes back into opt.ceres.interFrameRatio), as in the reference.
          if (opt.ceres.interFrameRatio != 1) problem.SetParameterBlockConstant(&opt.ceres.interFrameRatio);
          if (frameKey - 1 < opt.ceres.fixFirstNCameras) {   // :179-184 the previous frame is one of the fixed cameras
            for (auto& pose : f_1.poses) problem.SetParameterBlockConstant(pose.data());
            if (f_1.__isset.cam) problem.SetParameterBlockConstant(f_1.cam.data());
          }
        }
      }
      if ((opt.ceres.trustPriorCamRotation != 0 || opt.ceres.trustPriorCamPosition != 0) && f.__isset.priorPoses && !f.priorPoses.empty()) {   // :188-204
        if (!f.__isset.poses || f.poses.size() != f.priorPoses.size()) f.poses = f.priorPoses;
        for (size_t i = 0; i < f.poses.size(); i++)
          problem.AddResidualBlock(GoodPosePrior::Create(opt.ceres.trustPriorCamRotation, opt.ceres.trustPriorCamPosition), nullptr,
                                   f.priorPoses[i].data(), f.poses[i].data());
      }
    }
    if (!opt.model.use3Dpoints) throw std::runtime_error("structure-less costs (CeresHandler.h:303-332) are not built");
    std::vector<Track*> track_of(f.obs.size(), nullptr);
    // :220-236 "also add bad reprojections": an observation without a usable track takes the track of the first of its
    // matches whose point validates against this frame.  The candidates of the whole frame go through ONE batched
    // device validation; the first valid one per observation is what the reference's sequential loop finds.
    std::vector<size_t> cand_obs; std::vector<Track*> cand_track;
    for (size_t oi = 0; oi < f.obs.size(); ++oi) {
      Observation& o = f.obs[oi];
      Track* t = nullptr;
      if (o.__isset.track) {                                           // :215-218
        t = &sess.getTrack((size_t)o.track);
        if (!t->__isset.pt || (opt.ceres.useOnlyValidMatches && !t->valid)) t = nullptr;
      }
      if (!t && (uninitialized || !opt.ceres.useOnlyValidMatches)) {
        for (const ObservationRef& ref : o.matches) {
          const Observation& o2 = sess.frames[(size_t)ref.frame].obs[(size_t)ref.obs];
          if (!o2.__isset.track) continue;
          Track* c = &sess.getTrack((size_t)o2.track);
          if (!c->__isset.pt || (opt.ceres.useOnlyValidMatches && !c->valid)) continue;
          cand_obs.push_back(oi); cand_track.push_back(c);
        }
        continue;
      }
      if (t && (t->valid || !opt.ceres.useOnlyValidMatches)) track_of[oi] = t;   // :238
    }
    if (!cand_obs.empty()) {
      std::vector<const double*> pts; std::vector<std::array<double, 2>> xy;
      for (size_t k = 0; k < cand_obs.size(); ++k) { pts.push_back(cand_track[k]->pt.data()); xy.push_back({f.obs[cand_obs[k]].x, f.obs[cand_obs[k]].y}); }
      const std::vector<uint8_t> ok = validate(sess, f, opt, pts, xy);
      for (size_t k = 0; k < cand_obs.size(); ++k) {
        Track* c = cand_track[k];
        if (ok[k] && !track_of[cand_obs[k]] && (c->valid || !opt.ceres.useOnlyValidMatches)) track_of[cand_obs[k]] = c;   // first found wins
      }
    }
    if (opt.ceres.revalidateReprojections) {                           // :239-243, all observations of the frame in one launch
      std::vector<const double*> pts; std::vector<std::array<double, 2>> xy; std::vector<size_t> which;
      for (size_t oi = 0; oi < f.obs.size(); ++oi)
        if (track_of[oi]) { pts.push_back(track_of[oi]->pt.data()); xy.push_back({f.obs[oi].x, f.obs[oi].y}); which.push_back(oi); }
      const std::vector<uint8_t> ok = validate(sess, f, opt, pts, xy);
      for (size_t k = 0; k < which.size(); ++k) if (!ok[k]) track_of[which[k]] = nullptr;   // skip observation
    }
    for (size_t oi = 0; oi < f.obs.size(); ++oi) {
      Observation& o = f.obs[oi];
      double obs[2] = {o.x, o.y};
      Track* t = track_of[oi];
      if (!t) continue;
      if (f.poses.size() == 2) {                                       // :245-265 rolling shutter, two poses
        if (opt.model.constVelocity) throw std::runtime_error("constVelocity");   // the reference aborts (:246-247)
        if (opt.model.calibrated) {
          problem.AddResidualBlock(RsBundleAdjustment::Create(sess, opt, obs), lossFunction, f.poses[0].data(), f.poses[1].data(), t->pt.data());
        } else {
          problem.AddResidualBlock(RsBundleAdjustment::CreateWithCam(sess, opt, obs), lossFunction,
                                   f.__isset.cam ? f.cam.data() : sess.cam.data(), f.poses[0].data(), f.poses[1].data(), t->pt.data());
        }
      } else {                                                         // :266-286 one pose per observation
        if (opt.model.calibrated) {
          problem.AddResidualBlock(ReprojectionError::Create(f.__isset.cam ? f.cam.data() : sess.cam.data(), obs), lossFunction,
                                   getPose(sess, f, opt, obs), t->pt.data());
        } else {
          problem.AddResidualBlock(ReprojectionError::Create(obs), lossFunction, f.__isset.cam ? f.cam.data() : sess.cam.data(),
                                   getPose(sess, f, opt, obs), t->pt.data());
        }
        if (frameKey < opt.ceres.fixFirstNCameras) {
          problem.SetParameterBlockConstant(getPose(sess, f, opt, obs));
          if (!opt.model.calibrated && f.__isset.cam) problem.SetParameterBlockConstant(f.cam.data());
        }
      }
      bool fixedOldTrack = false;                                      // :288-300 window BA freezes old tracks
      if (startFrame > 0)
        for (const ObservationRef& ref : t->obs) if ((size_t)ref.frame < startFrame) { fixedOldTrack = true; break; }
      if (fixedOldTrack || opt.ceres.const3d) problem.SetParameterBlockConstant(t->pt.data());
    }
    if (problem.NumParameterBlocks() > formerParamNum) {               // :335-382, first matching rule wins
      if (frameKey < opt.ceres.fixFirstNCameras) {
        if (f.poses.size() == 2) { problem.SetParameterBlockConstant(f.poses[0].data()); problem.SetParameterBlockConstant(f.poses[1].data()); }
        if (!opt.model.calibrated && f.__isset.cam) problem.SetParameterBlockConstant(f.cam.data());
      } else if (opt.ceres.fixScale && (frameKey == 0 || frameKey == sess.frames.size() - 1)) {
        ceres::LocalParameterization* p = new ceres::SubsetParameterization(NUM_POSE_PARAMS, std::vector<int>{3, 4, 5});
        if (frameKey == 0) problem.SetParameterization(f.poses[0].data(), p); else problem.SetParameterization(f.poses.back().data(), p);
      } else if (opt.ceres.fixRotation) {
        for (auto& pose : f.poses) problem.SetParameterization(pose.data(), new ceres::SubsetParameterization(NUM_POSE_PARAMS, std::vector<int>{0, 1, 2}));
      } else if (opt.ceres.fixPosition) {
        for (auto& pose : f.poses) problem.SetParameterization(pose.data(), new ceres::SubsetParameterization(NUM_POSE_PARAMS, std::vector<int>{3, 4, 5}));
      }
    }
  }

  // CeresHandler.h:394-426
  ceres::Solver::Summary solve(ceres::Solver::Options* options = nullptr) {
    ceres::Solver::Options tmp;
    if (options == nullptr) {
      options = &tmp;
      options->linear_solver_type = ceres::SPARSE_SCHUR;
      options->minimizer_progress_to_stdout = true;
      options->max_num_iterations = 50;
    }
#ifdef NDEBUG   // :408-415: all hardware threads in Release builds only (the device path ignores host threads either way)
    const unsigned n = std::thread::hardware_concurrency();
    if (n > 0) options->num_linear_solver_threads = options->num_threads = (int)n;
#endif
    ceres::Solver::Summary summary;
    ceres::Solve(*options, &problem, &summary);
    if (opt.ceres.constFrameVelocity != 0 || opt.ceres.constFrameAcceleration != 0)   // :421-423
      std::cout << "interFrameRatio: " << opt.ceres.interFrameRatio << std::endl;
    return summary;
  }
};

// VideoSfMHandler::BA (VideoSfMHandler.cc:574-631) without the service bookkeeping: options as :579-583,
// Add frames [startFrame, endFrame], solve, print the report and the "average reprojection error"
// sqrt(final_cost / num_residual_blocks_reduced) (:627-628), return IsSolutionUsable() (:630).
inline bool BA(Session& sess, const int32_t startFrame, const int32_t endFrame, const SfmOptions& opt, const int32_t maxIter,
               ceres::Solver::Summary* out = nullptr, bool progress = true, std::vector<std::vector<double>>* covariances = nullptr) {
  ceres::Solver::Options cOpt;
  cOpt.linear_solver_type = ceres::SPARSE_SCHUR;
  cOpt.minimizer_progress_to_stdout = progress;
  cOpt.max_num_iterations = maxIter;
  cOpt.min_linear_solver_iterations = 3;
  CeresHandler cs(opt, (size_t)startFrame);
  for (int32_t fi = startFrame; fi <= endFrame; fi++) cs.Add((size_t)fi, sess);
  ceres::Solver::Summary summary = cs.solve(&cOpt);
  if (progress) std::cout << summary.FullReport() << std::endl;
  if (!summary.IsSolutionUsable()) std::cerr << summary.message << std::endl;
  if (opt.debug.calcCovariances) {                                   // :599-621 (rolling-shutter frames: two poses)
    for (int32_t fi = startFrame; fi <= endFrame; fi++) {
      const Frame& f = sess.frames[(size_t)fi];
      if (f.poses.size() != 2) continue;
      ceres::Covariance::Options options;
      ceres::Covariance covariance(options);
      std::vector<std::pair<const double*, const double*>> covariance_blocks;
      covariance_blocks.push_back(std::make_pair(f.poses[0].data(), f.poses[0].data()));
      covariance_blocks.push_back(std::make_pair(f.poses[0].data(), f.poses[1].data()));
      covariance_blocks.push_back(std::make_pair(f.poses[1].data(), f.poses[1].data()));
      if (covariance.Compute(covariance_blocks, &cs.problem)) {
        std::vector<double> ppe(3 * 36);
        covariance.GetCovarianceBlock(f.poses[0].data(), f.poses[0].data(), &ppe[0]);
        covariance.GetCovarianceBlock(f.poses[0].data(), f.poses[1].data(), &ppe[36]);
        covariance.GetCovarianceBlock(f.poses[1].data(), f.poses[1].data(), &ppe[72]);
        if (progress) {
          const char* names[3] = {"pp:", "pe:", "ee:"};
          for (int b = 0; b < 3; ++b) {
            std::cout << names[b] << std::endl;
            for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) std::cout << ppe[(size_t)b * 36 + r * 6 + c] << (c < 5 ? " " : ""); std::cout << std::endl; }
          }
        }
        if (covariances) { covariances->resize(sess.frames.size()); (*covariances)[(size_t)fi] = ppe; }
      }
    }
  }
  if (progress)
    std::cout << "average reprojection error: " << std::sqrt(summary.final_cost / summary.num_residual_blocks_reduced) << std::endl;
  if (out) *out = summary;
  return summary.IsSolutionUsable();
}

namespace detail {
// VideoSfMHandler.cc:163-172 = :197-206: with fewer than 100 valid tracks in the whole session the valid-only filter is dropped
inline void relax_valid_matches_rule(const Session& sess, SfmOptions& opt) {
  if (!opt.ceres.useOnlyValidMatches) return;
  size_t ntracks = 0;
  for (const Track& t : sess.tracks) ntracks += t.valid ? 1 : 0;
  if (ntracks >= 100) return;
  opt.ceres.useOnlyValidMatches = false;
  std::cout << "!!! Not enough valid matches: " << ntracks << " !!!" << std::endl;
}
}  // namespace detail

// VideoSfMHandler::fullBA (VideoSfMHandler.cc:153-180) without the RPC arguments and the PLY dump: every frame of the session, the
// options as configured except for the valid-matches rule above.  (`reproject` — createTracks after the solve — is the caller's.)
inline bool fullBA(Session& sess, const SfmOptions& options, const int32_t maxIter, ceres::Solver::Summary* out = nullptr, bool progress = true,
                   std::vector<std::vector<double>>* covariances = nullptr) {
  SfmOptions opt = options;
  detail::relax_valid_matches_rule(sess, opt);
  return BA(sess, 0, (int32_t)sess.frames.size() - 1, opt, maxIter, out, progress, covariances);
}

// VideoSfMHandler::windowedBA (VideoSfMHandler.cc:185-214): frames [startFrame, endFrame], the poses before the window stay as they
// are; fixScale is switched off (:195 — the window is anchored by the frozen tracks) and the valid-matches rule counts the tracks of
// the whole session, as the reference does (:199 "TODO check tracks within window?").
inline bool windowedBA(Session& sess, const SfmOptions& options, const int32_t startFrame, const int32_t endFrame, const int32_t maxIter,
                       ceres::Solver::Summary* out = nullptr, bool progress = true, std::vector<std::vector<double>>* covariances = nullptr) {
  if (endFrame >= (int32_t)sess.frames.size()) throw std::out_of_range("windowedBA: endFrame");   // CHECK_LT (:192)
  SfmOptions opt = options;
  opt.ceres.fixScale = false;
  detail::relax_valid_matches_rule(sess, opt);
  return BA(sess, startFrame, endFrame, opt, maxIter, out, progress, covariances);
}

}  // namespace rsba_amd
