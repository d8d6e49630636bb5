// Drop-in replacements for rsba's frame-to-frame motion prior functors: same struct names and factory signatures,
// typed handles that the facade lowers to rsba_set_motion_priors (kernels_prior.hip) instead of Jet-autodiff functors.
//   vision::RsConstVeloPrior            /root/reference/src/rsba/video_bundler_rs_inter.h:55-108
//   vision::RsConstAccelerationPrior    /root/reference/src/rsba/video_bundler_rs_inter.h:113-173
// Parameter blocks, in the reference's order: interFrameRatio[1], current frame first pose, current frame last pose,
// previous frame first pose, previous frame last pose.  The ratio block is either constant in the problem
// (CeresHandler.h:175-177, the case opt.ceres.interFrameRatio != 1) or free with the lower bound of :161 / :172 (the default).
#pragma once
#include "ceres_facade.hpp"

namespace rsba_amd {

struct RsConstVeloPrior {
  static const unsigned short NUM_RESIDUALS = 12;
  static ceres::CostFunction* Create(double scale) { return new ceres::MotionPriorCost(1, scale); }
};

struct RsConstAccelerationPrior {
  static const unsigned short NUM_RESIDUALS = 12;
  static ceres::CostFunction* Create(double scale) { return new ceres::MotionPriorCost(2, scale); }
};

}  // namespace rsba_amd
