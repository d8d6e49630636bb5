// Drop-in replacements for rsba's per-pose prior functors: same struct names and factory signatures, typed handles that
// the facade lowers to rsba_set_pose_priors (kernels_pose_prior.hip) instead of Jet-autodiff functors.
//   vision::sfm::SphericalPrior    /root/reference/src/rsba/CeresHandler.h:36-50   block: pose[6]; 2 residuals
//   vision::sfm::GoodPosePrior     /root/reference/src/rsba/CeresHandler.h:52-73   blocks: pose0 (the prior)[6], pose[6]; 6 residuals
// CeresHandler attaches both without a loss function (:129, :198-201).
#pragma once
#include "ceres_facade.hpp"

namespace rsba_amd {

struct SphericalPrior {
  static const unsigned short NUM_RESIDUALS = 2;
  static ceres::CostFunction* Create() { return new ceres::PosePriorCost(1, 0.0, 0.0); }
};

struct GoodPosePrior {
  static const unsigned short NUM_RESIDUALS = 6;
  static ceres::CostFunction* Create(double rotation, double position) { return new ceres::PosePriorCost(0, rotation, position); }
};

}  // namespace rsba_amd
