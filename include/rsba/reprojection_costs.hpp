// Drop-in replacements for rsba's two bundle-adjustment cost functors: same struct names, same factory
// names and argument lists, but the objects they return are typed handles that the facade lowers to the
// flat HIP path instead of Jet-autodiff functors.
//   vision::ReprojectionError            /root/reference/src/rsba/video_bundler_free.h:17-101
//   vision::sfm::RsBundleAdjustment      /root/reference/src/rsba/VideoSfmBaRs.h:15-84
#pragma once
#include "ceres_facade.hpp"

namespace rsba_amd {

// global-shutter pinhole + Brown distortion reprojection error: blocks [cam 9]? pose[6] point[3]
struct ReprojectionError {
  static const unsigned short NUM_RESIDUALS = 2;
  // video_bundler_free.h:70-79  AutoDiffCostFunction<ReprojectionError,2,9,6,3>: intrinsics are a parameter block
  static ceres::CostFunction* Create(const double* const observed) {
    return new ceres::ReprojectionCost(false, true, observed, nullptr, nullptr, nullptr);
  }
  // video_bundler_free.h:82-91  AutoDiffCostFunction<ReprojectionError,2,6,3>: intrinsics copied in as data
  static ceres::CostFunction* Create(const double* const camera_params, const double* const observed) {
    return new ceres::ReprojectionCost(false, false, observed, camera_params, nullptr, nullptr);
  }
};

// rolling-shutter reprojection error: blocks [cam 9]? pose0[6] pose1[6] point[3]; the pose is interpolated
// by the observation's scan-line position; session / option fields are read when the problem is solved
struct RsBundleAdjustment {
  static const unsigned short NUM_RESIDUALS = 2;
  // VideoSfmBaRs.h:53-64  <RsBundleAdjustment,2,6,6,3>, intrinsics = sess.cam copied at construction (:19)
  static ceres::CostFunction* Create(const Session& sess, const SfmOptions& opt, const double* const observed) {
    return new ceres::ReprojectionCost(true, false, observed, sess.cam.data(), &sess, &opt);
  }
  // VideoSfmBaRs.h:68-80  <RsBundleAdjustment,2,9,6,6,3>
  static ceres::CostFunction* CreateWithCam(const Session& sess, const SfmOptions& opt, const double* const observed) {
    return new ceres::ReprojectionCost(true, true, observed, nullptr, &sess, &opt);
  }
};

}  // namespace rsba_amd
