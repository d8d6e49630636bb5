/* ORACLE — TEST INFRASTRUCTURE ONLY (see rsba_oracle_math.hpp header).  C interface used through
 * ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product path
 * (rsba_amd/, include/) never includes, links or loads anything in oracle/.
 *
 * PARITY UNPINNED beyond mat_test.cc: see rsba_oracle_math.hpp. */
#ifndef RSBA_ORACLE_H_
#define RSBA_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Flat description of what CeresHandler::Add (CeresHandler.h:94-390) builds out of a Session:
 * one residual block per observation over user-owned parameter arrays. */
typedef struct orc_problem {
  int32_t shutter;               /* mat/cam.h:37-41: 0 GLOBAL, 1 HORIZONTAL, 2 VERTICAL (sess.rs) */
  int32_t scanlines[2];          /* sess.scanlines */
  int32_t interpolate_rotation;  /* opt.model.interpolateRotation */
  int32_t calibrated;            /* opt.model.calibrated: 1 = intrinsics are constants */
  int32_t poses_per_frame;       /* 2 -> RsBundleAdjustment, 1 -> ReprojectionError (CeresHandler.h:245,266) */
  int32_t num_frames, num_points, num_intrinsics;
  int64_t num_observations;
  double* poses;                 /* [F][P][6]  in/out */
  double* points;                /* [M][3]     in/out */
  double* intrinsics;            /* [NI][9]    in/out when !calibrated */
  const int32_t* frame_intrinsics; /* [F] index into intrinsics, NULL = all 0 (sess.cam) */
  const double* obs_xy;          /* [N][2] */
  const int32_t* obs_frame;      /* [N] */
  const int32_t* obs_point;      /* [N] */
  const uint8_t* pose_fixed_mask;   /* [F][P] bit i = coordinate i fixed (SubsetParameterization); 0x3f = constant block */
  const uint8_t* point_constant;    /* [M] */
  const uint8_t* intrinsics_constant; /* [NI] */
  double huber_a;                /* opt.ceres.huberLoss; <= 0 = no loss (CeresHandler.h:87-89) */
  /* Frame-to-frame motion priors (CeresHandler.h:147-185) with opt.ceres.interFrameRatio != 1, i.e. the ratio block
   * set constant (CeresHandler.h:175-177).  One 12-residual block per listed frame f >= 1 over the blocks
   * (f.poses[0], f.poses[1], f-1.poses[0], f-1.poses[1]); the shared loss function applies to it (:155,:166). */
  int32_t prior_kind;            /* 0 none, 1 RsConstVeloPrior, 2 RsConstAccelerationPrior (video_bundler_rs_inter.h:55-173) */
  int32_t num_priors;
  const int32_t* prior_frames;   /* [num_priors], strictly increasing, each >= 1 */
  double prior_scale;            /* opt.ceres.constFrameVelocity / constFrameAcceleration */
  double inter_frame_ratio;      /* opt.ceres.interFrameRatio */
  /* Per-pose prior blocks of CeresHandler::Add, neither with a loss function (nullptr in the reference):
   *   GoodPosePrior (CeresHandler.h:52-73, attached at :188-204): 6 residuals over TWO parameter blocks, the frame's
   *   priorPoses[i] — free, like every block Ceres is handed — and poses[i]: (prior - pose), rotation rows times
   *   pose_prior_rotation, position rows times pose_prior_position; the functor fails when residual[0] >= 1.
   *   SphericalPrior (:36-50, attached at :127-130 to poses[0] of frame 1 of a session that starts at the origin):
   *   residuals |rot|^2 and 1e20 (1 - |cx| - |cy| - |cz|); fails when |rot|^2 >= 1. */
  int32_t num_pose_priors;
  const int32_t* pose_prior_block;  /* [num_pose_priors] pose block index f * P + q */
  double* pose_prior_values;        /* [num_pose_priors][6] the priorPoses blocks, in/out */
  double pose_prior_rotation, pose_prior_position;   /* opt.ceres.trustPriorCamRotation / trustPriorCamPosition */
  int32_t spherical_pose_block;     /* pose block carrying the SphericalPrior; only read when has_spherical != 0 */
  int32_t has_spherical;
  int32_t no_validate;           /* 1 = the RS-PnP functor RsBA: w2i(..., validate = false) (solveRSpnp.cpp:67) */
  int32_t ratio_free;            /* 1 = interFrameRatio is a free, lower-bounded parameter block (the reference's default, option left at 1:
                                  * CeresHandler.h:161,172,175); orc_solve updates inter_frame_ratio in place */
  const uint8_t* frame_global;   /* [F] or NULL, poses_per_frame == 2 only: 1 = the frame has ONE pose in the session (CeresHandler.h:266-285: the
                                  * ReprojectionError functor on poses[f][0]); its second pose slot is not a parameter block (the caller marks it constant) */
} orc_problem;

/* Ceres 1.9 Solver::Options subset (defaults: SURVEY Appendix C.5) */
typedef struct orc_options {
  int32_t max_num_iterations;
  int32_t jacobi_scaling;
  int32_t max_num_consecutive_invalid_steps;
  int32_t num_threads;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
} orc_options;

enum { ORC_CONVERGENCE = 0, ORC_NO_CONVERGENCE = 1, ORC_FAILURE = 2 };

typedef struct orc_iteration {
  int32_t iteration, step_is_valid, step_is_successful, pad;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius, model_cost_change;
} orc_iteration;

typedef struct orc_summary {
  int32_t termination_type, num_successful_steps, num_unsuccessful_steps, num_iterations;
  int32_t num_residual_blocks, num_residual_blocks_reduced, num_parameters_reduced, pad;
  double initial_cost, final_cost, fixed_cost;
} orc_summary;

void orc_default_options(orc_options* o);
/* default OpenMP team size for the calls that do not take a thread count */
void orc_set_num_threads(int32_t n);

/* Number of Jacobian columns per observation: 9*(!calibrated) + 6*P + 3 */
int32_t orc_jacobian_cols(const orc_problem* p);

/* What CostFunction::Evaluate returns for every residual block, via forward-mode duals:
 * residuals [N][2]; jacobians [N][2][K] — per row the blocks are ordered [cam 9]?[pose0 6][pose1 6]?[point 3].
 * ok[N] (may be NULL) = functor return value.  No loss, no masks.  Returns number of failed blocks. */
int64_t orc_evaluate_blocks(const orc_problem* p, double* residuals, double* jacobians, uint8_t* ok, int32_t num_threads);

/* The Ceres way to run the same evaluation: one heap-allocated cost object per observation, each
 * evaluated through Dual<K> into per-block Jacobian arrays.  Used by the CPU baseline. */
int64_t orc_evaluate_blocks_ceres_style(const orc_problem* p, double* residuals, double* jacobians, int32_t num_threads);

/* Residuals only (T=double path). */
int64_t orc_evaluate_residuals(const orc_problem* p, double* residuals, uint8_t* ok, int32_t num_threads);

/* Problem::Evaluate: cost = 1/2 sum rho(|r|^2); gradient (loss-corrected J^T r on the masked tangent
 * space, zeros at fixed coordinates) laid out [F*P*6 | M*3 | NI*9]; any output may be NULL.
 * Returns 0 on success, 1 if any functor failed. */
int32_t orc_evaluate(const orc_problem* p, double* cost, double* gradient);

/* Loss-corrected, masked normal-equation blocks at the current parameters (no damping, no scaling):
 * U [F][CD][CD], gc [F][CD], V [M][3][3], gp [M][3] with CD = 6*P.  Calibrated problems only. */
int32_t orc_normal_equations(const orc_problem* p, double* U, double* gc, double* V, double* gp);

/* ceres::Solve(SPARSE_SCHUR) restated: LM trust region with exact Schur-complement solves.
 * Parameters are overwritten in place.  trace (may be NULL) receives up to trace_cap iteration records. */
int32_t orc_solve(orc_problem* p, const orc_options* opt, orc_summary* summary, orc_iteration* trace, int32_t trace_cap);

/* ceres::Covariance blocks of one frame's poses (VideoSfMHandler.cc:602-621): cov [CD][CD], CD = 6 * poses_per_frame;
 * returns 1 on success, 0 if J^T J is rank deficient or a functor fails. */
int32_t orc_pose_covariance(const orc_problem* p, int32_t frame, double* cov);

/* One RANSAC hypothesis of solveRsPnPRansac (solveRSpnp.cpp:265-335 pnpTask + :100-192 solveRsPnP), poses in rsba's
 * convention (angle-axis world->camera, camera centre) — the rvec/tvec conversions either side are the caller's:
 * the m points subset[] of the float object / image points (skipped when drop_coincident and two of them coincide, :283-293), LM over the two
 * pose blocks with RsBA<float> residual blocks (max_iter iterations, Ceres defaults), the result kept if usable, then
 * the inliers among all n points: float distance between the observation and the float-rounded projection at the
 * TRUE observation's scan line < reprojection_error (:225-258, :304-310).
 * Returns 0 = skipped (nothing written), 1 = done.  inlier_mask may be NULL. */
int32_t orc_pnp_task(const double cam[9], int32_t shutter, const int32_t scanlines[2], const float* object_points, const float* image_points,
                     int32_t n, const int32_t* subset, int32_t m, int32_t drop_coincident, const double init_poses[12], int32_t max_iter, double reprojection_error,
                     double poses_out[12], int32_t* usable, double* final_cost, int32_t* num_inliers, uint8_t* inlier_mask);

/* Scalar entry points for the known-answer tests (mat_test.cc) */
void orc_angle_axis_rotate(const double w[3], const double p[3], double out[3]);
void orc_lerp_rotation(const double r0[3], const double r1[3], double tau, double out[3]);
void orc_distort(const double cam[9], const double img[2], double out[2]);
int32_t orc_undistort(const double cam[9], const double img[2], double out[2]);
void orc_w2c(const double pose[6], const double X[3], double out[3]);
void orc_c2w(const double pose[6], const double pt[3], double out[3]);
int32_t orc_w2i(const double cam[9], const double pose[6], const double X[3], double out[2], int32_t validate);
int32_t orc_direction_world(const double pose[6], const double X[3], double d[3]);
int32_t orc_c2direction(const double pose[6], const double pt[3], double d[3]);
int32_t orc_direction_pixel(const double cam[9], const double pose[6], const double xy[2], double d[3]);
int32_t orc_ray_intersect(const double p2[3], const double d1[3], const double d2[3], double dist[3]);
int32_t orc_triangulate(const double c1[3], const double d1[3], const double c2[3], const double d2[3], double p[3]);
int32_t orc_validate(const double cam[9], const double pose[6], const double xy[2], const double X[3], double sq_threshold);
int32_t orc_ray_dist(const double cam[9], const double pose[6], const double obs[2], const double cam2[9],
                     const double pose2[6], const double obs2[2], double dist[3]);
double orc_norm3(const double v[3]);
void orc_interpolate_rs(const double p0[6], const double p1[6], int32_t shutter, const int32_t scan[2],
                        const double obs[2], int32_t interp_rotation, double out[6]);
void orc_huber(double a, double s, double rho[3]);
/* struct/VideoSfM.cc:83-97 getPose, frames with more than two poses: index of the pose an observation uses */
int32_t orc_scanline_pose_index(int32_t nposes, int32_t shutter, const double obs[2]);
/* struct/VideoSfM.cc:139-155 reproject; :159-169 validate (nposes 1, 2 or more: getPose's three cases) */
int32_t orc_reproject(const double cam[9], const double* poses, int32_t nposes, int32_t shutter, const int32_t scan[2],
                      int32_t interp_rotation, const double X[3], double sq_threshold, double obs[2]);
int32_t orc_validate_obs(const double cam[9], const double* poses, int32_t nposes, int32_t shutter, const int32_t scan[2],
                         int32_t interp_rotation, const double X[3], const double obs[2], double sq_threshold, double min_dist);

#ifdef __cplusplus
}
#endif
#endif
