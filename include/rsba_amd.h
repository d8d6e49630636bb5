/* rsba_amd — C ABI of the MI355X-native bundle-adjustment hot path.
 *
 * Drop-in boundary for the path rsba runs through Ceres-Solver today.  Every entry
 * point cites the reference interface it replaces; paths are relative to /root/reference/src/rsba/.
 * Plain pointers and sizes only; no C++ / torch types.  All functions return an rsba_status (0 = OK),
 * never throw, and — like ceres::Problem — are not re-entrant on one handle.
 *
 * The C++ facade in include/rsba/ceres_facade.hpp maps rsba's CostFunction / Problem / Solve usage
 * onto these calls; INTEGRATION.md shows the reference-side binding.
 */
#ifndef RSBA_AMD_H_
#define RSBA_AMD_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSBA_AMD_ABI_VERSION 3

typedef enum rsba_status {
  RSBA_OK = 0,
  RSBA_ERR_INVALID_ARGUMENT = 1,
  RSBA_ERR_NO_DEVICE = 2,        /* no gfx950 device / HIP runtime unusable: there is NO CPU fallback */
  RSBA_ERR_HIP = 3,              /* a HIP call failed: see rsba_last_error() */
  RSBA_ERR_EVALUATION_FAILED = 4,/* some functor returned false (point behind camera, mat/cam.h:410-412) */
  RSBA_ERR_OUT_OF_MEMORY = 5,
  RSBA_ERR_UNSUPPORTED = 6,
  RSBA_ERR_COMM = 7
} rsba_status;

/* mat/cam.h:37-41 */
enum { RSBA_SHUTTER_GLOBAL = 0, RSBA_SHUTTER_HORIZONTAL = 1, RSBA_SHUTTER_VERTICAL = 2 };

/* The residual blocks CeresHandler::Add (CeresHandler.h:94-390) creates from a Session
 * (sfm.thrift:13-74), flattened: one block per observation over caller-owned parameter arrays.
 *   poses_per_frame == 2 -> RsBundleAdjustment (VideoSfmBaRs.h:15-84; CeresHandler.h:245-265)
 *   poses_per_frame == 1 -> ReprojectionError  (video_bundler_free.h:17-101; CeresHandler.h:266-280)
 *   calibrated != 0      -> intrinsics are data (Create(cam,obs) / Create(sess,opt,obs))
 *   calibrated == 0      -> intrinsics are a parameter block (CreateWithCam / Create(obs)),
 *                           shared sess.cam or per frame f.cam via frame_intrinsics (CeresHandler.h:260,277)
 * Parameter arrays are written back by rsba_solve / rsba_download_parameters, as ceres::Solve
 * mutates the blocks it was given (CeresHandler.h:253-255). */
typedef struct rsba_problem_desc {
  int32_t shutter;               /* sess.rs */
  int32_t scanlines[2];          /* sess.scanlines */
  int32_t interpolate_rotation;  /* opt.model.interpolateRotation (SfmOptions.h:25) */
  int32_t calibrated;            /* opt.model.calibrated (SfmOptions.h:27) */
  int32_t poses_per_frame;       /* f.poses.size(): 1 or 2 */
  int32_t num_frames, num_points, num_intrinsics;
  int64_t num_observations;
  double* poses;                 /* [F][P][6]  angle-axis world->camera, camera centre (mat/cam.h:354-366) */
  double* points;                /* [M][3] */
  double* intrinsics;            /* [NI][9] {fx,fy,k1,k2,p1,p2,k3,cx,cy} (mat/cam.h:23-34) */
  const int32_t* frame_intrinsics; /* [F] or NULL (= every frame uses intrinsics[0]) */
  const double* obs_xy;          /* [N][2] */
  const int32_t* obs_frame;      /* [N] */
  const int32_t* obs_point;      /* [N] */
  const uint8_t* pose_fixed_mask;   /* [F][P]: bit i set = coordinate i held fixed (SubsetParameterization,
                                       CeresHandler.h:350-382); 0x3f = SetParameterBlockConstant (:342-348); NULL = free */
  const uint8_t* point_constant;    /* [M] SetParameterBlockConstant on points (CeresHandler.h:288-300); NULL = free */
  const uint8_t* intrinsics_constant; /* [NI] (CeresHandler.h:284,347); NULL = free */
  double huber_a;                /* opt.ceres.huberLoss: > 0 -> one shared ceres::HuberLoss(a) (CeresHandler.h:85-90) */
} rsba_problem_desc;

typedef struct rsba_handle rsba_handle;   /* stands for the ceres::Problem member of CeresHandler (CeresHandler.h:78) */

/* ceres::Solver::Options fields rsba writes (CeresHandler.h:403-412, VideoSfMHandler.cc:579-583)
 * plus the Ceres 1.9 defaults the LM loop depends on. */
typedef struct rsba_solver_options {
  int32_t max_num_iterations;               /* 50 (CeresHandler.h:405) / 20 (VideoSfMHandler.h:63) */
  int32_t jacobi_scaling;                   /* 1 */
  int32_t max_num_consecutive_invalid_steps;/* 5 */
  int32_t minimizer_progress_to_stdout;     /* CeresHandler.h:404 */
  double initial_trust_region_radius;       /* 1e4 */
  double max_trust_region_radius;           /* 1e16 */
  double min_trust_region_radius;           /* 1e-32 */
  double min_relative_decrease;             /* 1e-3 */
  double min_lm_diagonal, max_lm_diagonal;  /* 1e-6, 1e32 */
  double function_tolerance;                /* 1e-6 */
  double gradient_tolerance;                /* 1e-10 */
  double parameter_tolerance;               /* 1e-8 */
  int32_t level_scheduled_cholesky;         /* 0.  1 = factor the reduced camera system with one launch per elimination
                                             * level instead of the persistent task-DAG kernel (same arithmetic, same results,
                                             * no communication between workgroups inside a launch; slower) */
  int32_t profile_phases;                   /* 0.  1 = time the phases of every LM iteration with HIP events on the solver's
                                             * stream; read with rsba_get_phase_times after the solve */
} rsba_solver_options;

enum { RSBA_CONVERGENCE = 0, RSBA_NO_CONVERGENCE = 1, RSBA_FAILURE = 2 };   /* ceres::TerminationType subset */

/* ceres::IterationSummary subset (what minimizer_progress_to_stdout prints) */
typedef struct rsba_iteration {
  int32_t iteration, step_is_valid, step_is_successful, reserved;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius, model_cost_change;
} rsba_iteration;

/* ceres::Solver::Summary subset rsba reads (VideoSfMHandler.cc:593-596,627-630) + phase times that
 * FullReport() prints */
typedef struct rsba_solver_summary {
  int32_t termination_type, num_successful_steps, num_unsuccessful_steps, num_iterations;
  int32_t num_residual_blocks, num_residual_blocks_reduced, num_parameters_reduced, is_solution_usable;
  double initial_cost, final_cost, fixed_cost;
  double total_time_s, residual_jacobian_time_s, linear_solver_time_s;
  int32_t num_dag_fallbacks;     /* linear solves of the persistent Cholesky driver whose residual check failed and that were
                                  * repeated on the level schedule (0 in every run so far) */
  int32_t reserved;
} rsba_solver_summary;

/* Pointers into HBM for callers that keep results on the device.  All fp64, tiled component-major:
 * observations (INTERNAL frame-major order; order_host[i] = caller's index of internal observation i)
 * are grouped in tiles of `tile` = 256; component c of observation i lives at
 *   base[(i / tile) * ncomp * tile + c * tile + (i % tile)]
 * with ncomp = 2 for residuals and 2*K for jacobians (component r*K + c = row r, column c; columns
 * ordered [cam 9]? [pose0 6] [pose1 6]? [point 3]). */
typedef struct rsba_device_view {
  double* residuals;
  double* jacobians;        /* NULL until an evaluation WITH Jacobians has run on the handle (the buffer — 16 K bytes per observation — is allocated then, not by the view) */
  int64_t tile;
  int32_t jacobian_cols;    /* K */
  int32_t reserved;
  const int64_t* order_host;/* host array [N] */
  double* poses; double* points; double* intrinsics;   /* device parameter arrays */
} rsba_device_view;

int32_t rsba_abi_version(void);
const char* rsba_status_string(int32_t status);
const char* rsba_last_error(void);                 /* thread-local detail for the last non-OK status */
int32_t rsba_device_count(int32_t* count);         /* RSBA_ERR_NO_DEVICE when none: callers must fail, not fall back */

/* == ceres::Problem construction + the AddResidualBlock loop (CeresHandler.h:78, 208-301): uploads the
 * flat problem once; `device` is the HIP device ordinal (one process per GPU). */
int32_t rsba_create(const rsba_problem_desc* desc, int32_t device, rsba_handle** out);
void rsba_destroy(rsba_handle* h);                 /* == ~Problem: frees every device buffer exactly once */

/* Run on a caller-owned HIP stream (hipStream_t as void*), e.g. torch's current stream; NULL = the
 * handle's own stream. */
int32_t rsba_set_stream(rsba_handle* h, void* hip_stream);

/* Parameter blocks are caller-owned in Ceres; these move them between the caller's arrays and HBM. */
int32_t rsba_upload_parameters(rsba_handle* h, const double* poses, const double* points, const double* intrinsics);
int32_t rsba_download_parameters(rsba_handle* h, double* poses, double* points, double* intrinsics);

/* The metric's "residual + Jacobian evaluation": == CostFunction::Evaluate over every residual block
 * (VideoSfmBaRs.h:53-80 / video_bundler_free.h:70-91 through AutoDiffCostFunction).  Asynchronous on the
 * handle's stream; results stay in HBM (rsba_get_device_view).  with_jacobians == 0 is the T=double path. */
int32_t rsba_evaluate_device(rsba_handle* h, int32_t with_jacobians);

/* == ceres::Problem::Evaluate(EvaluateOptions(), &cost, &residuals, &gradient, &jacobian)
 * (CeresHandler.h:386-387 uses the cost-only form).  Host outputs in the CALLER's observation order:
 * residuals [N][2]; jacobians [N][2][K] raw CostFunction blocks (no loss, no masks); gradient
 * [F*P*6 | M*3 | NI*9] = loss-corrected J^T r with zeros at fixed coordinates; cost = 1/2 sum rho(|r|^2).
 * Any output may be NULL.  num_failed receives the number of blocks whose functor returned false;
 * the status is then RSBA_ERR_EVALUATION_FAILED.  Motion prior blocks (rsba_set_motion_priors) count in cost, gradient
 * and num_failed; residuals / jacobians cover the observation blocks only. */
int32_t rsba_evaluate(rsba_handle* h, double* cost, double* residuals, double* jacobians, double* gradient, int64_t* num_failed);

/* Loss-corrected, masked normal-equation blocks at the current parameters (no damping, no Jacobi
 * scaling) — what Ceres' SchurEliminator consumes: U [F][CD][CD], gc [F][CD] with
 * CD = 6*poses_per_frame, V [M][3][3], gp [M][3]; host arrays, any may be NULL.  These per-camera blocks
 * are also the payload of the multi-GPU exchange.  Calibrated problems only.  Motion prior blocks contribute their
 * share of U and gc; the (f, f-1) blocks they add to the reduced camera system are not part of this output. */
int32_t rsba_normal_equations(rsba_handle* h, double* U, double* gc, double* V, double* gp);

int32_t rsba_get_device_view(rsba_handle* h, rsba_device_view* view);

/* Average device time of one evaluation launch, measured with hipEvents on the handle's stream
 * around `iters` back-to-back launches (after `warmup` untimed ones). */
int32_t rsba_time_evaluate(rsba_handle* h, int32_t with_jacobians, int32_t warmup, int32_t iters, double* avg_ms);

/* == ceres::Solve(options, &problem, &summary) with linear_solver_type = SPARSE_SCHUR
 * (CeresHandler.h:394-426): LM trust region, Schur elimination of the points, Cholesky of the reduced
 * camera system, all on the device; parameters are written back to the arrays given to rsba_create.
 * trace (may be NULL) receives up to trace_capacity iteration records.
 * The trust-region decisions themselves (TrustRegionMinimizer's accept / reject, radius, convergence tests) are taken by a device kernel
 * for every problem this call takes (calibrated, one shared or several intrinsics blocks; with or without motion priors — interFrameRatio
 * known or free —, GoodPosePrior blocks and the SphericalPrior; one rank or several) — the host then waits once per iteration for the
 * state, never inside one; the same rules on the host with options.profile_phases, when a rank of a sharded solve asks for it (no
 * observations) and under RSBA_DEVICE_LM=0.  The two forms produce the same iteration records bit for bit. */
void rsba_default_solver_options(rsba_solver_options* opt);
int32_t rsba_solve(rsba_handle* h, const rsba_solver_options* opt, rsba_solver_summary* summary,
                   rsba_iteration* trace, int32_t trace_capacity);

/* Measurement aids: where an LM iteration spends its device time, and the sizes of the symbolic plan the
 * kernels' algorithmic bytes / flops follow from.  Phase p covers the launches listed; ms[p] is the HIP-event time summed
 * over the last rsba_solve that ran with options.profile_phases != 0, calls[p] how many times the phase ran. */
enum {
  RSBA_PHASE_EVAL_LM = 0,        /* eval_kernel<.., kLmJacobian> + cost reduction: r, J (loss-corrected, scaled) as per-wave camera blocks (point-major records only with several intrinsics blocks) */
  RSBA_PHASE_CAMERA_BLOCKS = 1,  /* per-frame J^T J / J^T r blocks (+ intrinsics border) from the per-wave partials */
  RSBA_PHASE_POINT_BLOCKS = 2,   /* V_j, g_p,j */
  RSBA_PHASE_POINT_FACTOR = 3,   /* (V_j + D^2)^-1 factors, z_j */
  RSBA_PHASE_PROJECT = 4,        /* P records (+ virtual intrinsics records) */
  RSBA_PHASE_SCHUR = 5,          /* clear + Schur tile products + merge: S, rhs */
  RSBA_PHASE_CHOLESKY = 6,       /* factor + forward / backward solve of the reduced camera system */
  RSBA_PHASE_BACK_SUBSTITUTE = 7,/* point steps + model cost change */
  RSBA_PHASE_CANDIDATE = 8,      /* x + delta, |step|, |x| */
  RSBA_PHASE_EVAL_TRIAL = 9,     /* evaluation of the candidate + cost reduction: in LM mode (the linearisation an accepted step re-uses) when the problem keeps no records, residual-only otherwise */
  RSBA_PHASE_PRIORS = 10,        /* motion-prior blocks, cost and model terms */
  RSBA_PHASE_EXCHANGE = 11,      /* multi-GPU all-reduces (pack / collective / unpack) */
  RSBA_PHASE_OTHER = 12,         /* diagonal clamp, gradient norm, scalar packing */
  RSBA_NUM_PHASES = 13
};
typedef struct rsba_phase_times { double ms[RSBA_NUM_PHASES]; int32_t calls[RSBA_NUM_PHASES]; int32_t reserved; } rsba_phase_times;
int32_t rsba_get_phase_times(rsba_handle* h, rsba_phase_times* out);
const char* rsba_phase_name(int32_t phase);
typedef struct rsba_plan_stats {
  int64_t tiles, factor_tiles, levels, tasks;       /* 48 x 48 tiles of S, tiles of its factor after fill, elimination levels, Cholesky tasks */
  int64_t schur_entries, schur_chunks;              /* (point, tile pair) entries of the Schur work list, workgroups */
  int64_t schur_block_products;                     /* CD x 3 by 3 x CD block products that are not structurally zero */
  int64_t cholesky_flops;                           /* of the tile factorisation incl. fill, forward and backward solve */
  int64_t exchange_doubles;                         /* payload (2) of the multi-GPU exchange: the structurally non-zero tiles of S + rhs */
  int64_t schur_groups;                             /* (point, frame tile) groups of P records */
  int64_t schur_mfma_issued, schur_launches;        /* fp64 MFMAs (2048 flop each) the Schur kernel issued so far — all-zero operand blocks are skipped — over so many launches */
  int64_t sharded_factorisation;                    /* 1: several ranks, each factoring its own part of the elimination tree (the points follow rsba_partition_points);
                                                     * exchange_doubles is then the separators' tiles | their rhs rows | the gather of the camera step */
  int64_t separator_tiles, separator_factor_tiles;  /* tile columns in the separators all ranks share, and tiles of the factor inside them (what exchange (2) carries) */
  int64_t local_tasks, separator_tasks;             /* Cholesky tasks of this rank's part (forward) / of the separators incl. both backward solves */
  int64_t local_levels, separator_levels;           /* the two dependency chains: elimination levels inside this rank's part / levels that hold a separator column */
  int64_t schur_group_bytes, schur_factored_groups; /* bytes of all groups: 3 x 48 doubles each in full form; 80 doubles for the groups stored FACTORED (two-pose frame tiles: the
                                                     * 6-row factor q = Jq^T Jp L^-T per frame + tau instead of the 12 rows (1 - tau) q | tau q), and how many those are */
  int64_t device_loop_solves, host_loop_solves;     /* rsba_solve calls on this plan whose trust-region loop ran without the host (decisions by device kernels: every problem rsba_solve takes,
                                                     * phase timing excepted) / with the host deciding */
} rsba_plan_stats;
int32_t rsba_get_plan_stats(rsba_handle* h, rsba_plan_stats* out);   /* runs the symbolic phase if it has not run yet */

/* ---- the steps either side of the solve: batched reprojection / validation filter ----
 * == vision::sfm::validate(sess, f, opt, pt, obs) for every observation of the problem
 * (struct/VideoSfM.cc:159-169 with getPose :103-133; callers: CeresHandler.h:239-243 revalidateReprojections,
 * VideoSfMHandler.cc evalTracks / createTracks): valid[i] = 1 iff |camera centre at the observation's scan
 * line - point| >= min_distance (opt.tracks.minDistanceToCamera) and the squared reprojection error <
 * sq_threshold (opt.tracks.sqrdThreshold).  Unlike the cost functor, tau is taken from the TRUE observation
 * (x for HORIZONTAL, y for VERTICAL).  valid is a host array [N] in the caller's observation order. */
int32_t rsba_validate_observations(rsba_handle* h, double sq_threshold, double min_distance, uint8_t* valid);
/* == vision::sfm::reproject(sess, f, opt, pt, obs) (struct/VideoSfM.cc:139-155) for n (frame, point) pairs:
 * fixed point on the scan-line time from the principal point, at most 49 projections, converged when the
 * projection moves <= 1e-3 px.  xy_out [n][2], ok_out [n] (host arrays). */
int32_t rsba_reproject(rsba_handle* h, const int32_t* frames, const int32_t* points, int64_t n, double* xy_out, uint8_t* ok_out);

/* The same two filters for ONE frame without a handle — the shape vision::sfm::validate / reproject are called in
 * (CeresHandler.h:220-243: the observations of the frame being added against the tracks' points): cam[9], poses
 * [num_poses][6], points [n][3], obs_xy [n][2]; valid / ok_out [n], xy_out [n][2]; all host arrays.  num_poses 1: that pose;
 * 2: interpolate_rs at the item's scan line; MORE ("fullDoF", a pose per scan line): the pose getPose picks for the item,
 * poses[round(clamp(line, 0, num_poses - 1))] with line = x for HORIZONTAL, y otherwise (struct/VideoSfM.cc:118-132) — reproject
 * re-picks it in every step of its fixed point, as the reference does.  Each host
 * thread keeps one device arena, pinned staging buffer and stream between calls, so a call is one upload, one launch and
 * one download. */
int32_t rsba_validate_frame(int32_t device, const double* cam, const double* poses, int32_t num_poses, int32_t shutter, const int32_t* scanlines,
                            int32_t interpolate_rotation, const double* points, const double* obs_xy, int64_t n, double sq_threshold,
                            double min_distance, uint8_t* valid);
int32_t rsba_reproject_frame(int32_t device, const double* cam, const double* poses, int32_t num_poses, int32_t shutter, const int32_t* scanlines,
                             int32_t interpolate_rotation, const double* points, int64_t n, double* xy_out, uint8_t* ok_out);

/* == the covariance blocks VideoSfMHandler::BA prints with opt.debug.calcCovariances (VideoSfMHandler.cc:602-621:
 * ceres::Covariance::Compute on (p0,p0), (p0,p1), (p1,p1) of a frame): cov [CD][CD] row-major,
 * CD = 6 * poses_per_frame, = the (frame, frame) block of (J^T J)^-1 at the current parameters, loss function applied,
 * zero rows / columns at fixed coordinates.  RSBA_ERR_UNSUPPORTED when J^T J is rank deficient (Compute returns false). */
int32_t rsba_pose_covariance(rsba_handle* h, int32_t frame, double* cov);

/* == the frame-to-frame motion priors CeresHandler::Add attaches to a rolling-shutter frame (CeresHandler.h:147-185;
 * SURVEY §8f row f1): one 12-residual block per listed frame f >= 1 over (f.poses[0], f.poses[1], f-1.poses[0],
 * f-1.poses[1]) — RsConstVeloPrior (kind 1, video_bundler_rs_inter.h:55-108) or RsConstAccelerationPrior (kind 2,
 * :113-173) with weight `scale` (opt.ceres.constFrameVelocity / constFrameAcceleration), rotation rows down-scaled
 * by 0.01, under the problem's loss function (huber_a).  inter_frame_ratio is opt.ceres.interFrameRatio: a CONSTANT
 * block — the case the reference takes when the option is != 1 (CeresHandler.h:175-177) — unless
 * rsba_set_inter_frame_ratio_free (below) makes it the free, lower-bounded parameter of the reference's default.
 * The blocks count in cost, gradient, num_residual_blocks and the solve; their validity flag (ratio >= 0 resp.
 * >= DBL_EPSILON) fails the evaluation like any functor returning false.  Needs poses_per_frame == 2 and a calibrated
 * or shared-intrinsics problem.  frames strictly increasing; must precede the first solve / gradient call;
 * kind 0 or count 0 removes them.  With rsba_set_exchange every rank passes the same list (rank 0 contributes them). */
int32_t rsba_set_motion_priors(rsba_handle* h, int32_t kind, double scale, double inter_frame_ratio, const int32_t* frames, int32_t count);
/* The reference's DEFAULT for these priors (opt.ceres.interFrameRatio left at 1, CeresHandler.h:175): the ratio is a free
 * parameter block with a lower bound (0 for kind 1, DBL_EPSILON for kind 2; :161,:172).  is_free != 0 makes rsba_solve
 * treat it so: one more unknown of the LM (its column is a 1-wide dense border of the reduced camera system, served by a
 * second solve through the factorisation), the candidate projected onto the bound as Ceres' ParameterBlock::Plus does,
 * the gradient norm taken of the projected gradient.  Ceres' extra projected line search for bounded problems (>= 1.10)
 * is not restated: steps follow the plain trust-region rules.  rsba_get_inter_frame_ratio returns the current value
 * (the solved one after rsba_solve).  Must precede the first solve / gradient call; evaluate / gradient treat the ratio
 * as the constant it currently is; rsba_pose_covariance includes its column (the pose block of the bordered inverse). */
int32_t rsba_set_inter_frame_ratio_free(rsba_handle* h, int32_t is_free);
int32_t rsba_get_inter_frame_ratio(rsba_handle* h, double* ratio);

/* == the per-pose prior blocks of CeresHandler::Add, neither with a loss function (nullptr in the reference):
 *   GoodPosePrior (CeresHandler.h:52-73, attached at :188-204 when opt.ceres.trustPriorCamRotation / trustPriorCamPosition are set
 *     and the frame has priorPoses): 6 residuals W (prior - pose) over TWO parameter blocks — the frame's priorPoses[i], a FREE
 *     block like every block Ceres is handed, and poses[i]; rotation rows times `rotation`, position rows times `position`; the
 *     functor fails when residual[0] >= 1.  pose_blocks[k] = f * poses_per_frame + i, distinct; prior_values [count][6] are the
 *     caller's priorPoses blocks: rsba_solve writes the solved values back, rsba_upload_parameters re-reads them.
 *   SphericalPrior (:36-50, attached at :127-130 to poses[0] of frame 1 of a session that starts at the origin): residuals
 *     |rot|^2 and 1e20 (1 - |cx| - |cy| - |cz|); fails when |rot|^2 >= 1.  spherical_pose_block = its pose block, -1 = none.
 *     (A 1e20-weighted residual: once it is met, its value is rounding noise times 1e20 — the cost Ceres and this library
 *     report then carries that noise; see DESIGN.md.)
 * The blocks count in cost, gradient (pose coordinates), num_residual_blocks and the solve; the priorPoses blocks are eliminated
 * in closed form like points.  Must precede the first solve / gradient call.  With rsba_set_exchange every rank passes the same
 * blocks (rank 0 contributes them). */
int32_t rsba_set_pose_priors(rsba_handle* h, double rotation, double position, const int32_t* pose_blocks, double* prior_values, int32_t count,
                             int32_t spherical_pose_block);

/* Sessions that mix rolling-shutter frames (two poses) with one-pose frames: CeresHandler::Add picks the functor per frame by
 * f.poses.size() (CeresHandler.h:245-286) — RsBundleAdjustment over (poses[0], poses[1], point) or ReprojectionError over
 * (getPose(...), point).  Create the problem with poses_per_frame = 2 and flag the one-pose frames here: is_global [num_frames],
 * 1 = the frame's observations use poses[f][0] alone (tau = 0; the validation / reprojection filters follow: getPose returns the
 * single pose, struct/VideoSfM.cc:103-133); the second pose slot of such a frame is not a parameter block — it is held constant and
 * left untouched.  NULL clears the flags.  Call before the first evaluation / solve.
 * Frames with MORE than two poses ("fullDoF", a pose per scan line; struct/VideoSfM.cc:83-97, CeresHandler.h:266-285) are the same
 * case once more: an observation of such a frame is a ReprojectionError block over ONE pose block, getPose's pick
 * poses[round(clamp(line, 0, size - 1))].  Every pose block that some observation picks is a "frame" of the flat problem (one-pose
 * problem, or a flagged frame of a two-pose one), obs_frame names it, pose blocks nobody picks are not part of the problem — as in
 * Ceres, which never hears of them.  include/rsba/ceres_handler.hpp (getPose + the facade's block-address lowering) and
 * rsba_amd/problem.py::lower_scanline_poses do exactly this. */
int32_t rsba_set_global_shutter_frames(rsba_handle* h, const uint8_t* is_global);

/* The symbolic phase of a handle's first solve works in ~40 bytes of host memory per observation.  That scratch is kept by the
 * library between handles (a fresh handle per call is windowedBA's pattern, VideoSfMHandler.cc:185-214, and mapping / unmapping
 * it per call cost more than the passes that fill it), and so are the device blocks, streams, events and the small pinned block of
 * destroyed handles (rsba_amd/csrc/devmem.hpp; device blocks up to RSBA_DEVICE_CACHE_MB, default 2048); this call gives all of it
 * back.  Safe at any time; the next handle simply allocates again. */
void rsba_release_host_scratch(void);

/* == the RANSAC hypotheses of vision::solveRsPnPRansac (solveRSpnp.cpp:413-524), batched: task t is
 * what pnpTask (:265-335) does for the subset subsets[t][0..m) of the n float points —
 *   skipped (status 0, nothing else written) when drop_coincident != 0 and two of its 3-D points coincide (:283-293;
 *   the final refinement over the inliers, :496-502, is the same call with drop_coincident = 0);
 *   vision::solveRsPnP (:100-192) from init_poses: ceres::Solve over the two pose blocks of one rolling-shutter frame,
 *   one RsBA<float> residual block per point (:25-97; w2i without validation, tau from observed_x), max_num_iterations
 *   (the reference: 10), all other options Ceres defaults; poses_out[t] = the result if usable (status 1), else the
 *   initial poses (status 2);
 *   num_inliers[t] = points whose float-rounded projection at the scan line of the TRUE observation lies closer than
 *   reprojection_error (float distance) to the observation (:225-258, :304-310).
 * Poses are rsba's 6-vectors (angle-axis world->camera, camera centre): [pose, pose2]; the rvec/tvec conversions of
 * :119-128, :166-176 and the OpenCV GS initialisation (:111-117) stay with the caller.  init_stride 12 = one initial
 * pair per task, 0 = one shared pair.  All pointers are host arrays; final_cost / num_inliers may be NULL.
 * rsba_pnp_inliers returns the inlier flags [n] of one pose pair (the winning hypothesis' list, :312-326). */
int32_t rsba_pnp_tasks(int32_t device, const double* cam, int32_t shutter, const int32_t* scanlines, const float* object_points,
                       const float* image_points, int32_t n, const int32_t* subsets, int32_t m, int32_t num_tasks,
                       const double* init_poses, int32_t init_stride, int32_t max_num_iterations, int32_t drop_coincident, float reprojection_error,
                       double* poses_out, uint8_t* status, double* final_cost, int32_t* num_inliers);
int32_t rsba_pnp_inliers(int32_t device, const double* cam, int32_t shutter, const int32_t* scanlines, const float* object_points,
                         const float* image_points, int32_t n, const double* poses, float reprojection_error, uint8_t* inlier_mask);

/* ---- multi-GPU: one process per GPU, observations partitioned BY POINT, cameras replicated ----
 * (the reference is single-process; this is the exchange step SURVEY §8e derives for the path).
 * Every rank creates a handle over its own observations (all frames / points arrays are full size, a rank
 * simply holds no observations of the points it does not own).  Per LM iteration the solver all-reduces
 *   (1) the per-camera gradient blocks g_c and diag(U)  + cost / failure scalars (+ a slot per rank: its gradient maximum)  [2*F*CD + 3 (+ world) doubles]
 *   (2) the packed non-zero tiles of its partial reduced camera system S and its rhs  [tile pairs*48*48 + F*CD] — or, when the points are cut along
 *       the top separators of the elimination tree (rsba_partition_points) and every rank factors its own part, (2') the separators' tiles only,
 *       between the two launches of the factorisation, and (4) the camera step, each rank its rows  [npad]
 *   (3) twelve step scalars (model decrease, |step|^2, |x|^2, gradient maximum, trial cost, failure flags, the factorisation's verification flag)
 * through RCCL directly (rsba_set_exchange_rccl, below) or through the callback below, for hosts that bring their own
 * transport (the tests stage it through gloo).
 * op: 0 = sum, 1 = max.  The buffer is device memory; the collective must be ordered after prior work
 * on `hip_stream` and complete (or be stream-ordered) before the callback returns.  Return 0 on success. */
typedef int32_t (*rsba_allreduce_fn)(void* ctx, double* device_buffer, int64_t count, int32_t op, void* hip_stream);
int32_t rsba_set_exchange(rsba_handle* h, rsba_allreduce_fn fn, void* ctx, int32_t rank, int32_t world);

/* Structure of the reduced camera system: mask[a*F + b] != 0 (a >= b) iff frames a and b share a point in
 * THIS rank's observations.  All ranks must install the same (union) structure before the first solve so
 * that their tile layouts — and therefore exchange buffer (2) — coincide; frame_obs_count [F] likewise
 * carries the per-frame observation count (summed over ranks before it is set back). */
int32_t rsba_get_block_structure(rsba_handle* h, uint8_t* mask, int64_t* frame_obs_count);
int32_t rsba_set_block_structure(rsba_handle* h, const uint8_t* mask, const int64_t* frame_obs_count);
/* The two calls above in one, over the installed exchange: every rank calls it after rsba_set_exchange[_rccl] and before
 * the first solve; the co-visibility masks are OR-ed and the per-frame counts summed through the all-reduce itself, and
 * the partition is checked (RSBA_ERR_INVALID_ARGUMENT when some point has observations on more than one rank). */
int32_t rsba_sync_block_structure(rsba_handle* h);

/* What the exchange of a sharded solve carried (a first multi-rank run should be diagnosable from its numbers alone): per kind of
 * collective the calls and fp64 elements since the handle was created, and — after an rsba_solve with options.profile_phases — the
 * HIP-event time of those of that solve, measured on the solver's stream around each collective. */
enum {
  RSBA_EXCHANGE_SETUP = 0,      /* co-visibility structure, problem-size counts, the form of the plan (once per handle / solve) */
  RSBA_EXCHANGE_CAMERA = 1,     /* (1) per-camera gradient blocks g_c | diag(U) | cost scalars | every rank's max |g_i| over its points, once per linearisation */
  RSBA_EXCHANGE_SYSTEM = 2,     /* (2) the reduced camera system: every structurally non-zero tile | rhs (replicated factorisation) or the
                                 *     separators' tiles | their rhs rows between the two launches of a sharded factorisation */
  RSBA_EXCHANGE_SCALARS = 3,    /* (3) step scalars and the verification flag: one sum of 12 doubles per iteration (the gradient max-norm rides in (1);
                                 *     with per-pose priors it still takes a MAX all-reduce of its own) */
  RSBA_EXCHANGE_STEP = 4,       /* (4) the gather of the camera step (sharded factorisation only) */
  RSBA_EXCHANGE_POINTS = 5,     /* the merge of the solved points at the end of a solve */
  RSBA_NUM_EXCHANGES = 6
};
typedef struct rsba_exchange_stats {
  int64_t calls[RSBA_NUM_EXCHANGES], doubles[RSBA_NUM_EXCHANGES];
  double ms[RSBA_NUM_EXCHANGES];
  int32_t rank, world;
} rsba_exchange_stats;
int32_t rsba_get_exchange_stats(rsba_handle* h, rsba_exchange_stats* out);
const char* rsba_exchange_name(int32_t kind);
/* == ncclGetVersion / ncclCommCount / ncclCommUserRank of a communicator, as the library's own RCCL sees it (-1 where the symbol is missing) */
int32_t rsba_rccl_describe(void* nccl_comm, int32_t* version, int32_t* nranks, int32_t* rank);

/* Which rank should own which point: owner [num_points] (host array), computed on the host from the WHOLE problem's description
 * (no device needed; every rank of a job computes the same answer from the same description).  Any by-point partition gives a
 * correct sharded solve; THIS one places the cut along the top separators of the nested dissection that orders the reduced camera
 * system's 48 x 48 tile columns: every rank's points then touch the tiles of its own subtree (and the separators) only, the columns
 * of its subtree are complete on that rank, and the solver factors them there — exchange (2) shrinks from every non-zero tile of S
 * to the separators' tiles, the factorisation's work is shared out (rsba_plan_stats.sharded_factorisation says whether the plan
 * took that form).  world == 1: all zero.  num_top_tiles (may be NULL): tile columns in the separators the ranks share.
 * RSBA_ERR_UNSUPPORTED when the co-visibility graph cannot be cut into `world` parts (a handful of frames, or not connected). */
int32_t rsba_partition_points(const rsba_problem_desc* desc, int32_t world, int32_t* owner, int32_t* num_top_tiles);

/* Native transport: RCCL's ncclAllReduce over xGMI, issued by the solver on its own HIP stream — nothing of the host
 * language runs inside an LM iteration.  librccl is resolved at run time (RSBA_RCCL_LIB, else an RCCL already loaded in
 * the process, else librccl.so.1); single-GPU users never load it.
 *   rsba_rccl_get_unique_id   one rank (usually 0) fills id[RSBA_RCCL_UNIQUE_ID_BYTES] (== ncclGetUniqueId); the host
 *                             hands the bytes to the other ranks by whatever means it has (MPI, a file, torch.distributed)
 *   rsba_rccl_comm_create     == ncclCommInitRank on `device`; collective over all ranks; *comm_out is an ncclComm_t
 *   rsba_set_exchange_rccl    installs the all-reduce over an ncclComm_t (one made above or the caller's own) on the
 *                             handle; must precede the first solve / gradient call, like rsba_set_exchange
 *   rsba_rccl_comm_destroy    == ncclCommDestroy (after the handles that use it)
 * After a sharded rsba_solve every rank holds the complete solved parameter arrays: the points are merged over the ranks
 * (each from the rank that owns its observations) before they are written back. */
#define RSBA_RCCL_UNIQUE_ID_BYTES 128
int32_t rsba_rccl_get_unique_id(void* id);
int32_t rsba_rccl_comm_create(const void* id, int32_t rank, int32_t world, int32_t device, void** comm_out);
void rsba_rccl_comm_destroy(void* comm);
int32_t rsba_set_exchange_rccl(rsba_handle* h, void* nccl_comm, int32_t rank, int32_t world);

#ifdef __cplusplus
}
#endif
#endif /* RSBA_AMD_H_ */
