// Device-resident problem state shared by the kernels and the C-ABI implementation.
//
// HBM layout (everything fp64 / int32, struct-of-arrays, rows padded to 64 elements = 512 B):
//   observations, frame-major (sorted by frame, stable):  xy[N] (double2), frame[N], point[N]
//   parameters:  poses[F][P][6], points[M][3], intr[NI][9], frame_intr[F]
//   evaluation:  res / jac in TILED component-major order: observations are grouped in tiles of 256 (one
//                workgroup); inside a tile every component is a contiguous run of 256 doubles, so each
//                store is one coalesced 512-B line per wave AND a workgroup's whole output is one
//                contiguous block (2 KB x components) — DRAM-page friendly.  Component c of observation i:
//                  base[(i / 256) * tile_stride + c * 256 + (i % 256)],  tile_stride = ncomp * 256
//                jac component = r*K + c (row r, column c); res component = r.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rsba {

struct DeviceProblem {
  // model
  int shutter, scan0, scan1, interp_rotation, calibrated, P;
  int F, M, NI;
  int64_t N, ntiles;             // observations, number of 256-observation tiles
  int K;                         // Jacobian columns per observation
  double huber_a;
  // observations (frame-major)
  const double2* xy;
  const int32_t* obs_frame;
  const int32_t* obs_point;
  // parameters
  double* poses;
  double* points;
  double* intr;
  const int32_t* frame_intr;     // [F]
  const uint8_t* frame_global;   // [F] or null (P == 2 only): 1 = a one-pose (global-shutter) frame of the session: tau = 0, the second pose slot is constant
  // column scales: 0 for a fixed coordinate, otherwise the Jacobi scale (1 before it is estimated)
  double* scale_pose;            // [F][P][6]
  double* scale_point;           // [M][3]
  double* scale_intr;            // [NI][9]
  // evaluation outputs
  double* res;                   // tiled component-major, 2 components
  double* jac;                   // tiled component-major, 2K components
  // LM mode only: a second, POINT-major copy of every observation's corrected record, one contiguous
  // run of rec_len = 2 + 2K doubles at slot obs_slot[i] (slots are sorted by point, then frame):
  //   [r0 r1 | Jp row0 (3) Jp row1 (3) | Jc row0 (K-3) Jc row1 (K-3)]
  // so that everything the point elimination needs about one point is contiguous in HBM.
  double* rec;                   // [N][rec_len]
  // A CANDIDATE's records go to a second buffer (round 6): a rejected step needs the records of the point it started from, and the loop
  // whose decisions are taken on the device cannot wait for the decision before it evaluates.  Which of the two holds the records of the
  // current point: rec, unless the device-side trust region says otherwise (ctl[kCtlRecSel], flipped by the deciding kernel on acceptance;
  // the host form swaps the two pointers instead).  null = one buffer (the host form evaluates candidates residual-only then).
  double* rec_alt;               // [N][rec_len] or null
  int rec_candidate;             // this launch's LM-mode evaluation is a candidate's: its records go to the buffer that is NOT current
  const int32_t* obs_slot;       // [N]
  // LM mode: the per-frame camera (and intrinsics border) blocks are formed inside the evaluation kernel (fp64 MFMA
  // over each wave's 64 observations) instead of from a second pass over the tiled Jacobian, which is then not
  // written at all.  Every wave stores the 16 x 16 blocks on and below the diagonal of [Ji | Jc | r]^T [Ji | Jc | r]
  // (one block, or three with rolling shutter + intrinsics) per frame it touches:
  double* cam_part;              // [segments][blocks][256]; null = write the tiled Jacobian
  const int32_t* wave_seg_base;  // [ceil(N / 64) + 1] first segment of each 64-observation wave
  const int32_t* frame_rank;     // [F] index of the frame among the frames that have observations
  double* cost_partial;          // [nblocks] 1/2 sum rho0 over non-dropped blocks of each workgroup
  double* fixed_partial;         // [nblocks] same over dropped (all-constant) blocks
  double* fail_partial;          // [nblocks] observations of each workgroup whose functor returned false
  int* fail_count;               // their total, written by the cost reduction (no atomics on the hot path)
  // frame-to-frame motion priors with a constant interFrameRatio (kernels_prior.hip); null = none
  const int32_t* prior_of;       // [F + 1] 1 = frame f carries a prior against frame f - 1 (entry F is 0)
  int prior_kind;                // 1 RsConstVeloPrior, 2 RsConstAccelerationPrior
  double prior_scale, prior_ratio;
  const double* prior_ratio_ptr; // where the interFrameRatio lives on the device (the loop whose decisions are taken on the device: the free ratio is part of its state); null = prior_ratio
  int prior_free;                // the interFrameRatio is a free parameter block: no prior block is "all constant" then
  // per-pose prior blocks (kernels_pose_prior.hip): GoodPosePrior on the listed pose blocks, each with its own
  // free priorPoses parameter block, and at most one SphericalPrior
  int pp_count;                  // GoodPosePrior blocks (0 = none)
  const int32_t* pp_block;       // [pp_count] pose block f * P + q
  double* pp_value;              // [pp_count][6] current priorPoses values
  double* pp_trial;              // [pp_count][6] candidate values
  double* pp_scale;              // [pp_count][6] column scales (1 before the Jacobi scale is estimated)
  double pp_rotation, pp_position;   // opt.ceres.trustPriorCamRotation / trustPriorCamPosition
  int pp_spherical;              // pose block carrying the SphericalPrior, -1 = none
  double* prior_partial;         // [2 * ceil(F / 64)] per-wave partial sums of the prior reductions
  unsigned* prior_ticket;        // arrival counter of those reductions (zero between launches)
  const double* ctl;             // trust-region state on the device (LmCtlSlot; solver.hip: the LM loop's decisions taken by a kernel), null = the host decides
};

// Trust-region control on the device: the scalars of Ceres' TrustRegionMinimizer loop live in HBM, a single-thread
// kernel takes its decisions (kernels_normal.hip: lm_decide_*), and the kernels of an iteration look at them instead of waiting for
// the host: the host enqueues iterations AHEAD and reads the state of each from a slot of host memory the last kernel of the iteration
// writes it to (stamped with the iteration's sequence number: no event, no copy in the stream).
enum LmCtlSlot : int {
  kCtlRadius = 0, kCtlDecrease = 1,       // trust-region radius, the factor of the next shrink (2, 4, ..)
  kCtlCost = 2, kCtlFixed = 3, kCtlGmax = 4,
  kCtlAccept = 5,                          // 1: the candidate of this iteration became x (its linearisation follows), 0: it did not
  kCtlStatus = 6,                          // 0: running; 1 + termination type: done; -1: the host must take over (a suspect factorisation)
  kCtlIteration = 7, kCtlInvalidStreak = 8, kCtlSuccessful = 9, kCtlUnsuccessful = 10, kCtlFinalCost = 11, kCtlNumTrace = 12,
  kCtlRecSel = 13,                         // problems that keep records: 1 = the records of the current point are in dp.rec_alt (flipped on every acceptance)
  kCtlPending = 16,                        // [4] relative decrease, cost change, step norm, model cost change of an accepted step whose record waits for its gradient
  kCtlSeq = 23,                            // in a snapshot on the host only: the sequence number of the iteration it was taken behind, written last
  kCtlSize = 24
};
__device__ __forceinline__ bool lm_stopped(const double* ctl) { return ctl && ctl[kCtlStatus] != 0.0; }
// the records of the current point (candidate = false) / where a candidate's evaluation puts its own (true)
__device__ __forceinline__ double* lm_records(const DeviceProblem& dp, bool candidate) {
  if (!dp.rec_alt) return dp.rec;
  const bool sel = dp.ctl && dp.ctl[kCtlRecSel] != 0.0;
  return (sel != candidate) ? dp.rec_alt : dp.rec;
}
__device__ __forceinline__ bool lm_not_accepted(const double* ctl) { return ctl && (ctl[kCtlStatus] != 0.0 || ctl[kCtlAccept] == 0.0); }

constexpr int kEvalBlock = 256;
constexpr int kStageFrames = 16;   // camera blocks staged through LDS per workgroup
// A wave's partial G = X^T X, X = [Ji | Jc | r] (ncol columns), as 16 x 16 blocks of v_mfma_f64_16x16x4_f64 in dp.cam_part ([segments][blocks][256]).
//   ncol <= 16: ONE block, rows and columns = the operand's columns.
//   16 < ncol <= 24 (rolling shutter + intrinsics: 22) — TWO products cover everything on and below the diagonal of a 22 x 22 Gram matrix, because a
//   product's two operands may select DIFFERENT columns and the upper triangle of a block is not needed (round 6; three blocks (0,0) (1,0) (1,1) until then):
//     block 0: rows = columns 0 .. 15,       columns = columns 0 .. 7 and 16 .. ncol-1   ->  G(i, j < 8) for i < 16, and G(i >= 16, j < 16) transposed
//     block 1: rows = columns 8 .. ncol-1,   columns = the same                          ->  G(i >= 8, j >= 8)
//   (selected columns behind the operand's last are a zero column.)  cam_part_entry() says where entry (a, b), a >= b, is.
__host__ __device__ constexpr int cam_part_blocks(int ncol) { return ncol <= 16 ? 1 : (ncol <= 24 ? 2 : (((ncol + 15) / 16) * ((ncol + 15) / 16 + 1)) / 2); }
__host__ __device__ constexpr int cam_part_entry(int a, int b) {   // entry (a, b), a >= b, of the two-block form -> block * 256 + row * 16 + column
  return (a >= 16 && b < 16) ? b * 16 + (a - 8) : (b < 8 ? a * 16 + b : 256 + (a - 8) * 16 + (b - 8));
}

enum EvalMode : int {
  kResidualOnly = 0,   // T=double path: residuals + cost (trial point of the trust-region loop)
  kRawJacobian = 1,    // what CostFunction::Evaluate returns: r and J, no loss, no masks (the metric)
  kLmJacobian = 2,     // loss-corrected, masked, column-scaled r and J for the normal equations
};

hipError_t launch_eval(const DeviceProblem& dp, EvalMode mode, hipStream_t stream);
int eval_num_blocks(int64_t n);
hipError_t launch_cost_reduce(const DeviceProblem& dp, double* out2, hipStream_t stream);
// tiled component-major res / jac -> rows[order[i]][..] in the caller's layout ([N][2] and [N][2][K], back to back)
hipError_t launch_untile(const DeviceProblem& dp, const int64_t* order, bool with_jacobians, double* res_rows, double* jac_rows, hipStream_t st);
// motion priors: cost2 += {cost, fixed cost} of the prior blocks at dp.poses (fail_count += invalid_blocks)
hipError_t launch_prior_cost(const DeviceProblem& dp, double* cost2, int invalid_blocks, hipStream_t st);
// per-pose priors: cost2 += {cost, fixed cost} of the blocks at dp.poses / dp.pp_value (fail_count += failed functors)
hipError_t launch_pose_prior_cost(const DeviceProblem& dp, double* cost2, hipStream_t st);
hipError_t launch_validate(const DeviceProblem& dp, double sq_threshold, double min_distance, uint8_t* valid, hipStream_t st);
hipError_t launch_scatter_flags(const uint8_t* in, const int64_t* order, int64_t n, uint8_t* out, hipStream_t st);   // out[order[i]] = in[i]
hipError_t launch_reproject(const DeviceProblem& dp, const int32_t* frames, const int32_t* points, int64_t n, double* xy_out, uint8_t* ok_out, hipStream_t st);

}  // namespace rsba
