// Device-side state of the LM / Schur solver (calibrated problems: camera blocks of CD = 6P unknowns,
// point blocks of 3).  Everything lives in HBM for the whole solve; the host reads back a handful of
// scalars per iteration to take the trust-region decisions.
//
//   J^T J = [ U  W ; W^T V ],  U = blockdiag(U_f) (CDxCD per frame),  V = blockdiag(V_j) (3x3 per point)
//   with V'_j = V_j + D_p^2 = L_j L_j^T and  P_o = Jc_o^T Jp_o L_j^-T  (CD x 3 per observation):
//     S   = U + D_c^2 - sum_o,o' in same point  P_o P_o'^T          (reduced camera system)
//     rhs = g_c - sum_o P_o z_j,   z_j = L_j^-1 g_p,j
//     y_p,j = L_j^-T ( z_j - sum_o P_o^T y_c(frame(o)) )            (back-substitution)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/rsba_amd.h"
#include "device_state.hpp"

namespace rsba {

constexpr int kSchurChunk = 512;   // entries per workgroup of the Schur kernel (its four waves take every fourth)
constexpr int kMaxRankSlots = 64;   // per-rank slots behind the camera exchange's payload (more ranks than this: the gradient maximum takes its own all-reduce)
constexpr int kTile = 48;   // Cholesky tile: 4 rolling-shutter frames (12 unknowns) or 8 global-shutter frames
constexpr int kGroupFull = 3 * kTile;   // doubles of a P-record group in full form: [3][kTile]
// ... and factored (two-pose frames: the 12 camera-side rows of a frame are (1 - tau) q | tau q with q = Jq^T Jp L^-T, 6 x 3):
//   [c][16] sources 0..15 of coordinate c = 0, 1, 2 (source s = 6 x frame-in-tile + pose coordinate) | [c][8] sources 16..23 | tau[4] | pad[4]
constexpr int kGroupFactored = 80;

// Intrinsics as parameter blocks (opt.model.calibrated == false: the shared sess.cam and / or per-frame f.cam blocks,
// CeresHandler.h:256-264,273-280): the 9 coordinates of block c ride as NPF = ceil(9/CD) extra "pseudo frames"
// F + c*NPF + v behind the real frames in the camera-side ordering (zero-scaled padding fills the last one), and every
// point gets one VIRTUAL observation slot per (intrinsics block it is seen through, pseudo frame) whose P record is
// Q_j,c = sum_{o through c} Ji_o^T Jp_o L^-T.  The pair lists of the symbolic phase then produce the rows of the reduced
// system that belong to the intrinsics (a dense 9-wide border for the shared block, band rows for per-frame blocks) with
// the same Schur and Cholesky kernels; only the (non block-diagonal) U cross terms need their own reductions.
struct SolverDev {
  int F, Fx, NPF;               // real frames, frames + pseudo frames, pseudo frames PER intrinsics block (0 when calibrated)
  int NIB;                      // intrinsics parameter blocks (0 when calibrated; 1 = the shared sess.cam; per-frame f.cam blocks: one each)
  const int32_t* intr_frame_ptr;// [NIB+1] frames that use each intrinsics block,
  const int32_t* intr_frame_list;//        ascending
  int64_t nvgroups;             // (point, intrinsics block it is seen through) pairs: each owns NPF virtual slots N + g*NPF + v
  const int32_t* vgroup_point;  // [nvgroups]
  const int32_t* vgroup_intr;   // [nvgroups]
  const int64_t* point_vgroup;  // [M] with one intrinsics block: the point's virtual group, -1 = not observed
  int CD;                       // 6 * P
  int64_t n, npad;              // camera unknowns F*CD, padded to a multiple of kTile
  int nt;                       // npad / kTile
  int nslots;                   // structurally non-zero tiles of the factor (each kTile*kTile doubles)
  // structure
  const int64_t* frame_ptr;     // [F+1] frame-major observation ranges
  const int64_t* point_ptr;     // [M+1] slot ranges per point
  const int32_t* slot_frame;    // [N + M*NPF]  (virtual slots behind the real ones: N + j*NPF + v)
  const int32_t* slot_point;    // [N + M*NPF]
  // point-elimination work list (symbolic phase): entries = (point, pair of frame tiles I >= J)
  int FT;                       // frames per tile = kTile / CD
  int ntp;                      // structurally non-zero tile pairs of S (before fill), frame order
  const int32_t* tp_I;          // [ntp]
  const int32_t* tp_J;
  const int64_t* tp_ptr;        // [ntp+1] into the entry list
  const uint32_t* ent_groups;   // [nent][2] where the point's (tile, layer) group on the I side and on the J side starts in Pm: element offset (a multiple of 16) | kind in bit 0 (1 = factored)
  const double2* slot_xy;       // [N] observations in slot order — calibrated problems only: their point-side passes recompute the records (lm_record.hpp); null = records in dp.rec
  const uint32_t* slot_gpos;    // [N + virtual] where the slot's P record goes: element offset of its group in Pm (a multiple of 16) | position of its frame in the tile << 1 | kind of the group (bit 0: factored)
  int64_t ngroups;              // (point, tile, layer) groups: each owns one block of Pm — kGroupFull doubles ([3][kTile]) or, a two-pose frame tile of a problem that recomputes
                                //   its records, kGroupFactored (the 6-row factor per frame + tau: see kernels_normal.hip) — one more, all zero, sits behind the last: padding entries of the Schur chunks
  uint32_t zero_off;            // element offset of that all-zero group
  const uint8_t* tile_factored; // [nt] 1 = the groups of this frame tile are stored factored (null: none is)
  int fused_sweep;              // 1 = the projection pass and the virtual-record sweep of the shared intrinsics block are ONE launch (kernels_normal.hip, virtual_project_rc_kernel)
  int all_real_factored;        // 1 = every tile that holds a real frame is stored factored: the projection pass stages 19 doubles per slot instead of 36 (kernels_normal.hip, project_rc_kernel)
  int lerp_rot;                 // interpolateRotation of a rolling-shutter model: the rotation rows of a factored group carry (1 - tau) / tau like the translation rows (else 1 / 0)
  const uint16_t* ent_mask;     // [nent] bit 3 I + J: block rows 16 I .. of the I-side group and 16 J .. of the J-side group both contain a frame that sees the point
  const int32_t* ent_pt;        // [nent] point index; top bit set = the entry carries the rhs term P z
  int schur_linear;             // debugging (RSBA_SCHUR_LINEAR=1): blockIdx -> chunk without the XCD map
  long long* schur_trace;       // debugging (RSBA_SCHUR_TRACE=<file>): [nchunk][8] {workgroup, xcc/cu id, start, tables staged, loop done, end (100 MHz ticks), entries, -} of the last launch
  int schur_variant;            // tuning aid (RSBA_SCHUR_VARIANT): 0 = two groups in flight, two waves per SIMD, resident workgroups; 1 = three groups, one wave; 2 = a workgroup per chunk; 4 / 5 = ablations
  int nchunk;                   // workgroups of the Schur kernel: kSchurChunk entries each
  const int32_t* chunk_tp;      // [nchunk]
  const int64_t* chunk_e0;      // [nchunk] first entry of the chunk
  const int32_t* chunk_n;       // [nchunk] entries in the chunk (<= kSchurChunk)
  const int4* chunk_info;       // [nchunk] the same in ONE 16-byte read: {first entry (low, high word), entries, bit 0 = a tile paired with itself | bit 1 / 2 = I / J side stored factored}
  unsigned* schur_next;         // [8][16] next list position of every XCD's eighth of the chunk list (the persistent form of the kernel: kernels_normal.hip) | [128] workgroups done
  const int32_t* tp_chunk0;     // [ntp+1] range of each tile pair in tp_chunk_list
  const int32_t* tp_chunk_list; // chunk ids of each tile pair, in entry order (heads of pre-merged groups for very long lists)
  int npremerge;                // groups of chunks summed ahead of the merge
  const int32_t* pm_ptr;        // [npremerge+1] into pm_list
  const int32_t* pm_list;       // chunk ids of each group; the sum lands in the first one's partial tile
  const int32_t* tp_dst;        // [ntp] packed tile slot that receives the pair
  const uint8_t* tp_trans;      // [ntp] 1 = the tile ordering swapped I and J: store transposed
  const int64_t* tp_add;        // [ntp][FT][FT] offset into U of the J^T J block to add, -1 none
  const int32_t* tp_desc;       // [ntp][16] the pair's line for the merge kernel: I, J, packed slot, flags (transposed | I side factored | J side factored), its range in tp_chunk_list, -, -, the first eight chunk ids
  double* schur_part;           // [nchunk][kTile*kTile + kTile] partial tiles (+ rhs partials) of the chunks
  // numeric
  double* U;                    // [F][CD][CD] frame blocks | [NPF][F][CD][CD] (pseudo v of the frame's intrinsics block) x frame | [NIB][NPF][NPF][CD][CD]
  double* gc;                   // [Fx][CD]     (scaled) J_c^T r, intrinsics gradient in the pseudo frames
  double* intr_part;            // [F][9*10/2 + 9] per-frame partials of Ji^T Ji and Ji^T r
  double* trial_intr;           // [NI][9] candidate intrinsics
  const double* inprog_intr;    // [NIB*NPF*CD]
  double* V;                    // [M][6]  xx xy xz yy yz zz
  double* gp;                   // [M][3]
  double* diag_c;               // [F*CD]  clamped squared column norms (LM "diagonal_")
  double* diag_p;               // [M*3]
  double* Linv;                 // [M][6]  lower-triangular inverse of chol(V')
  double* z;                    // [M][3]
  double* Pm;                   // P records by (point, tile) group; full form [3][kTile], component-major: row c of group g holds coordinate c of the
                                //   point against the kTile camera-side rows of the tile (FT frames x CD; zero where a frame does not see the point) —
                                //   one 16-row operand slice of the Schur kernel's MFMAs is ONE aligned 128-B line
  unsigned long long* schur_mfma_count;   // MFMAs the Schur kernel actually issued, summed over its launches (all-zero 16 x 16 operand blocks are skipped)
  double* S;                    // [nslots][kTile][kTile] packed tiles of the reduced camera system / its factor
  double* Lf;                   // [nslots][kTile][kTile] the factor's sub-diagonal tiles (S keeps the reduced system itself)
  double* zv;                   // [npad] forward solve z, directly followed by
  double* yv;                   // [npad] the camera step y (copied to rhs when the solve is done)
  // A second right-hand side that shares the factorisation (the free interFrameRatio's column b of the normal equations, a 1-wide border of S):
  // z2 = L^-1 b is formed by FWD2 tasks beside the factorisation, the ETA task turns the two forward solves into the ratio's step
  // eta = (g_s - s z2.z) / (h_s + D - s^2 z2.z2) and the BACK tasks solve L^T y = z - (s eta) z2 — ONE backward solve, no launch of its own.
  double* zv2;                  // [npad] write-once cells of z2 (null: one right-hand side)
  double* ceta;                 // [8] write-once cell {s eta} the BACK tasks wait for
  const double* border2;        // [npad] b
  double* rt;                   // [kRtSize] the ratio's scalars on the device (RatioSlot)
  double* Winv;                 // [nt][kTile][kTile] inverses of the factored diagonal tiles (by tile index)
  double* chol_part;            // [all chunks][kTile*kTile + kTile] partial update tiles (+ rhs partials)
  double* Xpub;                 // [nt][kTile][kTile] by DIAG item: tile (j, k*) less its updates, published by its SUB task for the DIAG task of column j
  double* rhs;                  // [npad] directly behind S (one exchange buffer): the right-hand side of the reduced system
  const double* step;           // [npad] the camera step y_c the point-side passes and the candidate read: the solve's y where it landed (sv.yv), or
                                //   rhs when the free interFrameRatio combined two solves there
  double* udiag;                // [F*CD] diag(U), global after the exchange
  double* xbuf;                 // [2*F*CD + 3] exchange buffer: g_c | diag(U) | cost, fixed cost, failed blocks
  const double* ctl;            // trust-region state on the device (device_state.hpp: LmCtlSlot), null = the host decides and passes the radius by value
  int lead;                     // 1 on the rank that contributes the replicated terms (D_c^2, g_c, camera norms)
  const double* frame_lead;     // [nt * FT] sharded factorisation: 1 where THIS rank adds the replicated terms of the frame to its partial S (its own part;
                                //   rank 0: the separators), null = `lead` decides for every frame
  double* yp;                   // [M][3]
  double* trial_poses;          // candidate x + delta
  double* trial_points;
  const double* inprog_pose;    // [F*CD] 1 if the coordinate's block is part of the reduced program
  const double* inprog_point;   // [M*3]
  double* partial;              // scratch for block partial sums
  double* partial_c;            // the candidate's two sums when they are reduced together with the model cost change (device-side trust region)
  double* scalars;              // [16] results of reductions (see ScalarSlot)
  int* chol_fail;               // set when a pivot is not positive / not finite
};

enum RatioSlot : int {
  kRtH = 0, kRtG = 1,            // J~_ratio^T J~_ratio, J~_ratio^T r~ of the last linearisation (prior_border_kernel)
  kRtDot1 = 2, kRtDot2 = 3,      // z2.z, z2.z2 of the last solve (ETA task)
  kRtDiagTerm = 4, kRtGs = 5, kRtScale = 6,   // h_s + D / radius, g_s, the ratio's Jacobi scale: what the ETA task reads
  kRtEta = 7, kRtC = 8,          // the ratio's scaled step eta and s eta (its step in its own units is -s eta)
  kRtRatio = 9, kRtRatioNew = 10, kRtDiag = 11, kRtLb = 12,   // device-side trust region: the ratio, its candidate, its clamped LM diagonal, its lower bound
  kRtRatioEval = 13,             // the candidate as the prior blocks are evaluated with it (the ratio itself when the candidate is not finite)
  kRtSize = 16
};

enum ScalarSlot : int {
  kModelCostChange = 0, kStepSq = 1, kXSq = 2, kGradMax = 3, kCost = 4, kFixedCost = 5, kEvalFailed = 6, kSolveFailed = 7,
  kDagSuspect = 11,   // set by the verification of the persistent Cholesky driver (slots 8-10: exchange scratch of the problem-size counts)
};

// Dynamic LDS above the 64 KB default needs the kernel's cap raised, once per (kernel, device).
hipError_t allow_dynamic_lds_impl(const void* kernel, size_t bytes);   // capi.hip
template <class K>
inline hipError_t allow_dynamic_lds(K kernel, size_t bytes) { return allow_dynamic_lds_impl(reinterpret_cast<const void*>(kernel), bytes); }

// Cholesky task plan (cholesky.hip): flattened work lists of the symbolic phase, device pointers
struct CholPlan {
  const int32_t *upd, *diag_info, *diag_ptr, *diag_list, *sub_info, *sub_ptr, *sub_list, *sub_col, *diag_own, *sub_own, *back_info, *back_ptr, *back_list;
  const int32_t* diag_fuse;  // per DIAG item: the SUB item of its last contributor, whose product with W the DIAG task forms itself (-1: none)
  const int32_t* sub_pub;    // per SUB item: DIAG item that takes its X = S_ij - updates from sv.Xpub (-1: nobody)
  int ndiag;                 // DIAG items = columns (the last ndiag tasks are the BACK tasks)
  const int32_t* tasks;      // [ntasks][2] {kind, item} in a topological order
  int ntasks;
  unsigned int* ticket;
  long long* trace;          // debugging (RSBA_CHOL_TRACE): [ntasks][8] {workgroup, claimed, inputs ready, 4 task-specific stamps, done} in 10 ns ticks
  int nslots, nparts;
  // second right-hand side (free interFrameRatio), per plan: a sharded factorisation runs its FWD2 / ETA tasks in two launches
  const int32_t* fwd_range;   // [ndiag][2] contributors (positions in diag_list) a FWD2 / FWD2P task of this plan sums over
  const int32_t* diag_toprow; // [ndiag] index of the item's column among the separators' tile columns, -1 = a part's column
  const double* fwd2_minus;   // [separator tile columns][kTile] sum over the ranks of what their parts' columns contribute to z2's right-hand side (exchange (2')); null: nothing to subtract
  double* fwd2_partial;       // ... and where THIS rank's share of it goes (FWD2P tasks, before the exchange)
  const uint8_t* eta_tiles;   // [nt] tile columns the ETA task of this plan takes the dots z2.z, z2.z2 over (null: all)
  const double* eta_extra;    // [2] + the other columns' share, summed by the exchange (null: none)
  double* eta_partial;        // non-null: the task only leaves its two dots here (launch A of a sharded factorisation)
};

// FWD2: item = DIAG item whose column's z2 it forms; FWD2P: the same item's sum over THIS rank's part only (sharded: what travels)
enum : int { kTaskUpdate = 0, kTaskDiag = 1, kTaskSub = 2, kTaskBack = 3, kTaskFwd2 = 4, kTaskEta = 5, kTaskFwd2P = 6 };

// per-pose priors (kernels_pose_prior.hip): linearisation of the priorPoses coordinates [pp_count][6] and where the pose
// entries sit in the packed tiles
struct PosePriorDev {
  double *v0, *g0, *cross;       // scaled J^T J diagonal, J^T r, and the (prior, pose) cross term of each coordinate
  double* diag;                  // LM "diagonal_" of the coordinates
  const int32_t* tile_diag_slot; // [nt] packed slot of the diagonal tile of each (unpermuted) tile index
};
hipError_t launch_pose_prior_blocks(const DeviceProblem& dp, const SolverDev& sv, const PosePriorDev& pp, hipStream_t st);
hipError_t launch_pose_prior_scale(const DeviceProblem& dp, const PosePriorDev& pp, hipStream_t st);
hipError_t launch_pose_prior_clamp(const DeviceProblem& dp, const PosePriorDev& pp, double lo, double hi, hipStream_t st);
hipError_t launch_pose_prior_gradmax(const DeviceProblem& dp, const SolverDev& sv, const PosePriorDev& pp, hipStream_t st, double* out = nullptr);   // out: one more partial maximum instead of folding into scalars[kGradMax]
hipError_t launch_pose_prior_take(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);   // device-side trust region: the accepted candidate's priorPoses values
hipError_t launch_pose_prior_reduce(const DeviceProblem& dp, const SolverDev& sv, const PosePriorDev& pp, double radius, hipStream_t st);
hipError_t launch_pose_prior_step(const DeviceProblem& dp, const SolverDev& sv, const PosePriorDev& pp, double radius, hipStream_t st);

// cholesky.hip
hipError_t launch_chol_level(const SolverDev& sv, const CholPlan& pl, int kind, int first, int count, hipStream_t st);
// res = rhs - S y with den = |rhs| + |S||y| over the packed tiles; *flag = 1 when |res| > tol * den somewhere, left alone otherwise (sticky; cholesky.hip)
// slot_tiles [nslots][2] = {row tile, column tile} (unpermuted tile indices) of every packed tile
// b_rhs: the right-hand side the system was solved for (a private copy: the caller overwrites sv.rhs with the step while the check runs on its own stream)
// row_mine (may be null): [nt] 1 = the rows of this tile are checked (sharded factorisation: the rows whose tiles are all on this rank)
hipError_t launch_chol_verify(const SolverDev& sv, const int32_t* slot_tiles, const double* b_rhs, double* res, double* den, double tol, double* flag, hipStream_t st, const uint8_t* row_mine = nullptr);
struct DagArgs { SolverDev sv; CholPlan pl; };   // device copy the persistent kernel reads its state through (uploaded once per plan)
hipError_t launch_chol_solve(const SolverDev& sv, const CholPlan& pl, const DagArgs* device_args, const double* b2, double* zy2, unsigned int* ticket, int workgroups, hipStream_t st);   // one more right-hand side through the factor of the last launch_chol_dag / level run
hipError_t launch_chol_solve_level(const SolverDev& sv, const CholPlan& pl, bool backward, int first, int count, const double* b2, double* zy2, hipStream_t st);   // the same tasks, one launch per level
hipError_t launch_chol_dag(const SolverDev& sv, const CholPlan& pl, const DagArgs* device_args, int workgroups, bool one_per_cu, hipStream_t st);   // one_per_cu: LDS request above half a CU's, so that two never share one

// kernels_normal.hip
hipError_t launch_camera_blocks(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st, bool take_candidate = false, bool padding_is_zero = false);   // take_candidate: launch_lm_take_candidate's copy rides along
hipError_t launch_point_blocks(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);
hipError_t launch_slot_xy(const DeviceProblem& dp, double2* slot_xy, hipStream_t st);   // slot_xy[obs_slot[i]] = xy[i]
hipError_t launch_jacobi_scale(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);           // scale = mask / (1 + sqrt(diag))
hipError_t launch_clamp_diagonal(const DeviceProblem& dp, const SolverDev& sv, double lo, double hi, hipStream_t st);
hipError_t launch_gradient_max(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);           // -> scalars[kGradMax]
hipError_t launch_point_factor(const DeviceProblem& dp, const SolverDev& sv, double radius, hipStream_t st, const double* clamp = nullptr);   // clamp = {lo, hi}: launch_clamp_diagonal's job rides along
hipError_t launch_project(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);
bool project_covers_virtual_records(const DeviceProblem& dp, const SolverDev& sv);   // launch_project runs the fused sweep: launch_virtual_records has nothing left to do
hipError_t launch_clear_system(const SolverDev& sv, hipStream_t st);   // S = 0 (fill tiles start from zero)
hipError_t launch_schur_blocks(const DeviceProblem& dp, const SolverDev& sv, double radius, hipStream_t st);
// sharded factorisation: the exchange between its two launches (kernels_normal.hip)
hipError_t launch_top_assemble(const SolverDev& sv, const int32_t* slots, const int32_t* info, const int32_t* asm_ptr, const int32_t* asm_list, const int32_t* top_tiles, int ntop_slots, double* buf, hipStream_t st);
hipError_t launch_top_unpack(const SolverDev& sv, const int32_t* slots, const int32_t* info, const int32_t* top_tiles, int ntop_slots, const double* buf, hipStream_t st);
hipError_t launch_step_rows(const double* yv, const uint8_t* row_mine, int64_t npad, double* ybuf, hipStream_t st);
hipError_t launch_zero_tiles(double* S, const int32_t* slots, int n, hipStream_t st);
hipError_t launch_back_substitute(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);
hipError_t launch_model_cost_change(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);      // -> scalars[kModelCostChange]
hipError_t launch_candidate(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);              // trial params, |step|^2, |x|^2
// trust-region control on the device: the two decisions of an iteration (after the candidate's evaluation; after an accepted step's
// linearisation) and the hand-over of an accepted candidate (kernels_normal.hip)
struct LmRules { int32_t max_num_iterations, max_num_consecutive_invalid_steps; double max_trust_region_radius, min_trust_region_radius, min_relative_decrease, function_tolerance, gradient_tolerance, parameter_tolerance; };
hipError_t launch_lm_decide_step(const SolverDev& sv, double* ctl, const LmRules& rules, rsba_iteration* trace, int trace_cap, hipStream_t st);
hipError_t launch_lm_decide_gradient(const SolverDev& sv, double* ctl, const LmRules& rules, rsba_iteration* trace, int trace_cap, hipStream_t st);
hipError_t launch_linearize_blocks(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st, bool* done);   // launch_camera_blocks(take_candidate) + launch_point_blocks by one launch, where that applies
hipError_t launch_lm_take_candidate(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);   // x = x + delta where ctl says "accepted"
// the same steps in fewer launches, for the loop that runs without the host (solver.hip: device-side trust region) — same arithmetic, same order:
hipError_t launch_candidate_and_model_cost(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);   // launch_model_cost_change + launch_candidate: the three sums by one launch
hipError_t launch_lm_verdict_step(const DeviceProblem& dp, const SolverDev& sv, double* cost2, double* ctl, const LmRules& rules, rsba_iteration* trace, int trace_cap, hipStream_t st, bool cost_reduced = false);   // launch_cost_reduce (unless cost_reduced: done, the priors' cost added) + launch_pack_trial + launch_lm_decide_step
hipError_t launch_lm_linearize_gradient(const DeviceProblem& dp, const SolverDev& sv, const double* cost2, hipStream_t st);   // launch_local_linearize + the per-workgroup maxima of launch_gradient_max
hipError_t launch_lm_verdict_gradient(const DeviceProblem& dp, const SolverDev& sv, double* ctl, const LmRules& rules, rsba_iteration* trace, int trace_cap, double* snapshot, double seq, hipStream_t st, bool gradmax_done = false, int extra_partials = 0);   // extra_partials: maxima behind the coordinates' own in sv.partial (the per-pose priors')   // the maxima reduced (unless gradmax_done: scalars[kGradMax] is there) + launch_lm_decide_gradient; ctl copied to `snapshot` (device-visible host memory), stamped `seq` last
// motion priors (kernels_prior.hip): U_f, g_f += their J^T J / J^T r, ucross[f] = the (f, f-1) block; model cost change
hipError_t launch_prior_blocks(const DeviceProblem& dp, const SolverDev& sv, double* ucross, hipStream_t st);
hipError_t launch_prior_model(const DeviceProblem& dp, const SolverDev& sv, double* model_cost_change, double ratio_step, hipStream_t st, const double* ratio_step_ptr = nullptr);
// free interFrameRatio: its column of the normal equations (border [F*12], hg = {h, g}), dots and the combined step
hipError_t launch_prior_border(const DeviceProblem& dp, const SolverDev& sv, double* border, double* hg, hipStream_t st);
hipError_t launch_exchange_pack(const SolverDev& sv, const int32_t* slots, int ntiles, double* buf, bool unpack, hipStream_t st);   // exchange (2): the structurally non-zero tiles of S | rhs <-> one contiguous buffer
hipError_t launch_border_dots(const double* b, const double* u, const double* v, int64_t n, double* out2, hipStream_t st);
hipError_t launch_border_combine(double* y, const double* u, const double* v, double c, int64_t n, hipStream_t st, const double* c_ptr = nullptr);   // y = u - c v
hipError_t launch_ratio_init(double* rt, double ratio, double scale, double lb, hipStream_t st);               // device-side trust region: the ratio joins the state on the device
hipError_t launch_ratio_prepare(double* rt, double diag_term, double gs, double scale, hipStream_t st);         // what the ETA task of the factorisation reads (RatioSlot), by value ...
hipError_t launch_ratio_prepare_ctl(double* rt, const double* ctl, double lo, double hi, hipStream_t st);     // ... or from the trust-region state on the device
hipError_t launch_ratio_candidate(double* rt, const double* ctl, hipStream_t st);                              // ratio_new = max(lb, ratio - s eta)
hipError_t launch_intr_blocks(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);
hipError_t launch_virtual_records(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);
hipError_t launch_pack_linearize(const DeviceProblem& dp, const SolverDev& sv, const double* cost2, hipStream_t st, int nslots = 0);   // nslots: zeroed slots behind the payload (the ranks' gradient maxima)
hipError_t launch_gradient_max_points(const DeviceProblem& dp, const SolverDev& sv, int rank, hipStream_t st);     // -> xbuf[2n + 3 + rank]
hipError_t launch_gradient_max_cameras(const DeviceProblem& dp, const SolverDev& sv, int world, hipStream_t st);  // -> scalars[kGradMax], with the ranks' slots
hipError_t launch_unpack_linearize(const DeviceProblem& dp, const SolverDev& sv, hipStream_t st);
hipError_t launch_begin_solve(const SolverDev& sv, hipStream_t st);   // the two failure flags of a linear solve cleared in one launch
hipError_t launch_local_linearize(const DeviceProblem& dp, const SolverDev& sv, const double* cost2, hipStream_t st);   // pack + unpack of a single rank in one launch
hipError_t launch_pack_trial(const DeviceProblem& dp, const SolverDev& sv, const double* cost2, hipStream_t st);
hipError_t launch_own_points(const DeviceProblem& dp, const SolverDev& sv, double* buf4m, hipStream_t st);   // sharded solve: [M][3] owned values | [M] owner flag
hipError_t launch_merge_points(const DeviceProblem& dp, const double* buf4m, hipStream_t st);
hipError_t launch_unscaled_gradient(const DeviceProblem& dp, const SolverDev& sv, double* g_pose, double* g_point, hipStream_t st);

}  // namespace rsba
