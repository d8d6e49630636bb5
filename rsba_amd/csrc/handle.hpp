// The object behind rsba_handle (include/rsba_amd.h): owns every device allocation of one problem.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../include/rsba_amd.h"
#include "device_state.hpp"

namespace rsba { struct Solver; }

struct rsba_handle {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  rsba_problem_desc desc;              // caller's descriptor (parameter pointers are written back by solve)
  rsba::DeviceProblem dp;
  std::vector<int64_t> order;          // internal (frame-major) index -> caller's observation index
  bool identity_order = true;
  std::vector<double> mask_pose, mask_point, mask_intr;   // 0 = fixed coordinate, 1 = free
  std::vector<double> mask_pose_caller;                  // mask_pose as rsba_create built it (rsba_set_global_shutter_frames holds more slots constant on top)
  double *d_mask_pose = nullptr, *d_mask_point = nullptr, *d_mask_intr = nullptr;   // device copies: the column scales are reset from them (no host traffic per solve)
  std::vector<int32_t> obs_frame, obs_point;            // host copies, internal (frame-major) order
  std::vector<int32_t> frame_intr;                       // [F] host copy of the frame -> intrinsics block map
  std::vector<void*> allocs;
  double* d_cost2 = nullptr;           // {cost, fixed cost}
  int64_t* d_order = nullptr;          // device copy of `order`, made on the first host-returning evaluation
  // scratch of the handle-level filter calls (rsba_validate_observations / rsba_reproject), kept between calls
  uint8_t* d_flags = nullptr; uint8_t* d_flags_out = nullptr; int64_t flags_cap = 0, flags_out_cap = 0;
  int32_t* d_pairs = nullptr; double* d_pair_xy = nullptr; int64_t pairs_cap = 0, pair_xy_cap = 0;
  double* d_rows = nullptr;            // [N][2 + 2K] results in the caller's layout and order, staged for one D2H copy each
  // motion priors (rsba_set_motion_priors): frames that carry one, device flags live in dp.prior_of
  std::vector<int32_t> prior_frames;
  bool prior_free = false;             // interFrameRatio is a free, lower-bounded parameter (rsba_set_inter_frame_ratio_free)
  double prior_ratio_result = 0.0;     // its value after the last solve
  int prior_invalid = 0;               // blocks whose functor returns false for the given interFrameRatio
  bool prior_split = false;            // sharded factorisation: dp.prior_of lists THIS rank's share of the priors (every rank contributes its own, not rank 0 all)
  const int32_t* prior_of_all = nullptr; int prior_invalid_all = 0;   // ... and what the handle had before the plan split them (the table lives in the PLAN's memory: rsba_destroy_solver puts these back)
  // per-pose priors (rsba_set_pose_priors)
  std::vector<int32_t> pp_blocks;      // pose blocks carrying a GoodPosePrior
  double* pp_host = nullptr;           // caller's priorPoses values [count][6], written back by rsba_solve
  rsba::Solver* solver = nullptr;      // normal-equation / Schur / LM state, built on first use
  // multi-GPU exchange (rsba_set_exchange / rsba_set_block_structure)
  rsba_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  int rank = 0, world = 1;
  int64_t x_calls[RSBA_NUM_EXCHANGES] = {}, x_doubles[RSBA_NUM_EXCHANGES] = {};   // collectives issued since the handle was created, by kind (rsba_get_exchange_stats)
  std::vector<uint8_t> union_mask;     // [F*F] structure installed by the host, empty = local structure
  std::vector<int64_t> frame_obs_total;// [F] global observation count per frame
};

// internal (not exported through the C header)
int32_t rsba_set_error(int32_t code, const char* msg);
int32_t rsba_gradient(rsba_handle* h, double* gradient_host);
void rsba_destroy_solver(rsba_handle* h);
void rsba_release_plan_scratch();   // solver.hip: the symbolic phase's host scratch (kept across handles)
