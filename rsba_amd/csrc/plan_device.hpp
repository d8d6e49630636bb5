// The symbolic phase's passes over observations, points and entries on the device (plan_device.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "solver_state.hpp"

namespace rsba {

struct DevicePlanIn {
  const int32_t* obs_frame;      // [N] device, frame-major
  const int32_t* obs_point;      // [N] device
  int64_t N;
  int M, FR, NPF, NIB, FT, CD, nt;
  const uint8_t* tile_factored;  // [nt] device, may be null (no tile stores its groups factored)
  const uint32_t* struct_keys;   // HOST: keys I * nt + J of the tile pairs that exist whatever the points say (duplicates allowed)
  int64_t num_struct_keys;
  int64_t chunk_block;           // > 0: the chunk numbering by blocks of so many points wants the SEGMENTS of the entry list — maximal runs of one tile pair inside one block
  bool want_slot_frame_host;
  hipStream_t stream;
};

struct DevicePlanOut {
  std::vector<void*> owned;      // every device array below (the caller's to free)
  int64_t* point_ptr = nullptr; int32_t* obs_slot = nullptr; int32_t* slot_frame = nullptr; int32_t* slot_point = nullptr; uint32_t* slot_gpos = nullptr;
  int64_t nvgroups = 0; int32_t* vgroup_point = nullptr; int32_t* vgroup_intr = nullptr; int64_t* point_vgroup = nullptr;
  int64_t ngroups = 0, group_doubles = 0, factored_groups = 0;
  int64_t nent = 0, products = 0; uint32_t* ent_groups = nullptr; int32_t* ent_pt = nullptr; uint16_t* ent_mask = nullptr;
  // host copies of what the rest of the plan is built from
  std::vector<int64_t> point_ptr_h, tp_ptr;
  std::vector<int32_t> tp_I, tp_J, slot_frame_h;
  std::vector<int32_t> seg_pair, seg_block; std::vector<int64_t> seg_start;   // (chunk_block > 0) in entry order: pair-major, blocks ascending inside a pair
};

hipError_t device_plan_lists(const DevicePlanIn& in, DevicePlanOut* out);

}  // namespace rsba
