// The symbolic phase's passes over observations, points and (point, tile pair) entries ON THE DEVICE (round 5).
//
// What Ceres does in its preprocessor for /root/reference/src/rsba/CeresHandler.h:419 (block structure detection, the Schur ordering's
// e-block lists) and what this solver did on host threads until round 4 (solver.hip: 11.6 of the 14.2 ms symbolic
// phase of a 1k-camera problem, 75 of 84 ms at 4k cameras — and the reference builds a fresh problem per BA() call,
// VideoSfMHandler.cc:586-592): the counting sort of the observations by point, the (point, frame tile) groups of P records, and the
// entry list of the point elimination sorted by tile pair.  The lists come out exactly as from the host passes (same order, same
// offsets: tests/test_gpu_solve.py::test_the_symbolic_phase_on_the_device_builds_the_same_plan) — every ordering here is decided by
// stable sorts and prefix sums, never by the arrival order of an atomic.  Nested dissection, the symbolic factorisation and the task
// lists stay on the host: they are O(tiles).
//
// Building blocks: an exclusive scan (tiles of 2 048, two levels) and a stable LSD radix sort of (key, value) pairs by 4-bit digits
// (per-workgroup histograms, one scan, a scatter that ranks by wave ballots) — hand-written: HBM-bound integer work, no library.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "devmem.hpp"
#include "plan_device.hpp"

namespace rsba {

namespace {

constexpr int kScanTile = 2048;   // 256 threads x 8
constexpr int kSortTile = 4096;   // 256 threads x 16 chunks

#define PD_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

// ---- exclusive scan --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scan_tile_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ sums, int64_t n) {
  __shared__ uint32_t s_wave[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * 8;
  uint32_t v[8], run = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? in[base + k] : 0u; run += v[k]; }
  uint32_t inc = run;   // inclusive scan of the threads' sums inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t before = inc - run;
  for (int w = 0; w < wave; ++w) before += s_wave[w];
#pragma unroll
  for (int k = 0; k < 8; ++k) { if (base + k < n) out[base + k] = before; before += v[k]; }
  if (tid == 255) sums[blockIdx.x] = before;
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t* sums, int nt, uint32_t* total) {   // one workgroup: exclusive scan in place
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nt; b0 += 1024) {
    const uint32_t v = b0 + tid < nt ? sums[b0 + tid] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = s_carry + inc - v;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (b0 + tid < nt) sums[b0 + tid] = before;
    __syncthreads();
    if (tid == 1023) s_carry = before + v;
    __syncthreads();
  }
  if (tid == 0 && total) *total = s_carry;
}
__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ sums, int64_t n, uint32_t* tail) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 8;
  const uint32_t add = sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 8; ++k) if (base + k < n) out[base + k] += add;
  (void)tail;
}
// out[0 .. n) = exclusive scan of in, *total (device, may be null) = the sum; scratch: ceil(n / kScanTile) uint32
hipError_t scan_exclusive(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* scratch, uint32_t* total, hipStream_t st) {
  if (n <= 0) { if (total) return hipMemsetAsync(total, 0, sizeof(uint32_t), st); return hipSuccess; }
  const int nt = (int)((n + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(scan_tile_kernel, dim3(nt), dim3(256), 0, st, in, out, scratch, n);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, scratch, nt, total);
  hipLaunchKernelGGL(scan_add_kernel, dim3(nt), dim3(256), 0, st, out, scratch, n, (uint32_t*)nullptr);
  return hipGetLastError();
}
inline size_t scan_scratch(int64_t n) { return (size_t)((n + kScanTile - 1) / kScanTile) + 1; }

// ---- stable radix sort of (key, value) pairs, 4 bits per pass ---------------------------------------------------------
__global__ __launch_bounds__(256) void sort_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ hist, int nblocks) {
  __shared__ uint32_t s_h[16];
  if (threadIdx.x < 16) s_h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
  for (int c = 0; c < kSortTile / 256; ++c) {
    const int64_t i = base + c * 256 + threadIdx.x;
    if (i < n) atomicAdd(&s_h[(keys[i] >> shift) & 15u], 1u);   // (integer counts: the order of the additions does not matter)
  }
  __syncthreads();
  if (threadIdx.x < 16) hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_h[threadIdx.x];
}
__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                           int64_t n, int shift, const uint32_t* __restrict__ offs, int nblocks) {
  __shared__ uint32_t s_base[16];
  __shared__ uint32_t s_cnt[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 16) s_base[tid] = offs[(size_t)tid * nblocks + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int c = 0; c < kSortTile / 256; ++c) {
    const int64_t i = base + c * 256 + tid;
    const bool live = i < n;
    const uint32_t k = live ? keys[i] : 0u, v = live ? vals[i] : 0u;
    const unsigned d = live ? (k >> shift) & 15u : 16u;
    unsigned rank = 0, mine = 0;
#pragma unroll
    for (unsigned b = 0; b < 16; ++b) {
      const unsigned long long m = __ballot(d == b);
      if (d == b) rank = (unsigned)__builtin_popcountll(m & below);
      if (lane == (int)b) mine = (unsigned)__builtin_popcountll(m);   // lane b keeps the wave's count of digit b
    }
    if (lane < 16) s_cnt[wave][lane] = mine;
    __syncthreads();
    if (live) {
      uint32_t at = s_base[d] + rank;
      for (int w = 0; w < wave; ++w) at += s_cnt[w][d];
      keys_out[at] = k; vals_out[at] = v;
    }
    __syncthreads();
    if (tid < 16) s_base[tid] += s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
    __syncthreads();
  }
}
// sorts by the low `bits` bits of the keys; the result ends up in (*keys, *vals) — the pointers may be swapped with the tmp pair
hipError_t sort_pairs(uint32_t** keys, uint32_t** vals, uint32_t** keys_tmp, uint32_t** vals_tmp, int64_t n, int bits, uint32_t* hist, uint32_t* scratch, hipStream_t st) {
  if (n <= 1) return hipSuccess;
  const int nblocks = (int)((n + kSortTile - 1) / kSortTile);
  for (int shift = 0; shift < bits; shift += 4) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(nblocks), dim3(256), 0, st, *keys, n, shift, hist, nblocks);
    PD_TRY(scan_exclusive(hist, hist, (int64_t)16 * nblocks, scratch, nullptr, st));
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblocks), dim3(256), 0, st, *keys, *vals, *keys_tmp, *vals_tmp, n, shift, hist, nblocks);
    std::swap(*keys, *keys_tmp); std::swap(*vals, *vals_tmp);
  }
  return hipGetLastError();
}

// ---- the plan's passes ------------------------------------------------------------------------------------------------
struct Geo { int FR, NPF, FT, nt, M, virt; int64_t N; };   // virt: every observed point owns NPF virtual slots (one shared intrinsics block)

__global__ void count_keys_kernel(const int32_t* __restrict__ key, int64_t n, uint32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) atomicAdd(&cnt[key[i]], 1u);
}
__global__ void count_keys_u32_kernel(const uint32_t* __restrict__ key, int64_t n, uint32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) atomicAdd(&cnt[key[i]], 1u);
}
__global__ void iota_keys_kernel(const int32_t* __restrict__ op, int64_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { keys[i] = (uint32_t)op[i]; vals[i] = (uint32_t)i; }
}
__global__ void widen_kernel(const uint32_t* __restrict__ in, int64_t n, const uint32_t* __restrict__ total, int64_t* __restrict__ out) {   // out[0 .. n] (the total behind the prefix)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i]; else if (i == n) out[n] = *total;
}
// slot k of the point-major order holds observation vals[k]
__global__ void slots_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, const int32_t* __restrict__ of, int64_t n,
                             int32_t* __restrict__ obs_slot, int32_t* __restrict__ slot_frame, int32_t* __restrict__ slot_point) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const uint32_t i = vals[k];
  obs_slot[i] = (int32_t)k; slot_frame[k] = of[i]; slot_point[k] = (int32_t)keys[k];
}
__global__ void observed_kernel(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ total, int M, uint32_t* __restrict__ flag) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < M) flag[j] = ((j + 1 < M ? ptr[j + 1] : *total) > ptr[j]) ? 1u : 0u;
}
__global__ void vgroups_kernel(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ vidx, Geo g, int32_t* __restrict__ vgroup_point, int32_t* __restrict__ vgroup_intr,
                               int64_t* __restrict__ point_vgroup, int32_t* __restrict__ slot_frame, int32_t* __restrict__ slot_point) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= g.M) return;
  if (!flag[j]) { point_vgroup[j] = -1; return; }
  const uint32_t v = vidx[j];
  vgroup_point[v] = j; vgroup_intr[v] = 0; point_vgroup[j] = (int64_t)v;
  for (int q = 0; q < g.NPF; ++q) { slot_frame[g.N + (int64_t)v * g.NPF + q] = g.FR + q; slot_point[g.N + (int64_t)v * g.NPF + q] = j; }
}

// One walk over a point's slots in ascending frame order (its virtual slots last): on_group(g, tile) for every new (tile, layer) group
// g = 0, 1, .. of the point, on_slot(g, pos, slot) — the host's `walk` (solver.hip), statement by statement.
template <class OnGroup, class OnSlot>
__device__ __forceinline__ int walk_point(const Geo& g, int64_t lo, int64_t hi, bool observed, uint32_t vg, const int32_t* __restrict__ slot_frame, OnGroup on_group, OnSlot on_slot) {
  int prev_frame = -1, layer = 0, cur_tile = -1, ng = 0, tile_first = 0;
  const int nv = (g.virt && observed) ? g.NPF : 0;
  for (int64_t x = lo; x < hi + nv; ++x) {
    const int64_t sl = x < hi ? x : g.N + (int64_t)vg * g.NPF + (x - hi);
    const int f = x < hi ? slot_frame[sl] : g.FR + (int)(x - hi);
    const int tile = f / g.FT, pos = f % g.FT;
    layer = (f == prev_frame) ? layer + 1 : 0; prev_frame = f;
    if (tile != cur_tile) { cur_tile = tile; tile_first = ng; }
    while (ng - tile_first <= layer) { on_group(ng, tile); ++ng; }
    on_slot(tile_first + layer, pos, sl);
  }
  return ng;
}
__device__ __forceinline__ int64_t ptr_at(const uint32_t* ptr, const uint32_t* total, int M, int j) { return j < M ? ptr[j] : *total; }

__global__ void group_count_kernel(Geo g, const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ total, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ vidx,
                                   const int32_t* __restrict__ slot_frame, const uint8_t* __restrict__ tile_factored, uint32_t* __restrict__ ngroups, uint32_t* __restrict__ gsize16) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= g.M) return;
  uint32_t s16 = 0;
  const int ng = walk_point(g, ptr[j], ptr_at(ptr, total, g.M, j + 1), flag[j] != 0, g.virt ? vidx[j] : 0u, slot_frame,
                            [&](int, int tile) { s16 += (tile_factored && tile_factored[tile]) ? kGroupFactored / 16 : kGroupFull / 16; }, [](int, int, int64_t) {});
  ngroups[j] = (uint32_t)ng; gsize16[j] = s16;
}
__global__ void group_fill_kernel(Geo g, int CD, const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ total, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ vidx,
                                  const int32_t* __restrict__ slot_frame, const uint8_t* __restrict__ tile_factored, const uint32_t* __restrict__ pt_group, const uint32_t* __restrict__ pt_goff16,
                                  int32_t* __restrict__ g_tile, uint32_t* __restrict__ g_off, uint32_t* __restrict__ slot_gpos, uint8_t* __restrict__ group_mask, uint8_t* __restrict__ group_present) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= g.M) return;
  const uint32_t base = pt_group[j];
  uint32_t at = pt_goff16[j] * 16u;
  walk_point(g, ptr[j], ptr_at(ptr, total, g.M, j + 1), flag[j] != 0, g.virt ? vidx[j] : 0u, slot_frame,
             [&](int gg, int tile) {
               const bool fac = tile_factored && tile_factored[tile];
               g_tile[base + gg] = tile; g_off[base + gg] = at; group_mask[base + gg] = 0; group_present[base + gg] = 0;
               at += fac ? kGroupFactored : kGroupFull;
             },
             [&](int gg, int pos, int64_t sl) {
               const uint32_t G = base + gg;
               const bool fac = tile_factored && tile_factored[g_tile[G]];
               slot_gpos[sl] = g_off[G] | ((uint32_t)pos << 1) | (fac ? 1u : 0u);
               group_present[G] = (uint8_t)(group_present[G] + 1);
               uint8_t m = group_mask[G];
               if (fac) m |= (uint8_t)(pos < 2 ? 0b011 : pos == 2 ? 0b111 : 0b100);
               else for (int row = pos * CD; row < (pos + 1) * CD; row += 4) m |= (uint8_t)(1u << (row / 16));
               group_mask[G] = m;
             });
}
// entries of point j: every pair of its tiles (X >= Y) times every combination of their layers — the host's for_each_entry
template <class Fn>
__device__ __forceinline__ void entries_of(uint32_t g0, uint32_t g1, const int32_t* __restrict__ g_tile, Fn fn) {
  for (uint32_t xa = g0; xa < g1;) {
    uint32_t xb = xa; while (xb < g1 && g_tile[xb] == g_tile[xa]) ++xb;
    for (uint32_t ya = g0; ya < xb;) {
      uint32_t yb = ya; while (yb < g1 && g_tile[yb] == g_tile[ya]) ++yb;
      for (uint32_t gx = xa; gx < xb; ++gx) for (uint32_t gy = ya; gy < yb; ++gy) fn(gx, gy);
      ya = yb;
    }
    xa = xb;
  }
}
__global__ void entry_count_kernel(int M, int nt, const uint32_t* __restrict__ pt_group, const uint32_t* __restrict__ ng_total, const int32_t* __restrict__ g_tile,
                                   uint32_t* __restrict__ nent, uint32_t* __restrict__ pair_flag, unsigned long long* __restrict__ total64) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  uint32_t c = 0;
  if (j < M) {
    entries_of(pt_group[j], j + 1 < M ? pt_group[j + 1] : *ng_total, g_tile, [&](uint32_t gx, uint32_t gy) { ++c; pair_flag[(size_t)g_tile[gx] * nt + g_tile[gy]] = 1u; });   // (everybody writes the same 1)
    nent[j] = c;
  }
  // the entries' total in 64 bits beside the 32-bit prefix sums (entries grow with the SQUARE of a point's tiles: a long-track scene can pass
  // 2^32 of them below the 2^32-doubles limit of the groups — the scan would wrap without a word).  An integer sum: order-free.
  unsigned long long w = c;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
  if ((threadIdx.x & 63) == 0 && w) atomicAdd(total64, w);
}
__global__ void mark_keys_kernel(const uint32_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[keys[i]] = 1u;
}
__global__ void pair_list_kernel(int nt, const uint32_t* __restrict__ pair_flag, const uint32_t* __restrict__ pair_idx, int32_t* __restrict__ tp_I, int32_t* __restrict__ tp_J) {
  const int64_t key = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (key >= (int64_t)nt * nt || !pair_flag[key]) return;
  tp_I[pair_idx[key]] = (int32_t)(key / nt); tp_J[pair_idx[key]] = (int32_t)(key % nt);
}
__global__ void entry_fill_kernel(int M, int nt, const uint32_t* __restrict__ pt_group, const uint32_t* __restrict__ ng_total, const int32_t* __restrict__ g_tile, const uint32_t* __restrict__ ent_off,
                                  const uint32_t* __restrict__ pair_idx, uint32_t* __restrict__ keyE, uint32_t* __restrict__ valE, uint32_t* __restrict__ gxE, uint32_t* __restrict__ gyE,
                                  int32_t* __restrict__ ptE) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  uint32_t e = ent_off[j];
  entries_of(pt_group[j], j + 1 < M ? pt_group[j + 1] : *ng_total, g_tile, [&](uint32_t gx, uint32_t gy) {
    keyE[e] = pair_idx[(size_t)g_tile[gx] * nt + g_tile[gy]]; valE[e] = e; gxE[e] = gx; gyE[e] = gy; ptE[e] = j | (gx == gy ? (int32_t)0x80000000 : 0);
    ++e;
  });
}
__global__ __launch_bounds__(256) void entry_gather_kernel(int64_t nent, const uint32_t* __restrict__ valS, const uint32_t* __restrict__ gxE, const uint32_t* __restrict__ gyE, const int32_t* __restrict__ ptE,
                                                           const int32_t* __restrict__ g_tile, const uint32_t* __restrict__ g_off, const uint8_t* __restrict__ tile_factored,
                                                           const uint8_t* __restrict__ group_mask, const uint8_t* __restrict__ group_present,
                                                           uint32_t* __restrict__ ent_groups, int32_t* __restrict__ ent_pt, uint16_t* __restrict__ ent_mask, unsigned long long* __restrict__ products) {
  __shared__ unsigned long long s_p[4];
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long prod = 0;
  if (w < nent) {
    const uint32_t e = valS[w], gx = gxE[e], gy = gyE[e];
    ent_groups[2 * w] = g_off[gx] | ((tile_factored && tile_factored[g_tile[gx]]) ? 1u : 0u);
    ent_groups[2 * w + 1] = g_off[gy] | ((tile_factored && tile_factored[g_tile[gy]]) ? 1u : 0u);
    ent_pt[w] = ptE[e];
    const unsigned ma = group_mask[gx], mb = group_mask[gy];
    unsigned pm = 0;
    for (int I = 0; I < 3; ++I) if ((ma >> I) & 1u) pm |= mb << (3 * I);
    ent_mask[w] = (uint16_t)pm;
    prod = (unsigned long long)group_present[gx] * group_present[gy];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) prod += __shfl_down(prod, off, 64);
  if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = prod;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(products, s_p[0] + s_p[1] + s_p[2] + s_p[3]);   // (a statistic: integer, order-free)
}

// tp_ptr[k] = first position whose key is >= k, for the keys k that begin at position w (pairs without entries take their successor's start)
__global__ void pair_bounds_kernel(int64_t nent, const uint32_t* __restrict__ keyS, int ntp, uint32_t* __restrict__ tp_ptr) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= nent) return;
  const uint32_t k = keyS[w], lo = w == 0 ? 0u : keyS[w - 1] + 1;
  if (w == 0 || keyS[w - 1] != k) for (uint32_t q = lo; q <= k; ++q) tp_ptr[q] = (uint32_t)w;
  if (w == nent - 1) for (uint32_t q = k + 1; q < (uint32_t)ntp; ++q) tp_ptr[q] = (uint32_t)nent;
}
// entry w (sorted by pair, points ascending inside a pair) opens a segment when its pair or its block of points differs from entry w - 1's
__global__ void segment_flag_kernel(int64_t nent, const uint32_t* __restrict__ keyS, const int32_t* __restrict__ ent_pt, int32_t block, uint32_t* __restrict__ flag) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= nent) return;
  flag[w] = (w == 0 || keyS[w] != keyS[w - 1] || (ent_pt[w] & 0x7fffffff) / block != (ent_pt[w - 1] & 0x7fffffff) / block) ? 1u : 0u;
}
__global__ void segment_list_kernel(int64_t nent, const uint32_t* __restrict__ keyS, const int32_t* __restrict__ ent_pt, int32_t block, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ idx,
                                    int32_t* __restrict__ seg_pair, int32_t* __restrict__ seg_block, uint32_t* __restrict__ seg_start) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= nent || !flag[w]) return;
  seg_pair[idx[w]] = (int32_t)keyS[w]; seg_block[idx[w]] = (ent_pt[w] & 0x7fffffff) / block; seg_start[idx[w]] = (uint32_t)w;
}

inline unsigned blocks256(int64_t n) { return (unsigned)std::max<int64_t>((n + 255) / 256, 1); }
inline int bits_for(uint64_t n) { int b = 1; while (b < 32 && (1ull << b) < n) ++b; return b; }

}  // namespace

// several temporaries out of ONE block (a 4k-camera plan has a dozen of 40 MB each: a hipMalloc and a synchronising hipFree apiece otherwise)
struct Arena {
  std::vector<std::pair<void**, size_t>> want;
  template <class T> void add(T** p, size_t count) { want.emplace_back(reinterpret_cast<void**>(p), ((std::max<size_t>(count, 1) * sizeof(T)) + 255) & ~(size_t)255); }
  hipError_t commit(std::vector<void*>& keep) {
    size_t total = 0;
    for (auto& w : want) total += w.second;
    void* base = nullptr;
    hipError_t e = dev_malloc(&base, total);
    if (e != hipSuccess) return e;
    keep.push_back(base);
    size_t at = 0;
    for (auto& w : want) { *w.first = static_cast<char*>(base) + at; at += w.second; }
    want.clear();
    return hipSuccess;
  }
};

#define PD_ALLOC(ptr, count) do { void* q_ = nullptr; PD_TRY(dev_malloc(&q_, std::max<size_t>((count), 1) * sizeof(*(ptr)))); (ptr) = static_cast<decltype(ptr)>(q_); } while (0)
#define PD_TMP(ptr, count) arena.add(&(ptr), (count))   // (allocated by the next arena.commit)
#define PD_KEEP(ptr, count) do { PD_ALLOC(ptr, count); out->owned.push_back(ptr); } while (0)

hipError_t device_plan_lists(const DevicePlanIn& in, DevicePlanOut* out) {
  hipStream_t st = in.stream;
  const int M = in.M, nt = in.nt;
  const int64_t N = in.N;
  Geo g{in.FR, in.NPF, in.FT, nt, M, in.NIB == 1 ? 1 : 0, N};
  const bool dbg = std::getenv("RSBA_DEBUG_PLAN") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = now(); char stages[256] = ""; int stages_n = 0;
  auto stage = [&](const char* name) { if (!dbg) return; (void)hipStreamSynchronize(st); const double t = now(); stages_n += std::snprintf(stages + stages_n, sizeof stages - stages_n, " %s %.2f;", name, t - t_last); t_last = t; };
  std::vector<void*> tmp;
  struct Cleanup { std::vector<void*>& t; hipStream_t st; ~Cleanup() { (void)hipStreamSynchronize(st); for (void* p : t) dev_free(p); } } cleanup{tmp, st};
  const int64_t nkeys = (int64_t)nt * nt;
  uint32_t *scratch = nullptr, *totals = nullptr;   // totals: [0] N, [1] observed points, [2] groups, [3] group doubles / 16, [4] entries, [5] tile pairs, [6..7] products (u64)
  const size_t scratch_n = scan_scratch(std::max<int64_t>({(int64_t)M + 1, nkeys, (int64_t)16 * ((std::max<int64_t>(N, 1) + kSortTile - 1) / kSortTile + 1)})) + 16;
  Arena arena;
  PD_TMP(scratch, scratch_n);
  PD_TMP(totals, 16);

  // ---- observations by point (stable): point_ptr, slots ----
  uint32_t *cnt = nullptr, *ptr32 = nullptr, *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr, *hist = nullptr;
  uint32_t *flag = nullptr, *vidx = nullptr, *ngr = nullptr, *gs16 = nullptr, *pt_group = nullptr, *pt_goff16 = nullptr, *nent_pt = nullptr, *ent_off = nullptr, *pair_flag = nullptr, *pair_idx = nullptr;
  PD_TMP(cnt, (size_t)M + 1); PD_TMP(ptr32, (size_t)M + 1);
  PD_TMP(keys, (size_t)N); PD_TMP(vals, (size_t)N); PD_TMP(keys2, (size_t)N); PD_TMP(vals2, (size_t)N);
  PD_TMP(hist, (size_t)16 * ((std::max<int64_t>(N, 1) + kSortTile - 1) / kSortTile + 1));
  PD_TMP(flag, (size_t)M + 1); PD_TMP(vidx, (size_t)M + 1);
  PD_TMP(ngr, (size_t)M + 1); PD_TMP(gs16, (size_t)M + 1); PD_TMP(pt_group, (size_t)M + 1); PD_TMP(pt_goff16, (size_t)M + 1);
  PD_TMP(nent_pt, (size_t)M + 1); PD_TMP(ent_off, (size_t)M + 1); PD_TMP(pair_flag, (size_t)nkeys); PD_TMP(pair_idx, (size_t)nkeys);
  PD_TRY(arena.commit(tmp));
  PD_TRY(hipMemsetAsync(totals, 0, 16 * sizeof(uint32_t), st));
  PD_TRY(hipMemsetAsync(cnt, 0, ((size_t)M + 1) * sizeof(uint32_t), st));
  if (N > 0) hipLaunchKernelGGL(count_keys_kernel, dim3(blocks256(N)), dim3(256), 0, st, in.obs_point, N, cnt);
  PD_TRY(scan_exclusive(cnt, ptr32, M, scratch, totals + 0, st));
  PD_KEEP(out->point_ptr, (size_t)M + 1);
  hipLaunchKernelGGL(widen_kernel, dim3(blocks256((int64_t)M + 1)), dim3(256), 0, st, ptr32, (int64_t)M, totals + 0, out->point_ptr);
  if (N > 0) hipLaunchKernelGGL(iota_keys_kernel, dim3(blocks256(N)), dim3(256), 0, st, in.obs_point, N, keys, vals);
  stage("count+scan");
  PD_TRY(sort_pairs(&keys, &vals, &keys2, &vals2, N, bits_for((uint64_t)std::max(M, 2)), hist, scratch, st));
  stage("sort by point");
  // virtual groups (one shared intrinsics block): one per observed point, NPF virtual slots each behind the real ones
  hipLaunchKernelGGL(observed_kernel, dim3(blocks256(M)), dim3(256), 0, st, ptr32, totals + 0, M, flag);
  PD_TRY(scan_exclusive(flag, vidx, M, scratch, totals + 1, st));
  uint32_t h_tot[8];
  PD_TRY(hipMemcpyAsync(h_tot, totals, sizeof h_tot, hipMemcpyDeviceToHost, st));
  PD_TRY(hipStreamSynchronize(st));
  const int64_t NVG = g.virt ? (int64_t)h_tot[1] : 0, NS = N + NVG * in.NPF;
  out->nvgroups = NVG;
  PD_KEEP(out->obs_slot, (size_t)N); PD_KEEP(out->slot_frame, (size_t)NS); PD_KEEP(out->slot_point, (size_t)NS); PD_KEEP(out->slot_gpos, (size_t)NS);
  if (N > 0) hipLaunchKernelGGL(slots_kernel, dim3(blocks256(N)), dim3(256), 0, st, keys, vals, in.obs_frame, N, out->obs_slot, out->slot_frame, out->slot_point);
  out->vgroup_point = nullptr; out->vgroup_intr = nullptr; out->point_vgroup = nullptr;
  if (g.virt) {
    PD_KEEP(out->vgroup_point, (size_t)NVG); PD_KEEP(out->vgroup_intr, (size_t)NVG); PD_KEEP(out->point_vgroup, (size_t)M);
    hipLaunchKernelGGL(vgroups_kernel, dim3(blocks256(M)), dim3(256), 0, st, flag, vidx, g, out->vgroup_point, out->vgroup_intr, out->point_vgroup, out->slot_frame, out->slot_point);
  }
  stage("slots");
  // ---- (point, tile, layer) groups ----
  hipLaunchKernelGGL(group_count_kernel, dim3(blocks256(M)), dim3(256), 0, st, g, ptr32, totals + 0, flag, vidx, out->slot_frame, in.tile_factored, ngr, gs16);
  PD_TRY(scan_exclusive(ngr, pt_group, M, scratch, totals + 2, st));
  PD_TRY(scan_exclusive(gs16, pt_goff16, M, scratch, totals + 3, st));
  PD_TRY(hipMemcpyAsync(h_tot, totals, sizeof h_tot, hipMemcpyDeviceToHost, st));
  PD_TRY(hipStreamSynchronize(st));
  const int64_t NG = h_tot[2];
  out->ngroups = NG; out->group_doubles = (int64_t)h_tot[3] * 16;
  int32_t* g_tile = nullptr; uint32_t* g_off = nullptr; uint8_t *gmask = nullptr, *gpresent = nullptr;
  PD_TMP(g_tile, (size_t)NG + 1); PD_TMP(g_off, (size_t)NG + 1); PD_TMP(gmask, (size_t)NG + 1); PD_TMP(gpresent, (size_t)NG + 1);
  PD_TRY(arena.commit(tmp));
  hipLaunchKernelGGL(group_fill_kernel, dim3(blocks256(M)), dim3(256), 0, st, g, in.CD, ptr32, totals + 0, flag, vidx, out->slot_frame, in.tile_factored, pt_group, pt_goff16,
                     g_tile, g_off, out->slot_gpos, gmask, gpresent);
  stage("groups");
  // ---- entries: count per point, which tile pairs exist, the point-major list, sorted by pair ----
  PD_TRY(hipMemsetAsync(pair_flag, 0, (size_t)nkeys * sizeof(uint32_t), st));
  if (in.num_struct_keys > 0) {   // the pairs that exist whatever the points say (diagonal, priors, intrinsics, other ranks'): their keys ride in the (not yet used) entry-offset array
    uint32_t* d_keys = nullptr;
    PD_ALLOC(d_keys, (size_t)in.num_struct_keys); tmp.push_back(d_keys);
    PD_TRY(hipMemcpyAsync(d_keys, in.struct_keys, (size_t)in.num_struct_keys * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mark_keys_kernel, dim3(blocks256(in.num_struct_keys)), dim3(256), 0, st, d_keys, in.num_struct_keys, pair_flag);
  }
  hipLaunchKernelGGL(entry_count_kernel, dim3(blocks256(M)), dim3(256), 0, st, M, nt, pt_group, totals + 2, g_tile, nent_pt, pair_flag, reinterpret_cast<unsigned long long*>(totals + 10));
  PD_TRY(scan_exclusive(nent_pt, ent_off, M, scratch, totals + 4, st));
  PD_TRY(scan_exclusive(pair_flag, pair_idx, nkeys, scratch, totals + 5, st));
  PD_TRY(hipMemcpyAsync(h_tot, totals, sizeof h_tot, hipMemcpyDeviceToHost, st));
  PD_TRY(hipStreamSynchronize(st));
  const int64_t nent = h_tot[4]; const int ntp = (int)h_tot[5];
  {
    unsigned long long nent64 = 0;
    PD_TRY(hipMemcpyAsync(&nent64, totals + 10, sizeof nent64, hipMemcpyDeviceToHost, st));
    PD_TRY(hipStreamSynchronize(st));
    if (nent64 != (unsigned long long)nent) return hipErrorInvalidValue;   // more than 2^32 - 1 entries: the 32-bit lists cannot hold them (the caller says so)
  }
  out->nent = nent;
  int32_t *d_tpI = nullptr, *d_tpJ = nullptr;
  PD_TMP(d_tpI, (size_t)ntp); PD_TMP(d_tpJ, (size_t)ntp);
  uint32_t *keyE = nullptr, *valE = nullptr, *keyE2 = nullptr, *valE2 = nullptr, *gxE = nullptr, *gyE = nullptr, *histE = nullptr, *tpcnt = nullptr, *tpptr32 = nullptr; int32_t* ptE = nullptr;
  PD_TMP(keyE, (size_t)nent); PD_TMP(valE, (size_t)nent); PD_TMP(keyE2, (size_t)nent); PD_TMP(valE2, (size_t)nent); PD_TMP(gxE, (size_t)nent); PD_TMP(gyE, (size_t)nent); PD_TMP(ptE, (size_t)nent);
  PD_TMP(histE, (size_t)16 * ((std::max<int64_t>(nent, 1) + kSortTile - 1) / kSortTile + 1));
  uint32_t* scratchE = nullptr;
  PD_TMP(scratchE, scan_scratch((int64_t)16 * ((std::max<int64_t>(nent, 1) + kSortTile - 1) / kSortTile + 1)) + 16);
  PD_TMP(tpcnt, (size_t)ntp + 1); PD_TMP(tpptr32, (size_t)ntp + 1);
  uint32_t *segflag = nullptr, *segidx = nullptr;
  if (in.chunk_block > 0) { PD_TMP(segflag, (size_t)nent + 1); PD_TMP(segidx, (size_t)nent + 1); }
  PD_TRY(arena.commit(tmp));
  hipLaunchKernelGGL(pair_list_kernel, dim3(blocks256(nkeys)), dim3(256), 0, st, nt, pair_flag, pair_idx, d_tpI, d_tpJ);
  hipLaunchKernelGGL(entry_fill_kernel, dim3(blocks256(M)), dim3(256), 0, st, M, nt, pt_group, totals + 2, g_tile, ent_off, pair_idx, keyE, valE, gxE, gyE, ptE);
  stage("entry list");
  PD_TRY(sort_pairs(&keyE, &valE, &keyE2, &valE2, nent, bits_for((uint64_t)std::max(ntp, 2)), histE, scratchE, st));
  // where every pair's entries start: read off the sorted keys (a histogram by atomics would hammer the few pairs every point has an
  // entry in — the diagonal pair of the intrinsics pseudo tile: half a million increments of one counter)
  PD_TRY(hipMemsetAsync(tpptr32, 0, ((size_t)ntp + 1) * sizeof(uint32_t), st));
  if (nent > 0) hipLaunchKernelGGL(pair_bounds_kernel, dim3(blocks256(nent)), dim3(256), 0, st, nent, keyE, ntp, tpptr32);
  stage("sort by pair");
  PD_KEEP(out->ent_groups, 2 * (size_t)nent); PD_KEEP(out->ent_pt, (size_t)nent); PD_KEEP(out->ent_mask, (size_t)nent);
  if (nent > 0) hipLaunchKernelGGL(entry_gather_kernel, dim3(blocks256(nent)), dim3(256), 0, st, nent, valE, gxE, gyE, ptE, g_tile, g_off, in.tile_factored, gmask, gpresent,
                                   out->ent_groups, out->ent_pt, out->ent_mask, reinterpret_cast<unsigned long long*>(totals + 6));
  // ---- the chunk numbering by blocks of points (large problems) only needs where, inside a pair's entries, the block changes ----
  int32_t *d_seg_pair = nullptr, *d_seg_block = nullptr; uint32_t* d_seg_start = nullptr; int64_t nseg = 0;
  if (in.chunk_block > 0 && nent > 0) {
    hipLaunchKernelGGL(segment_flag_kernel, dim3(blocks256(nent)), dim3(256), 0, st, nent, keyE, out->ent_pt, (int32_t)in.chunk_block, segflag);
    PD_TRY(scan_exclusive(segflag, segidx, nent, scratchE, totals + 9, st));
    uint32_t ns = 0;
    PD_TRY(hipMemcpyAsync(&ns, totals + 9, sizeof ns, hipMemcpyDeviceToHost, st));
    PD_TRY(hipStreamSynchronize(st));
    nseg = ns;
    PD_TMP(d_seg_pair, (size_t)nseg); PD_TMP(d_seg_block, (size_t)nseg); PD_TMP(d_seg_start, (size_t)nseg);
    PD_TRY(arena.commit(tmp));
    hipLaunchKernelGGL(segment_list_kernel, dim3(blocks256(nent)), dim3(256), 0, st, nent, keyE, out->ent_pt, (int32_t)in.chunk_block, segflag, segidx, d_seg_pair, d_seg_block, d_seg_start);
  }
  stage("gather");
  // ---- what the host goes on with: point_ptr, the tile pairs and their entry ranges, the statistics ----
  out->point_ptr_h.resize((size_t)M + 1); out->tp_I.resize((size_t)ntp); out->tp_J.resize((size_t)ntp); out->tp_ptr.assign((size_t)ntp + 1, 0);
  std::vector<uint32_t> tp32((size_t)ntp + 1, 0);
  PD_TRY(hipMemcpyAsync(out->point_ptr_h.data(), out->point_ptr, ((size_t)M + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  if (ntp > 0) {
    PD_TRY(hipMemcpyAsync(out->tp_I.data(), d_tpI, (size_t)ntp * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PD_TRY(hipMemcpyAsync(out->tp_J.data(), d_tpJ, (size_t)ntp * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PD_TRY(hipMemcpyAsync(tp32.data(), tpptr32, (size_t)ntp * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  }
  unsigned long long prod = 0;
  PD_TRY(hipMemcpyAsync(&prod, totals + 6, sizeof prod, hipMemcpyDeviceToHost, st));
  std::vector<uint32_t> seg32((size_t)nseg);
  if (nseg > 0) {
    out->seg_pair.resize((size_t)nseg); out->seg_block.resize((size_t)nseg);
    PD_TRY(hipMemcpyAsync(out->seg_pair.data(), d_seg_pair, (size_t)nseg * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PD_TRY(hipMemcpyAsync(out->seg_block.data(), d_seg_block, (size_t)nseg * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PD_TRY(hipMemcpyAsync(seg32.data(), d_seg_start, (size_t)nseg * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  }
  if (in.want_slot_frame_host) { out->slot_frame_h.resize((size_t)N); if (N > 0) PD_TRY(hipMemcpyAsync(out->slot_frame_h.data(), out->slot_frame, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, st)); }
  PD_TRY(hipStreamSynchronize(st));
  out->seg_start.assign(seg32.begin(), seg32.end());
  stage("to host");
  if (dbg) std::fprintf(stderr, "[rsba plan] device lists (ms):%s\n", stages);
  for (int t = 0; t < ntp; ++t) out->tp_ptr[t] = tp32[t];
  out->tp_ptr[ntp] = nent;
  out->products = (int64_t)prod;
  // the groups stored factored: from the two totals (every group is 144 or 80 doubles)
  out->factored_groups = (NG * (int64_t)kGroupFull - out->group_doubles) / (kGroupFull - kGroupFactored);
  return hipGetLastError();
}

}  // namespace rsba
