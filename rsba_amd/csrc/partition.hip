// rsba_partition_points (include/rsba_amd.h): which rank of a sharded solve owns which point — host only, no device.
// The reference is single-process (/root/reference/src/rsba/CeresHandler.h:394-426); the partition is the cut SURVEY §8e derives for
// the path — by point, cameras replicated — placed so that the reduced camera system can be FACTORED where it is formed: along the
// top separators of the nested dissection of its tile graph (tile_order.hpp).
#include <algorithm>
#include <cstring>
#include <vector>

#include "handle.hpp"
#include "solver_state.hpp"
#include "tile_order.hpp"

using namespace rsba;

extern "C" int32_t rsba_partition_points(const rsba_problem_desc* d, int32_t world, int32_t* owner, int32_t* num_top_tiles) {
  if (!d || !owner || world < 1) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "bad partition arguments");
  const int P = d->poses_per_frame;
  if (P != 1 && P != 2) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "poses_per_frame must be 1 or 2");
  const int FR = d->num_frames, M = d->num_points, CD = 6 * P, FT = kTile / CD;
  const int NIB = d->calibrated ? 0 : d->num_intrinsics, NPF = d->calibrated ? 0 : (9 + CD - 1) / CD;
  const int F = FR + NIB * NPF, nt = (F + FT - 1) / FT;
  const int64_t N = d->num_observations;
  if (num_top_tiles) *num_top_tiles = 0;
  if (world == 1 || N == 0) { std::fill(owner, owner + M, 0); return RSBA_OK; }
  for (int64_t i = 0; i < N; ++i)
    if (d->obs_frame[i] < 0 || d->obs_frame[i] >= FR || d->obs_point[i] < 0 || d->obs_point[i] >= M) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "observation index out of range");
  auto intr_of = [&](int f) { return (NIB > 1 && d->frame_intrinsics) ? d->frame_intrinsics[f] : 0; };
  // the tiles every point is seen in (real frames and, with intrinsics as parameter blocks, the pseudo frames of the blocks it is seen through)
  std::vector<int64_t> ptr((size_t)M + 1, 0);
  for (int64_t i = 0; i < N; ++i) ptr[d->obs_point[i] + 1]++;
  for (int j = 0; j < M; ++j) ptr[j + 1] += ptr[j];
  std::vector<int32_t> fr((size_t)N);
  { std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1); for (int64_t i = 0; i < N; ++i) fr[fill[d->obs_point[i]]++] = d->obs_frame[i]; }
  std::vector<uint8_t> pair((size_t)nt * nt, 0);
  std::vector<double> weight(nt, 0.0);
  for (int64_t i = 0; i < N; ++i) weight[d->obs_frame[i] / FT] += 1.0;
  std::vector<int32_t> tiles;
  for (int j = 0; j < M; ++j) {
    tiles.clear();
    for (int64_t x = ptr[j]; x < ptr[j + 1]; ++x) {
      tiles.push_back(fr[x] / FT);
      for (int v = 0; v < NPF; ++v) tiles.push_back((FR + intr_of(fr[x]) * NPF + v) / FT);
    }
    std::sort(tiles.begin(), tiles.end()); tiles.erase(std::unique(tiles.begin(), tiles.end()), tiles.end());
    for (size_t a = 0; a < tiles.size(); ++a) for (size_t b = 0; b < a; ++b) pair[(size_t)tiles[a] * nt + tiles[b]] = 1;
  }
  for (int f = 0; f < FR && NIB > 0; ++f) for (int v = 0; v < NPF; ++v) { const int a = (FR + intr_of(f) * NPF + v) / FT, b = f / FT; if (a != b) pair[(size_t)std::max(a, b) * nt + std::min(a, b)] = 1; }
  for (int c = 0; c < NIB; ++c) for (int v = 0; v < NPF; ++v) for (int w = 0; w < v; ++w) { const int a = (FR + c * NPF + v) / FT, b = (FR + c * NPF + w) / FT; if (a != b) pair[(size_t)std::max(a, b) * nt + std::min(a, b)] = 1; }
  std::vector<std::vector<int32_t>> adj(nt);
  for (int a = 0; a < nt; ++a) for (int b = 0; b < a; ++b) if (pair[(size_t)a * nt + b]) { adj[a].push_back(b); adj[b].push_back(a); }
  for (auto& l : adj) std::sort(l.begin(), l.end());
  const TileOrder ord = nested_dissection(nt, adj, plan_leaf_size(world, nt), world, &weight);
  if (!ord.parts_ok) return rsba_set_error(RSBA_ERR_UNSUPPORTED, "the co-visibility graph cannot be cut into that many parts (too few frames, or not connected)");
  int ntop = 0;
  for (int t = 0; t < nt; ++t) ntop += ord.part_of[t] < 0;
  if (num_top_tiles) *num_top_tiles = ntop;
  // a point belongs to the part of any of its tiles that has one; points seen in separator tiles only touch the tiles every rank
  // shares, so any rank may own them: they go, in point order, to whoever holds the fewest observations so far
  std::vector<int64_t> load(world, 0);
  for (int j = 0; j < M; ++j) {
    int own = -1;
    for (int64_t x = ptr[j]; x < ptr[j + 1]; ++x) {
      const int p = ord.part_of[fr[x] / FT];
      if (p < 0) continue;
      if (own >= 0 && own != p) return rsba_set_error(RSBA_ERR_HIP, "internal: a point is seen on both sides of a separator");
      own = p;
    }
    owner[j] = own;
    if (own >= 0) load[own] += ptr[j + 1] - ptr[j];
  }
  for (int j = 0; j < M; ++j) if (owner[j] < 0) {
    const int own = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    owner[j] = own; load[own] += ptr[j + 1] - ptr[j];
  }
  return RSBA_OK;
}
