// Host side of the symbolic phase that several callers share: the fill-reducing, parallelism-exposing order of the 48 x 48 tile
// columns of the reduced camera system (what CHOLMOD's analyse step does for Ceres behind
// /root/reference/src/rsba/CeresHandler.h:403,419) and — multi-GPU — the partition of that order among the ranks.
//
// Nested dissection by BFS level structures (George): a video's co-visibility graph is a band, whose BFS levels are band-wide
// separators; cutting it into independent segments turns the factorisation's serial tile chain into a shallow elimination tree.
// Tiles adjacent to (almost) everything — the intrinsics border — are ordered last.
//
// With nparts > 1 the TOP of the tree is cut for the ranks of a sharded solve: ceil(log2 nparts) levels of separators — the frontier of
// a prefix of the level-by-level order, placed tile by tile where the loads balance — split the tiles into nparts "parts" of about
// equal weight (observations), one per rank; part_of[tile] = the rank whose subtree
// the tile belongs to, -1 for the tiles of those top separators (and the dense border).  No point is seen on both sides of a
// separator (its two sides are not adjacent in the tile graph), so every point that is seen in a part's tile at all belongs to
// that part alone: the rank that owns the part owns the point (every observation of a point on one rank), the columns
// of its part of S are complete on that rank without any exchange, and it can factor them on its own.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <vector>

namespace rsba {

// tiles per leaf of the dissection (RSBA_CHOL_LEAF: tuning aid).  The factorisation is a latency chain as long as the GPU has idle
// workgroups — smaller leaves shorten it at the price of fill — and a throughput problem beyond: on one GPU 8 tiles up to 64 tile
// columns (100 cameras: 0.285 -> 0.270 ms), 12 up to 500 (1k cameras: 0.758 -> 0.745), 24 above (4k cameras: 2.38 against 2.43 with
// 12).  A sharded factorisation (nparts ranks): 12 whatever the size — a rank's share of the tasks leaves most of its
// GPU idle: 0.756 -> 0.662 ms at 1k cameras on 8 ranks, 0.672 -> 0.647 on 2, 1.164 -> 1.131 at 4k cameras on 4
// (profiles/r04/shard_leaf_sweep.txt).  Every rank — and rsba_partition_points — must use the same.
inline int plan_leaf_size(int nparts, int nt) {
  int k = nparts > 1 ? 12 : nt <= 64 ? 8 : nt <= 500 ? 12 : 24;
  if (const char* e = std::getenv("RSBA_CHOL_LEAF")) k = std::max(2, std::atoi(e));
  return k;
}

struct TileOrder {
  std::vector<int32_t> perm;      // perm[new] = old tile
  std::vector<int32_t> leaf_of;   // per old tile: leaf of the dissection, -1 = a separator / the dense border
  std::vector<int32_t> part_of;   // per old tile: rank whose subtree holds it, -1 = top separator / dense border (all 0 when nparts == 1)
  int nleaf = 0;
  bool parts_ok = true;           // false: the graph could not be cut into nparts parts (disconnected or too short): no sharded factorisation
};

inline TileOrder nested_dissection(int nt, const std::vector<std::vector<int32_t>>& adj, int kLeaf, int nparts = 1, const std::vector<double>* weight = nullptr) {
  TileOrder out;
  out.perm.reserve(nt); out.leaf_of.assign(nt, -1); out.part_of.assign(nt, nparts > 1 ? -1 : 0);
  std::vector<uint8_t> dense(nt, 0);
  for (int t = 0; t < nt; ++t) if (nt > 8 && (int)adj[t].size() > (3 * nt) / 4) dense[t] = 1;
  std::vector<int32_t> tag(nt, -1);                   // membership of the subgraph being processed
  int tagc = 0, next_part = 0;
  auto bfs_levels = [&](int root, int mytag, std::vector<std::vector<int32_t>>& lv) {
    lv.clear(); std::vector<int32_t> cur{root}; std::vector<uint8_t> seen(nt, 0); seen[root] = 1;
    while (!cur.empty()) { lv.push_back(cur); std::vector<int32_t> nxt; for (int u : cur) for (int v : adj[u]) if (tag[v] == mytag && !seen[v]) { seen[v] = 1; nxt.push_back(v); } std::sort(nxt.begin(), nxt.end()); cur.swap(nxt); }
  };
  // the level structure of `nodes` from a pseudo-peripheral root; false when the nodes are not connected
  auto level_structure = [&](const std::vector<int32_t>& nodes, int mytag, std::vector<std::vector<int32_t>>& lv) {
    int root = nodes[0];
    for (int u : nodes) if (adj[u].size() < adj[root].size()) root = u;
    for (int it = 0; it < 2; ++it) { bfs_levels(root, mytag, lv); int best = lv.back()[0]; for (int u : lv.back()) if (adj[u].size() < adj[best].size()) best = u; root = best; }
    bfs_levels(root, mytag, lv);
    size_t reached = 0; for (auto& l : lv) reached += l.size();
    return reached == nodes.size();
  };
  std::function<void(std::vector<int32_t>)> nd = [&](std::vector<int32_t> nodes) {
    if (nodes.empty()) return;
    const int mytag = ++tagc;
    for (int u : nodes) tag[u] = mytag;
    std::vector<std::vector<int32_t>> lv;
    if (!level_structure(nodes, mytag, lv)) {     // disconnected: split off this component and recurse on both
      std::vector<uint8_t> in(nt, 0); std::vector<int32_t> comp, rest;
      for (auto& l : lv) for (int u : l) { in[u] = 1; comp.push_back(u); }
      for (int u : nodes) if (!in[u]) rest.push_back(u);
      nd(comp); nd(rest); return;
    }
    if ((int)nodes.size() <= kLeaf || lv.size() < 3) { for (auto& l : lv) for (int u : l) { out.perm.push_back(u); out.leaf_of[u] = out.nleaf; } ++out.nleaf; return; }
    size_t half = nodes.size() / 2, acc = 0, cut = 1;
    for (size_t l = 0; l < lv.size(); ++l) { acc += lv[l].size(); if (acc >= half) { cut = std::min(std::max<size_t>(l, 1), lv.size() - 2); break; } }
    std::vector<int32_t> left, right;
    for (size_t l = 0; l < cut; ++l) left.insert(left.end(), lv[l].begin(), lv[l].end());
    for (size_t l = cut + 1; l < lv.size(); ++l) right.insert(right.end(), lv[l].begin(), lv[l].end());
    const std::vector<int32_t> sep = lv[cut];
    nd(left); nd(right);
    for (int u : sep) out.perm.push_back(u);
  };
  // The top of the tree, cut for `parts` ranks.  The nodes are laid out level by level (Cuthill-McKee from a pseudo-peripheral root);
  // a cut at position p of that order leaves the first p nodes on the left, and the separator is their FRONTIER — the later nodes
  // that touch one of them — so every position is a candidate, not only the level boundaries (a level of a video's tile graph is as
  // wide as its band: six tiles of the thirty-one a rank gets at 1 000 cameras on 8 ranks).  A side's load is what its rank will
  // own: its tiles' weight, half of the separator's (a point seen in a separator tile belongs to whichever side it is also seen
  // on), and the halves of the outer separators (`outer`: node set, half weight) that lie against it.  The cut is where the two
  // sides' loads per rank are closest.
  struct OuterSep { std::vector<int32_t> nodes; double half; };
  std::function<void(std::vector<int32_t>, int, std::vector<OuterSep>)> nd_parts = [&](std::vector<int32_t> nodes, int parts, std::vector<OuterSep> outer) {
    if (parts <= 1 || nodes.empty()) {
      const int p = next_part++;
      for (int u : nodes) out.part_of[u] = p;
      nd(std::move(nodes));
      return;
    }
    const int mytag = ++tagc;
    for (int u : nodes) tag[u] = mytag;
    std::vector<std::vector<int32_t>> lv;
    if (!level_structure(nodes, mytag, lv) || lv.size() < 3) {   // cannot be cut: everything to one rank (the caller falls back to the replicated factorisation)
      out.parts_ok = false;
      const int p = next_part; next_part += parts;
      for (int u : nodes) out.part_of[u] = p;
      nd(std::move(nodes));
      return;
    }
    const int pl = parts / 2, pr = parts - pl;
    auto w = [&](int u) { return weight ? (*weight)[u] : 1.0; };
    const int n = (int)nodes.size();
    std::vector<int32_t> order; order.reserve(n);
    for (auto& l : lv) order.insert(order.end(), l.begin(), l.end());
    std::vector<int32_t> pos(nt, -1);
    for (int i = 0; i < n; ++i) pos[order[i]] = i;
    // Inside a level the nodes come by index, which is the wrong way round whenever the root sits at the high end (a frontier then
    // holds the rest of one level AND the next): a few barycentre sweeps — every node to the mean position of itself and its
    // neighbours, ties by the old position — turn the level order into a proper linear arrangement (for a band: the natural one).
    for (int sweep = 0; sweep < 4; ++sweep) {
      std::vector<std::pair<double, int32_t>> key(n);
      for (int i = 0; i < n; ++i) {
        double sum = i; int cnt = 1;
        for (int v : adj[order[i]]) if (tag[v] == mytag && pos[v] >= 0) { sum += pos[v]; ++cnt; }
        key[i] = {sum / cnt, (int32_t)i};
      }
      std::sort(key.begin(), key.end());
      std::vector<int32_t> next(n);
      for (int i = 0; i < n; ++i) next[i] = order[key[i].second];
      order.swap(next);
      for (int i = 0; i < n; ++i) pos[order[i]] = i;
    }
    // first_nb[i]: the earliest position of a neighbour of the node at position i; the node is in the frontier of every cut p with first_nb < p <= i
    std::vector<int32_t> first_nb(n);
    std::vector<double> sep_delta((size_t)n + 2, 0.0), prefix((size_t)n + 1, 0.0);
    for (int i = 0; i < n; ++i) {
      int f = n;
      for (int v : adj[order[i]]) if (tag[v] == mytag && pos[v] >= 0) f = std::min(f, (int)pos[v]);
      first_nb[i] = f;
      prefix[i + 1] = prefix[i] + w(order[i]);
      if (f < i) { sep_delta[f + 1] += w(order[i]); sep_delta[i + 1] -= w(order[i]); }
    }
    const double total = prefix[n];
    // where the outer separators lie: the median position of their neighbours among these nodes (none: they do not count here)
    std::vector<std::pair<int, double>> outer_at;   // (position, half weight)
    std::vector<int> outer_pos(outer.size(), -1);
    for (size_t q = 0; q < outer.size(); ++q) {
      std::vector<int32_t> at;
      for (int u : outer[q].nodes) for (int v : adj[u]) if (tag[v] == mytag && pos[v] >= 0) at.push_back(pos[v]);
      if (at.empty()) continue;
      std::nth_element(at.begin(), at.begin() + at.size() / 2, at.end());
      outer_pos[q] = at[at.size() / 2];
      outer_at.emplace_back(outer_pos[q], outer[q].half);
    }
    double outer_total = 0.0; for (auto& o : outer_at) outer_total += o.second;
    int cut = -1; double best = -1.0, ws = 0.0, ws_at_cut = 0.0;
    for (int p = 1; p < n; ++p) {
      ws += sep_delta[p];
      const double wl = prefix[p], wr = total - wl - ws;
      if (wr <= 0.0) break;                                    // nothing would be left on the right
      double bl = 0.0; for (auto& o : outer_at) if (o.first < p) bl += o.second;
      const double miss = std::abs((wl + 0.5 * ws + bl) / pl - (wr + 0.5 * ws + (outer_total - bl)) / pr);
      if (best < 0.0 || miss < best) { best = miss; cut = p; ws_at_cut = ws; }
    }
    if (cut < 0) {   // (a complete graph: no cut leaves two sides)
      out.parts_ok = false;
      const int p = next_part; next_part += parts;
      for (int u : nodes) out.part_of[u] = p;
      nd(std::move(nodes));
      return;
    }
    std::vector<int32_t> left(order.begin(), order.begin() + cut), right, sep;
    for (int i = cut; i < n; ++i) (first_nb[i] < cut ? sep : right).push_back(order[i]);
    std::vector<OuterSep> outer_l, outer_r;
    for (size_t q = 0; q < outer.size(); ++q) if (outer_pos[q] >= 0) (outer_pos[q] < cut ? outer_l : outer_r).push_back(outer[q]);
    outer_l.push_back(OuterSep{sep, 0.5 * ws_at_cut}); outer_r.push_back(OuterSep{sep, 0.5 * ws_at_cut});
    nd_parts(std::move(left), pl, std::move(outer_l)); nd_parts(std::move(right), pr, std::move(outer_r));
    for (int u : sep) out.perm.push_back(u);
  };
  std::vector<int32_t> sparse_nodes;
  for (int t = 0; t < nt; ++t) if (!dense[t]) sparse_nodes.push_back(t);
  for (int t = 0; t < nt; ++t) if (dense[t]) tag[t] = -2;   // dense tiles are invisible to the dissection
  if (nparts > 1) nd_parts(sparse_nodes, nparts, {}); else nd(sparse_nodes);
  for (int t = 0; t < nt; ++t) if (dense[t]) out.perm.push_back(t);
  if (nparts > 1 && next_part != nparts) out.parts_ok = false;
  return out;
}

}  // namespace rsba
