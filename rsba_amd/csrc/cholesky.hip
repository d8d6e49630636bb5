// K6  reduced camera system  S y = rhs  by a tile-sparse Cholesky factorisation, fp64.
//
// Replaces the CHOLMOD sparse Cholesky behind Ceres' SPARSE_SCHUR (reference call site
// /root/reference/src/rsba/CeresHandler.h:403,419).  S (lower triangle) is a grid of 48x48 tiles; the
// symbolic phase (host, once per problem) finds the tiles that are structurally non-zero after fill-in
// and only those are stored — packed back to back in HBM (tile slot s at S + s*48*48, row-major), which is
// also the buffer the multi-GPU exchange all-reduces.  rsba's problems are video:
// frames only share points with frames a few dozen positions away, so S is block-banded, fill stays
// inside the band, and the factorisation is O(n b^2) instead of O(n^3).
//
// A band factored column after column is a serial chain of ~n/48 dependent steps (250 at 1k cameras),
// each a few tens of microseconds of latency: the GPU idles.  So the host reorders the tile columns by
// nested dissection (BFS-level separators cut the band into independent segments), computes the levels of
// the resulting elimination structure, and the factorisation runs as a DAG of tile tasks in one persistent
// kernel (below); the forward solve rides along, the backward solve walks the tree back down in the same launch.
// The reduced system S itself is left untouched: the factor's sub-diagonal tiles go to their own array (Lf).
// 1k cameras: ~42 levels instead of 250 steps.
#include "solver_state.hpp"

namespace rsba {

namespace {

constexpr int T = kTile;
constexpr int TP = T + 1;   // LDS row pitch (doubles): odd pitch keeps column walks conflict-free

// 1/sqrt(x) to full fp64 accuracy: hardware estimate + two Newton steps (avoids the long fp64
// sqrt-then-divide dependency chain on the factorisation's critical path)
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

__device__ __forceinline__ double* tile_ptr(const SolverDev& sv, int slot) { return sv.S + (size_t)slot * (T * T); }

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Workgroup barrier that orders LDS traffic only.  Everything the waves of a workgroup hand each other here goes through
// LDS; what they write to HBM (write-once cells, partial tiles) is for OTHER workgroups, which find it by polling.  A full
// __syncthreads() would also wait for those global stores — and for every prefetch still in flight — to drain (vmcnt(0)).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// (accessors of the write-once cells in HBM; the scheme itself is described with the task drivers below)
#define RSBA_GLOBAL __attribute__((address_space(1)))
template <class V> __device__ __forceinline__ V gl(const V* p) { return *(const RSBA_GLOBAL V*)p; }
template <bool DAG> __device__ __forceinline__ double ld(const double* p) {
  if (DAG) return __hip_atomic_load((const RSBA_GLOBAL double*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return gl(p);
}
template <bool DAG> __device__ __forceinline__ void st(double* p, double v) {
  if (DAG) __hip_atomic_store((RSBA_GLOBAL double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *(RSBA_GLOBAL double*)p = v;
}

constexpr int NB = 16;   // panel width of the tile factorisation: one MFMA block column
typedef double dbl4_t __attribute__((ext_vector_type(4)));

// (Round-1 form of the tile factorisation; the tasks now use factor_invert_tile below.  Kept as the baseline that
// tools/tile_factor_bench.hip times and checks it against.)
// Factor the T x T tile held in LDS (lower triangle, pitch TP) in place: blocked right-looking Cholesky.
// Each 16-column panel is factored by the first wave entirely in registers (lane = row, pivot rows
// broadcast with v_readlane, 1/sqrt instead of sqrt + divide: no LDS round trips or barriers on the
// serial chain); the rank-16 trailing update runs on the matrix pipe, its blocks dealt to the four waves.
// All threads must call.
// Returns false (in the first wave) on a non-positive pivot.
__device__ __forceinline__ bool potrf_blocked(double* A, int tid) {
  bool ok = true;
#pragma unroll
  for (int bc = 0; bc < T / NB; ++bc) {
    const int c0 = NB * bc;
    if (tid < 64) {
      const int row = c0 + tid;
      const bool live = row < T;
      const int rr = live ? row : T - 1;
      double a[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) a[m] = A[rr * TP + c0 + m];
      // right-looking inside the panel: once column jj is scaled, the columns behind it are updated by independent
      // FMAs (the next pivot first), which fill the latency of the next 1/sqrt instead of forming one long sum
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        const double djj = readlane_f64(a[jj], jj);
        ok = ok && (djj > 0.0) && isfinite(djj);
        // the column is broadcast BEFORE it is scaled: the lane reads do not wait for the 1/sqrt chain, only the FMAs do
        double u[NB];
#pragma unroll
        for (int m = jj + 1; m < NB; ++m) u[m] = readlane_f64(a[jj], m);
        const double rinv = rsqrt_nr(djj);
        const double t = a[jj] * (rinv * rinv);                 // a_r / d
#pragma unroll
        for (int m = jj + 1; m < NB; ++m) a[m] -= t * u[m];      // a_rm -= a_r a_m / d
        a[jj] = (tid == jj) ? djj * rinv : a[jj] * rinv;
      }
      if (live) {
#pragma unroll
        for (int m = 0; m < NB; ++m) if (c0 + m <= row) A[row * TP + c0 + m] = a[m];
      }
    }
    lds_barrier();
    // rank-16 trailing update A[r][c] -= sum_m L[r][c0 + m] L[c][c0 + m] for c0 + 12 <= c <= r on the matrix pipe
    // (K = 16 = four MFMA steps): the 16 x 16 blocks of the tile grid that reach into the trailing part are
    // dealt to the waves, products formed in full and subtracted under a mask.
    if (c0 + NB < T) {
      const int wave = tid >> 6, lane = tid & 63, mi = lane & 15, mg = lane >> 4;
      const int first = (c0 + NB) >> 4;                        // first block row / column that has trailing entries
      int blk = 0;
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
          if (J < first) continue;                             // I >= J >= first
          if ((blk++ & 3) != wave) continue;
          dbl4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < NB / 4; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(16 * I + mi) * TP + c0 + 4 * kk + mg], A[(16 * J + mi) * TP + c0 + 4 * kk + mg], acc, 0, 0, 0);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int r = 16 * I + mg + 4 * v, c = 16 * J + mi;
            if (c >= c0 + NB && c <= r) A[r * TP + c] -= acc[v];
          }
        }
      lds_barrier();
    }
  }
  return ok;
}

// LDS map (doubles): four partial-tile buffers, then small vectors
constexpr int kBuf = T * TP;
constexpr int kVecOff = 4 * kBuf;            // [4][T] per-wave rhs partials, [T] b, [T] unused, [T] unused, [T] z of the last contributor
constexpr int kCholLds = kVecOff + 8 * T;    // 78 KB: two workgroups fit a CU
// Nothing else is allocated: what a DIAG task needs beyond the four tile buffers lives in parts of them that are idle at the time —
//   X (tile (j, k*) less its updates) in buffer 1 and L = X W^T in buffer 3 while the last contributor is applied (the
//   factorisation, which uses buffers 1 .. 3 as W, T and L-panel scratch, has not started), and
//   the pivot messages between the two waves of the tile factorisation in rows 0 .. 15 of buffers 2 and 3, which the
//   factorisation never touches (its scratch blocks sit in rows 16 .. 47): pivots 0 .. 7 in buffer 2, 8 .. 15 in buffer 3.
constexpr int kXBuf = 1 * kBuf, kLBuf = 3 * kBuf;
constexpr int kMsg = 64;                     // one pivot's message: a double per lane
__device__ __forceinline__ constexpr int msg_off(int jj) { return (jj < 8 ? 2 * kBuf : 3 * kBuf) + (jj & 7) * kMsg; }
static_assert(8 * kMsg <= 16 * TP, "the messages of eight pivots must fit the sixteen idle rows of a scratch tile");

// "empty" bit pattern of the write-once cells (below) — also of the pivot messages in LDS
__device__ __forceinline__ bool filled(double v) { return __double_as_longlong(v) != -1ll; }

// ---- factorisation + inverse of one 48 x 48 tile on the matrix pipe -------------------------------------------------
// The serial part of the whole solve is the chain of diagonal tiles: W_j = chol(D_j)^-1 for one tile after the other.  A
// lane-per-row Cholesky pays ~30 cross-lane broadcasts (v_readlane) per pivot.  Here a pivot is ONE rank-1 MFMA instead:
// the 16 x 16 diagonal block lives in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane (c, g): rows g + 4v, column
// c), kept in full (symmetric), so row jj — the multipliers of pivot jj, by symmetry — already sits in lane group
// g = jj % 4, register jj / 4, one column per lane: exactly where the instruction wants the k = g slice of its A and B
// operands.  D -= (row / d) row^T is then a select and a multiply per lane, no data movement at all.  The same multipliers
// applied to an identity give L'^-1 of the unit-lower LDL^T factor (second MFMA of the pivot), and
// W = chol(D)^-1 = diag(d)^-1/2 L'^-1: no square root on the chain, and the factor itself is never formed (nothing
// downstream reads it).  The reciprocal of a pivot (v_rcp + two Newton steps) is taken one pivot ahead.
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}

// fp64 MFMAs and fp64 vector instructions do not overlap on this chip (tools/ldl_probe.hip: their times add up, the matrix
// pipe runs on the vector unit's fp64 datapath), so whatever one wave does per pivot is serial.  The step is therefore
// split over TWO waves (two SIMDs): the first eliminates (one MFMA, the pivot's reciprocal by v_rcp + y0 (1 + e + e^2),
// ~180 cycles per pivot) and posts each pivot's multipliers and the pivot itself in LDS; the second picks a message up as
// soon as it is complete (the data is its own flag, as with the write-once cells in HBM), applies it to the identity and
// ends up with the inverse a few hundred cycles after the last pivot.  The messages must be armed (armed = empty pattern)
// before the workgroup barrier that precedes the step.
__device__ __forceinline__ void arm_pivot_messages(double* smem, int tid) {
  for (int e = tid; e < 16 * kMsg; e += 256) smem[msg_off(e / kMsg) + e % kMsg] = __longlong_as_double(-1ll);
}
// one wave re-arms the messages of pivots [j0, j1) (between two diagonal blocks, while it has nothing else to do)
__device__ __forceinline__ void rearm_pivot_messages(double* smem, int lane, int j0, int j1) {
  for (int jj = j0; jj < j1; ++jj) smem[msg_off(jj) + lane] = __longlong_as_double(-1ll);
}
// first wave: d = symmetric positive definite 16 x 16 block, accumulator layout.  false on a non-positive / non-finite pivot.
__device__ __forceinline__ bool ldl16_eliminate(dbl4_t d, double* msg, int lane, long long* estamp = nullptr) {
  const int c = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) double* lds_ptr;
  lds_ptr lm = (lds_ptr)msg;
  bool ok = true;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int gg = jj & 3, vv = jj >> 2;
    const bool grp = g == gg;
    const double dv = d[vv];
    const double piv = readlane_f64(dv, 16 * gg + jj);
    ok = ok && (piv > 0.0) && isfinite(piv);
    // -dv / piv = -(dv y0) (1 + e + e^2), e = 1 - piv y0: v_rcp_f64 is good to ~2^-23, e^3 is far below rounding; dv y0 runs
    // beside e, so the chain behind v_rcp is three operations long
    const double y0 = __builtin_amdgcn_rcp(piv), e = fma(-piv, y0, 1.0), t = dv * y0;
    const double q = fma(t, fma(e, e, e), t);
    const double row = grp ? dv : 0.0;                  // B: row jj of D   (k = gg slice, zero elsewhere)
    const double mul = (grp && c > jj) ? -q : 0.0;      // A: -D[jj][i] / d for the rows i below the pivot
    lm[msg_off(jj) + lane] = (grp && c == jj) ? dv : mul;  // the message: the multipliers, with the pivot itself in the (otherwise zero) lane of the diagonal
    // The message must LEAVE here: left alone the compiler pairs the sixteen stores up (ds_write2st64_b64) and sinks them behind the
    // thirteenth MFMA of the unrolled loop — the following wave then sees its first message 2 800 cycles into a 3 400-cycle block and
    // ends 2 450 cycles after this one, three times per tile (round 6: read off the ISA; profiles/r06/tile_factor.txt).
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (estamp && lane == 0) estamp[jj] = clock64();       // (tools/tile_factor_bench.hip: when each message is posted)
    if (jj < 15) d = __builtin_amdgcn_mfma_f64_16x16x4f64(mul, row, d, 0, 0, 0);
  }
  return ok;
}
// second wave: W = chol(d)^-1 (lower, exact zeros above the diagonal), accumulator layout
__device__ __forceinline__ dbl4_t ldl16_follow(const double* msg, int lane, long long* fstamp = nullptr) {
  const int c = lane & 15, g = lane >> 4;
  typedef const volatile __attribute__((address_space(3))) double* lds_cvptr;   // volatile: re-read on every look; explicitly LDS (a volatile generic pointer would be read with flat loads)
  lds_cvptr vm = (lds_cvptr)msg;
  dbl4_t w;
#pragma unroll
  for (int v = 0; v < 4; ++v) w[v] = (g + 4 * v == c) ? 1.0 : 0.0;
  double pv[4] = {1.0, 1.0, 1.0, 1.0};   // the pivots of this lane's four rows
  double next = vm[msg_off(0) + lane];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int gg = jj & 3, vv = jj >> 2;
    const bool grp = g == gg;
    double m = next;
    while (__ballot(!filled(m)) != 0ull) m = vm[msg_off(jj) + lane];
    // the next message is requested BEFORE this pivot's MFMA and looked at after it (a speculative read: what is not there yet
    // is read again above) — the LDS round trip would otherwise sit between every two MFMAs of this wave
    if (jj < 15) next = vm[msg_off(jj < 15 ? jj + 1 : 15) + lane];
    __builtin_amdgcn_sched_barrier(0);
    const double piv = readlane_f64(m, 16 * gg + jj);
    const double mul = (grp && c == jj) ? 0.0 : m;
    const double wrow = grp ? w[vv] : 0.0;                     // B: row jj of W' = L'^-1 as it stands
    if (jj < 15) w = __builtin_amdgcn_mfma_f64_16x16x4f64(mul, wrow, w, 0, 0, 0);
    pv[vv] = grp ? piv : pv[vv];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) w[v] *= rsqrt_nr(pv[v]);
  return w;
}

// Second wave, round-3 form.  The MFMA form above applies a pivot's multipliers to the identity with one rank-1 MFMA whose B operand
// is a row of its own result: a chain of dependent MFMAs (81 cycles each) with an LDS message wait in every link — 285 cycles per
// pivot against the 206 of the eliminating wave, so W trailed the last pivot of every 16-pivot block by ~2 400 cycles: a third of the
// tile's time (tools/tile_factor_bench, round 2).  Nothing here needs the matrix pipe.  Lane c (< 16) keeps COLUMN c of
// E = L'^-1 in sixteen registers; pivot jj adds (-m_i) E[jj][c] to the rows i > jj — its own register jj times multipliers that are
// the same for every lane and come as broadcast reads of the message — 15 - jj independent FMAs, so the wave keeps pace with the
// eliminating one; row jj is final the moment pivot jj has been applied and goes to W (scaled by d_jj^-1/2, whose Newton steps hide
// the LDS latency of the next message) right away: what trails the last pivot is one reciprocal square root and one row.
// Writes the full 16 x 16 block W(o.., o..) (zeros above the diagonal) into Wl.
// 1/sqrt(x) from the hardware estimate y0 (good to ~2^-23) by one cubic step: e = 1 - x y0^2, y = y0 (1 + e/2 + 3 e^2/8); e^3 is far
// below rounding.  Four dependent operations behind v_rsq_f64 (the two Newton steps of rsqrt_nr are ten).
__device__ __forceinline__ double rsqrt_cubic(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y0, y0, 1.0);
  return fma(y0 * e, fma(0.375, e, 0.5), y0);
}
__device__ __forceinline__ void ldl16_follow_rows(const double* msg, double* Wl, int o, int lane, long long* fstamp = nullptr) {
  typedef const volatile __attribute__((address_space(3))) double* lds_cvptr;
  typedef double dbl2_t __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) dbl2_t* lds_c2ptr;
  typedef __attribute__((address_space(3))) double* lds_ptr;
  lds_cvptr vm = (lds_cvptr)msg;
  lds_ptr wl = (lds_ptr)Wl;
  if (lane >= 16) return;   // (the other lanes would only repeat the sixteen columns)
  const int c = lane;
  double e[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) e[i] = (i == c) ? 1.0 : 0.0;
  // Software pipeline, one pivot deep: when message jj is seen its multipliers are REQUESTED (same address in every lane: broadcast
  // reads, two cells each) and, while they travel, pivot jj - 1 — whose multipliers came in during the step before — is applied and
  // row jj - 2 (final since then) scaled and stored.  The eliminating wave writes a message with ONE LDS store and the sixteen cells of
  // lane group jj % 4 belong to one pass of it: once the pivot is there so are the multipliers; the compiler must not move their
  // loads above the polling loop, though.
  dbl2_t m2[2][8];
  double piv_prev = 0.0, piv_cur = 0.0;   // the pivots of messages jj - 1 and jj (each is used one step after it is seen)
  double next = vm[msg_off(0)];   // the pivot of message jj sits in the cell of lane 16 (jj % 4) + jj; requested one step ahead, looked at when needed
#pragma unroll
  for (int jj = 0; jj <= 16; ++jj) {
    if (jj < 16) {
      const int cell = msg_off(jj) + 16 * (jj & 3);
      // The look at message jj was requested one step ago, BEFORE that step's multipliers: this first test waits for that one read only
      // (s_waitcnt lgkmcnt(n) with the multipliers still in flight).  Written as a loop the compiler waits for everything at its head —
      // then a step costs a whole LDS round trip of its five reads, ~230 cycles against the 204 of the eliminating wave, and the wave
      // ends 1 650 cycles behind it.  The second look (message not there yet) may wait for all it likes.
      double piv = next;
      if (__builtin_expect(__ballot(!filled(piv)) != 0ull, 0)) { do { piv = vm[cell + jj]; } while (__ballot(!filled(piv)) != 0ull); }
      piv_prev = piv_cur; piv_cur = piv;
      if (fstamp && lane == 0) fstamp[4 * jj] = clock64();   // (tools/tile_factor_bench.hip: when each message is seen)
      asm volatile("" ::: "memory");
      if (jj < 15) next = vm[msg_off(jj + 1) + 16 * ((jj + 1) & 3) + jj + 1];
#pragma unroll
      for (int h = 0; h < 8; ++h) if (2 * h + 1 > jj) m2[jj & 1][h] = *(lds_c2ptr)(msg + cell + 2 * h);
    }
    if (jj >= 1) {   // pivot jj - 1
      const int k = jj - 1;
      // d_k^-1/2 (rsqrt_cubic, spelled out) is threaded through the row updates: its four dependent steps wait out their latencies
      // behind independent FMAs.  Everything is pinned where it stands: left alone, the compiler turns the fifteen independent
      // updates of a pivot into one dependent sum per row, formed right before the row is used — k FMA latencies on the chain.
      const double x = jj < 16 ? piv_prev : piv_cur;   // pivot k (at jj = 16 no new message has shifted the pair)
      double y0 = __builtin_amdgcn_rsq(x), t0 = 0.0, q0 = 0.0, p0 = 0.0;
      asm volatile("" : "+v"(y0));
#pragma unroll
      for (int i = k + 1; i < 16; ++i) {
        e[i] = fma(m2[k & 1][i >> 1][i & 1], e[k], e[i]); asm volatile("" : "+v"(e[i]));
        const int step = i - k;
        if (step == 2) { t0 = -x * y0; asm volatile("" : "+v"(t0)); }
        if (step == 4) { t0 = fma(t0, y0, 1.0); asm volatile("" : "+v"(t0)); }                       // e = 1 - x y0^2
        if (step == 6) { q0 = y0 * t0; p0 = fma(0.375, t0, 0.5); asm volatile("" : "+v"(q0), "+v"(p0)); }
      }
      const int nstep = 15 - k;   // (the pivots near the end of the block have too few updates to hide behind: the rest of the chain follows here)
      if (nstep < 2) t0 = -x * y0;
      if (nstep < 4) t0 = fma(t0, y0, 1.0);
      if (nstep < 6) { q0 = y0 * t0; p0 = fma(0.375, t0, 0.5); }
      const double rs = fma(q0, p0, y0);
      wl[(o + k) * TP + o + c] = rs * e[k];   // row k has been final since pivot k - 1 (exact zeros right of the diagonal: E is lower triangular)
      if (fstamp && lane == 0) fstamp[4 * k + 3] = clock64();   // (... and when each row has been stored)
    }
  }
}

// the 16 x 16 block at (o, o) of a tile whose lower triangle is valid (LDS, pitch TP), mirrored into the accumulator layout
__device__ __forceinline__ dbl4_t load_sym16(const double* D, int o, int lane) {
  const int c = lane & 15, g = lane >> 4;
  dbl4_t d;
#pragma unroll
  for (int v = 0; v < 4; ++v) { const int r = g + 4 * v; d[v] = D[(o + (r > c ? r : c)) * TP + o + (r > c ? c : r)]; }
  return d;
}
__device__ __forceinline__ dbl4_t load16(const double* M, int rb, int cb, int lane) {
  dbl4_t d;
#pragma unroll
  for (int v = 0; v < 4; ++v) d[v] = M[(rb + (lane >> 4) + 4 * v) * TP + cb + (lane & 15)];
  return d;
}
__device__ __forceinline__ void put16(double* M, int rb, int cb, dbl4_t x, int lane) {
#pragma unroll
  for (int v = 0; v < 4; ++v) M[(rb + (lane >> 4) + 4 * v) * TP + cb + (lane & 15)] = x[v];
}
// 16 x 16 x 16 block products out of LDS (pitch TP): acc + sign * X(xr.., xc..) Y(yr.., yc..)^T  resp.  acc + sign * X Y
__device__ __forceinline__ dbl4_t mm16_nt(const double* X, int xr, int xc, const double* Y, int yr, int yc, dbl4_t acc, double sign, int lane) {
  const int mi = lane & 15, mg = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * X[(xr + mi) * TP + xc + 4 * kk + mg], Y[(yr + mi) * TP + yc + 4 * kk + mg], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ dbl4_t mm16_nn(const double* X, int xr, int xc, const double* Y, int yr, int yc, dbl4_t acc, double sign, int lane) {
  const int mi = lane & 15, mg = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * X[(xr + mi) * TP + xc + 4 * kk + mg], Y[(yr + 4 * kk + mg) * TP + yc + mi], acc, 0, 0, 0);
  return acc;
}

// W = chol(D)^-1 for the T x T tile D in LDS (lower triangle valid, pitch TP; destroyed) into Wl (lower blocks; the blocks
// above the diagonal are left untouched), by 16 x 16 blocks: the first two waves walk the three diagonal blocks (the serial
// chain), the others keep the off-diagonal algebra out of their way:
//   L_I0 = D_I0 W_00^T ;  D_11 -= L_10 L_10^T, D_21 -= L_20 L_10^T, D_22 -= L_20 L_20^T ;  L_21 = D_21 W_11^T ;  D_22 -= L_21 L_21^T
//   W_10 = -W_11 (L_10 W_00),  W_21 = -W_22 (L_21 W_11),  W_20 = -W_22 (L_20 W_00 + L_21 W_10).
// D, Wl, Tm, Lp must be the task's LDS buffers 0 .. 3 (the pivot messages live in idle rows of Tm and Lp, see the LDS map) and the
// messages armed (arm_pivot_messages) before the barrier in front of this call.  All threads must call; ends with a barrier.  Returns (in the first wave)
// false on a non-positive pivot.
// wout (HBM, row-major T x T; null = keep W in LDS only): the inverse leaves by ROW BLOCKS as they become final — rows 0-15 after the
// first diagonal block (a third into the factorisation), rows 16-31 after the second, rows 32-47 at the end, stored by waves that
// have nothing else to do — so that the DIAG task of the next column on the critical chain (task_diag) forms L = X W^T and D -= L L^T
// column block by column block under the rest of this factorisation; what is left for it when the last rows land is a third of both.
template <bool TRACE = false, bool MFMA_FOLLOWER = false, bool DAG = false>
__device__ __forceinline__ bool factor_invert_tile(double* D, double* Wl, double* Tm, double* Lp, int tid, long long* stamps = nullptr, double* wout = nullptr) {
  const int wave = tid >> 6, lane = tid & 63;
  const dbl4_t zero = {0.0, 0.0, 0.0, 0.0};
  bool ok = true;
  int nstamp = 0;
  auto stamp = [&]() { if (TRACE && tid == 0) stamps[nstamp++] = clock64(); };   // tools/tile_factor_bench.hip
  double* msg = D;   // (D is the first LDS buffer of the task; the messages sit at msg_off() behind it)
  auto publish_rows = [&](int rb, int nthreads, int t) {   // rows 16 rb .. 16 rb + 15 of W: lower blocks from Wl, exact zeros right of the diagonal block
    for (int e = t; e < 16 * T; e += nthreads) { const int r = 16 * rb + e / T, c = e % T; st<DAG>(wout + r * T + c, c < 16 * (rb + 1) ? Wl[r * TP + c] : 0.0); }
  };
  stamp();
  // ONE copy of the two sixteen-pivot loops (the bulk of this function's code), gone through three times: inside the solver a task runs
  // this function once, and its instructions come from memory — with the three diagonal blocks spelled out one after the other the
  // factorisation took 8.3 us there against 7.1 us in tools/tile_factor_bench.hip's warm loop, and nothing that made the warm loop
  // faster showed up in the solver.  Blocks 1 and 2 now run from the instruction cache.
#pragma clang loop unroll(disable)
  for (int b = 0; b < 3; ++b) {
    const int o = 16 * b;
    if (wave == 0) {
      dbl4_t d = load_sym16(D, o, lane);
      if (b > 0) d = mm16_nt(Lp, o, o - 16, Lp, o, o - 16, d, -1.0, lane);   // D_bb - L_b,b-1 L_b,b-1^T (block 2 has lost L_20 L_20^T already)
      ok = ldl16_eliminate(d, msg, lane) && ok;
    } else if (wave == 1) {
      if (MFMA_FOLLOWER) put16(Wl, o, o, ldl16_follow(msg, lane, (TRACE && b == 0) ? stamps + 16 : nullptr), lane); else ldl16_follow_rows(msg, Wl, o, lane);
    } else if (b == 1) {
      if (wave == 2) {
        put16(D, 32, 16, mm16_nt(Lp, 32, 0, Lp, 16, 0, load16(D, 32, 16, lane), -1.0, lane), lane);
        put16(Tm, 16, 0, mm16_nn(Lp, 16, 0, Wl, 0, 0, zero, 1.0, lane), lane);                                         // T_10 = L_10 W_00
      } else {
        put16(D, 32, 32, mm16_nt(Lp, 32, 0, Lp, 32, 0, load16(D, 32, 32, lane), -1.0, lane), lane);                     // (its upper half is never read)
        if (wout) publish_rows(0, 64, lane);   // W_00 has been final since the first barrier; this step is long (sixteen pivots), the one before is not
      }
    } else if (b == 2) {
      if (wave == 2) put16(Tm, 32, 16, mm16_nn(Lp, 32, 16, Wl, 16, 16, zero, 1.0, lane), lane);                        // T_21 = L_21 W_11
      else {
        put16(Tm, 32, 0, mm16_nn(Lp, 32, 16, Wl, 16, 0, load16(Tm, 32, 0, lane), 1.0, lane), lane);                     // T_20 += L_21 W_10
        if (wout) publish_rows(1, 64, lane);   // W_10, W_11 have been final since the barrier above
      }
    }
    stamp(); lds_barrier(); stamp();
    if (b == 0) {
      if (wave < 2) put16(Lp, 16 + 16 * wave, 0, mm16_nt(D, 16 + 16 * wave, 0, Wl, 0, 0, zero, 1.0, lane), lane);   // L_10, L_20
      else rearm_pivot_messages(D, lane, 8 * (wave - 2), 8 * (wave - 1));   // (both waves of block 0 are done with them; block 1 starts behind the next barrier)
    } else if (b == 1) {
      if (wave == 0) put16(Lp, 32, 16, mm16_nt(D, 32, 16, Wl, 16, 16, zero, 1.0, lane), lane);       // L_21
      else if (wave == 1) put16(Wl, 16, 0, mm16_nn(Wl, 16, 16, Tm, 16, 0, zero, -1.0, lane), lane);  // W_10
      else if (wave == 2) put16(Tm, 32, 0, mm16_nn(Lp, 32, 0, Wl, 0, 0, zero, 1.0, lane), lane);     // T_20 = L_20 W_00
      else rearm_pivot_messages(D, lane, 0, 16);
    } else {
      if (wave == 0) put16(Wl, 32, 16, mm16_nn(Wl, 32, 32, Tm, 32, 16, zero, -1.0, lane), lane);      // W_21
      else if (wave == 1) put16(Wl, 32, 0, mm16_nn(Wl, 32, 32, Tm, 32, 0, zero, -1.0, lane), lane);   // W_20
    }
    stamp(); lds_barrier(); stamp();
  }
  if (wout) publish_rows(2, 256, tid);
  return ok;
}

// ---- MFMA tile products ------------------------------------------------------------------------------------
// C(48x48) += A B^T on v_mfma_f64_16x16x4_f64, operands straight from HBM/L2 into registers — no LDS, no
// barrier in the accumulation loop.  The four waves of the workgroup split K: wave w owns columns
// [12w, 12w+12) of both operand tiles, lane (r = lane & 15, g = lane >> 4) holds columns 12w + 3g + t (t < 3)
// of rows 16I + r — for MFMA step t that is A[i = r][k = g] and B[k = g][j = r] of the 16x16x4 product over
// the k values {12w + 3g' + t}.  Every wave therefore accumulates all nine 16x16 blocks over its quarter of
// K (27 MFMAs, 64 cycles each: the fp64 matrix rate of gfx950 equals its vector rate, but the VALU form of
// this product is bound by LDS operand reads at a third of it) and the four partial tiles meet once, in LDS,
// when the task has gone through all its contributors.
// Result layout of the instruction (measured, tools/mfma_f64_check.hip): lane holds rows g + 4v (v < 4), column r.
typedef double dbl4 __attribute__((ext_vector_type(4)));

constexpr bool kCachedOperands = true;   // (false: every operand load coherent, as in rounds 1 - 4; A/B by rebuilding)
struct Frag {
  double v[3][3];   // [row block I][step t]
  template <bool DAG, bool COHERENT>
  __device__ __forceinline__ void load(const double* tile, int wave, int lane);
};

struct Acc {
  dbl4 c[3][3];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
      for (int J = 0; J < 3; ++J) c[I][J] = dbl4{0.0, 0.0, 0.0, 0.0};
  }
  template <bool LOWER>
  __device__ __forceinline__ void mac(const Frag& a, const Frag& b) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J)
          if (!LOWER || J <= I) c[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[I][t], b.v[J][t], c[I][J], 0, 0, 0);
  }
  // this wave's partial tile -> its LDS buffer (row-major, pitch TP)
  __device__ __forceinline__ void spill(double* buf, int lane) const {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
      for (int J = 0; J < 3; ++J)
#pragma unroll
        for (int v = 0; v < 4; ++v) buf[(16 * I + g + 4 * v) * TP + 16 * J + r] = c[I][J][v];
  }
};

// ---- left-looking tile tasks ----------------------------------------------------------------------------
// The host orders the tile columns by nested dissection and lists, per tile of the factor, the finished tiles it
// pulls its updates from (left-looking: no two workgroups ever write the same tile, so no fp atomics and a fixed
// summation order).  Four task kinds, each run by one 256-thread workgroup:
//   UPDATE  partial = sum_{k in chunk} L_ik L_jk^T (+ sum L_jk z_k for a diagonal tile) -> scratch
//   DIAG    S_jj -= updates ; S_jj = L_jj L_jj^T ; W_j = L_jj^-1 ; z_j = W_j (b_j - updates)
//   SUB     S_ij -= updates ; L_ij = S_ij W_j^T
//   BACK    y_j = W_j^T ( z_j - sum_i L_ij^T y_i )
// The explicit inverse of the 48x48 diagonal factor turns every triangular solve behind it (dozens of SUB tiles
// per column, the forward and the backward substitution) into a product without a serial chain.
// Two drivers run the same task bodies:
//   * chol_dag_kernel — ONE persistent launch per solve.  Workgroups draw tickets from a global counter over the
//     topologically sorted task list.  A claimed task only waits for tasks with smaller tickets, which are claimed by
//     running workgroups, so progress never depends on how many workgroups are resident.  There are no flags:
//     everything a task hands to another one (factor tiles, partial tiles, W, z, y) is a WRITE-ONCE CELL — its
//     array is filled with an "empty" bit pattern before the launch, a producer stores each double exactly once,
//     and a consumer that reads "empty" simply reads again.  Waiting for a tile therefore costs one memory
//     round trip after it lands (the data is its own flag) instead of a flag round trip plus a data round trip,
//     and a producer has nothing to wait for before it moves on: measured 9 us less per level of the dependency
//     chain than release / flag / acquire hand-offs.
//   * chol_level_kernel — one launch per (level, kind), no waiting at all: the reference schedule the DAG driver is
//     tested against bit for bit (RSBA_CHOL_LEVELS=1).

__shared__ long long* s_trace_slot;   // RSBA_CHOL_TRACE: where the running task logs its time stamps (null = off)
#define CHOL_STAMP(k) do { if (DAG && tid == 0 && s_trace_slot) s_trace_slot[k] = wall_clock64(); } while (0)

// Write-once cells.  "Empty" is all ones — what hipMemset(0xFF) leaves; as a double it is a NaN with a payload that
// no arithmetic produces (a failed factorisation yields the canonical NaN, which counts as data: no task can hang).
// Cells move through agent-coherent accesses: relaxed agent-scope atomic loads / stores of the individual doubles,
// which gfx950 issues with sc1 (the L2 of the other XCDs is not coherent with ours; sc1 accesses go to the memory
// side), so neither side needs cache maintenance.  The level driver (DAG = false) uses plain loads and stores.
// Every access to HBM below says so explicitly (address space 1).  The persistent kernel reads its pointers out of a device copy
// of the plan, so the compiler cannot tell where they point: left alone it issues FLAT loads, which count on the LDS counter
// (lgkmcnt) as well — and every LDS wait and LDS-only barrier of a task then also waits for the prefetches in flight.
// Waiting costs memory traffic: a task that found a group of cells incomplete does not keep re-reading the whole group
// (hundreds of claimed-but-waiting tasks doing that saturate the memory system) — it watches ONE cell of the
// missing input, with a pause between looks, and reads the group again once that cell has landed.
// Instrumented build only (-DRSBA_TEST_HOOKS, librsba_amd_hooks.so): what the persistent kernel reads COHERENTLY (from the memory side), by
// kind — tools/chol_poll_split.py sets it beside the fabric-read counter of a rocprofv3 pass.  Wave-level events, one atomic per event by lane 0.
//   [0] looks at a watched cell (one 64-byte line each)      [1] of them: looks that found the cell still empty
//   [2] operand fragments read again coherently (9 x 512 B)   [3] W row blocks / tiles read (12 or 24 x 512 B)   [4] of them: reads that came back incomplete
#ifdef RSBA_TEST_HOOKS
__device__ unsigned long long g_chol_coherent[8];
#define CHOL_COUNT(k, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_chol_coherent[k], (unsigned long long)(n)); } while (0)
#else
#define CHOL_COUNT(k, n) do { } while (0)
#endif
template <bool DAG>
__device__ __forceinline__ void watch_cell(const double* p) {
  if (!DAG) return;
  for (;;) {
    const bool there = filled(ld<true>(p));
    CHOL_COUNT(0, 1);
    if (there) break;
    CHOL_COUNT(1, 1);
    __builtin_amdgcn_s_sleep(8);
  }
}
// a waiting task marks when its last late input arrived (RSBA_CHOL_TRACE)
__device__ __forceinline__ void note_late_input() { if (threadIdx.x == 0 && s_trace_slot) s_trace_slot[2] = wall_clock64(); }

__device__ __forceinline__ double* factor_ptr(const SolverDev& sv, int slot) { return sv.Lf + (size_t)slot * (T * T); }



// COHERENT = false: ordinary loads, served by this XCD's L2 (and the CU's L1) when they hold the line.  A cell is written once, so what
// such a load returns is either the cell's final value or "empty" — possibly a stale empty (a copy cached before the producer's store, which
// went to the memory side); the caller checks every double it consumes and reads an incomplete fragment again COHERENTLY (sc1: from the
// memory side).  A factor tile is an operand of ~12 products in as many tasks (chol_dag_kernel pulled 12 x the factor's size from the
// fabric when every operand load was coherent): the tasks that come after the first find it in L2.  Lines of an EARLIER solve cannot
// be met: the cells are re-armed by a launch of their own, and a launch starts with its XCD's L2 invalidated.
// What that last sentence rests on: the two sets of cells alternate from solve to solve; a set is re-armed (hipMemsetAsync on
// the arming stream) while the OTHER set's kernel runs, and this kernel waits for its own set's re-arming through an event.  A kernel
// dispatch packet carries an acquire fence of agent scope or wider (HSA AQL header, set by the HIP runtime for every launch): the L2 of
// every XCD the launch lands on is invalidated before the first wave runs, so a FILLED line of this set from two solves ago cannot be in
// any L2 when the kernel starts, and inside the kernel a line of this set can only have been cached empty or final.  Should a runtime
// ever launch without that fence, the result is a wrong factor — which the residual check that follows every DAG solve (verify_dag, on
// by default: solver.hip) catches and repairs on the level schedule; tools/dag_stress.py alternates both sets for thousands of solves.
template <bool DAG, bool COHERENT>
__device__ __forceinline__ void Frag::load(const double* tile, int wave, int lane) {
  const double* p = tile + (lane & 15) * T + 12 * wave + 3 * (lane >> 4);
#pragma unroll
  for (int I = 0; I < 3; ++I)
#pragma unroll
    for (int t = 0; t < 3; ++t) v[I][t] = (DAG && COHERENT) ? ld<true>(p + 16 * I * T + t) : gl(p + 16 * I * T + t);
  if (DAG && COHERENT) CHOL_COUNT(2, 1);
}

// sum_{p in [p0,p1)} L_a(p) L_b(p)^T into acc (K-split over the waves), and for DIAG lists sum L_jk z_k into
// bz[I] (this lane's share of rows 16I + r).  DIAG list entries are {slot_jk, tile of z_k}: A = B = L_jk and only
// the lower blocks are formed; SUB entries are {slot_ik, slot_jk}.
template <bool DAG, bool DIAG>
__device__ __forceinline__ void accumulate(const SolverDev& sv, const CholPlan& pl, const int32_t* list, int p0, int p1, Acc& acc, double bz[3],
                                           int wave, int lane) {
  if (p0 >= p1) return;
  // Operands travel in groups of kGroup contributors, one group ahead of the MFMAs: the HBM round trip of a tile
  // (~2 us) is several times the 0.7 us its product takes.  Loads are issued in straight lines (the tail of the list
  // re-reads its last contributor rather than branch).  The prefetch is speculative — a tile that has not been
  // produced yet reads as empty cells — and a group is checked when it is about to be multiplied: a wave whose
  // share of it is incomplete reads it again until it is.
  constexpr int kGroup = 1;   // (one contributor ahead: with two, the operand registers push the persistent kernel past 256 per lane and a CU holds one workgroup instead of two)
  struct Group { Frag a[kGroup], b[kGroup]; double z[kGroup][3]; };
  Group cur, nxt;
  auto fetch_group = [&](Group& g, int p, auto coherent) {   // (coherent: std::true_type — the second look at a group that came in incomplete; the look ahead goes through the caches)
    constexpr bool C = decltype(coherent)::value || !kCachedOperands;
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
      const int q = min(p + u, p1 - 1);
      g.a[u].template load<DAG, C>(factor_ptr(sv, gl(list + 2 * q)), wave, lane);
      if (!DIAG) g.b[u].template load<DAG, C>(factor_ptr(sv, gl(list + 2 * q + 1)), wave, lane);
      else {
        const double* zk = sv.zv + (size_t)gl(list + 2 * q + 1) * T + 12 * wave + 3 * (lane >> 4);
        g.z[u][0] = ld<DAG>(zk); g.z[u][1] = ld<DAG>(zk + 1); g.z[u][2] = ld<DAG>(zk + 2);
      }
    }
  };
  auto complete = [&](const Group& g) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int t = 0; t < 3; ++t) { ok = ok && filled(g.a[u].v[I][t]); if (!DIAG) ok = ok && filled(g.b[u].v[I][t]); }
      if (DIAG) ok = ok && filled(g.z[u][0]) && filled(g.z[u][1]) && filled(g.z[u][2]);
    }
    return __ballot(!ok) == 0ull;
  };
  auto mac_group = [&](const Group& g, int p) {
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
      if (p + u < p1) {
        if (DIAG) {
          acc.mac<true>(g.a[u], g.a[u]);
#pragma unroll
          for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int t = 0; t < 3; ++t) bz[I] += g.a[u].v[I][t] * g.z[u][t];
        } else {
          acc.mac<false>(g.a[u], g.b[u]);
        }
      }
    }
  };
  fetch_group(cur, p0, std::false_type{});
  for (int p = p0; p < p1; p += kGroup) {
    const int pn = p + kGroup;
    if (pn < p1) fetch_group(nxt, pn, std::false_type{});
    if (DAG) {
      bool late = false;
      while (!complete(cur)) {
        // the list is in the order the contributors finish: watch the last one of the group
        const int q = min(p + kGroup, p1) - 1;
        watch_cell<DAG>(factor_ptr(sv, gl(list + 2 * q)) + 12 * wave);
        if (!DIAG) watch_cell<DAG>(factor_ptr(sv, gl(list + 2 * q + 1)) + 12 * wave);
        fetch_group(cur, p, std::true_type{});
        late = true;
      }
      if (late) note_late_input();
    }
    mac_group(cur, p);
    if (pn < p1) cur = nxt;
  }
}

// this wave's rhs partials: fold the four lane groups, lanes g == 0 write rows 16I + r of their wave's LDS vector
__device__ __forceinline__ void spill_bz(double bz[3], double* vec, int lane) {
#pragma unroll
  for (int I = 0; I < 3; ++I) {
    double x = bz[I];
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    if (lane < 16) vec[16 * I + lane] = x;
  }
}

template <bool DAG>
__device__ __forceinline__ void task_update(const SolverDev& sv, const CholPlan& pl, int item, double* smem, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  const int32_t* u = pl.upd + 4 * item;
  const int u0 = gl(u), u1 = gl(u + 1), u2 = gl(u + 2), u3 = gl(u + 3);
  const bool diag = u0 == 0;
  Acc acc; acc.clear();
  double bz[3] = {0, 0, 0};
  if (diag) accumulate<DAG, true>(sv, pl, pl.diag_list, u1, u2, acc, bz, wave, lane);
  else accumulate<DAG, false>(sv, pl, pl.sub_list, u1, u2, acc, bz, wave, lane);
  CHOL_STAMP(3);
  acc.spill(smem + wave * kBuf, lane);
  if (diag) spill_bz(bz, smem + kVecOff + wave * T, lane);
  lds_barrier();
  double* out = sv.chol_part + (size_t)u3 * (T * T + T);
  for (int e = tid; e < T * T; e += 256) {
    const int o = (e / T) * TP + e % T;
    st<DAG>(out + e, (smem[o] + smem[kBuf + o]) + (smem[2 * kBuf + o] + smem[3 * kBuf + o]));
  }
  if (diag && tid < T) { const double* v = smem + kVecOff; st<DAG>(out + T * T + tid, (v[tid] + v[T + tid]) + (v[2 * T + tid] + v[3 * T + tid])); }
  CHOL_STAMP(4);
}

// the partial tiles (and, WITH_B, rhs partials) of UPDATE tasks, four at a time: each thread re-reads its own nine
// cells of a partial until they are all there
template <bool DAG, bool WITH_B>
__device__ __forceinline__ void subtract_partials(const SolverDev& sv, int part0, int nparts, double sreg[9], double& breg, int tid) {
  for (int c0 = 0; c0 < nparts; c0 += 4) {
    double pv[4][9], pb[4] = {0.0, 0.0, 0.0, 0.0};
    bool late = false;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* part = sv.chol_part + (size_t)(part0 + min(c0 + u, nparts - 1)) * (T * T + T);
#pragma unroll
        for (int q = 0; q < 9; ++q) pv[u][q] = ld<DAG>(part + tid + 256 * q);
        if (WITH_B && tid < T) pb[u] = ld<DAG>(part + T * T + tid);
      }
      if (!DAG) break;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int q = 0; q < 9; ++q) ok = ok && filled(pv[u][q]);
        ok = ok && filled(pb[u]);
      }
      if (ok) break;
      late = true;
      watch_cell<DAG>(sv.chol_part + (size_t)(part0 + min(c0 + 3, nparts - 1)) * (T * T + T) + tid);
    }
    if (late) note_late_input();
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c0 + u < nparts) {
#pragma unroll
        for (int q = 0; q < 9; ++q) sreg[q] -= pv[u][q];
        breg -= pb[u];
      }
  }
}

// (Round-1 form, see potrf_blocked: only tools/tile_factor_bench.hip calls it.)
// W = L^-1 for the factored lower-triangular tile L in LDS (pitch TP), into Wl (pitch TP; blocks above the
// diagonal left untouched), by 16 x 16 blocks over all 256 threads — thread (r, c) = (tid >> 4, tid & 15) owns
// element (r, c) of every block:
//   W_bb = L_bb^-1 (forward substitution, one wave per block, a lane per column), then, as 16 x 16 MFMA products,
//   W_10 = -W_11 (L_10 W_00),  W_21 = -W_22 (L_21 W_11),  W_20 = -W_22 (L_20 W_00 + L_21 W_10).
// Tm (pitch TP) is scratch for the inner products.  All threads must call; ends with a barrier.
__device__ __forceinline__ void invert_lower_blocked(const double* L, const double* dinv, double* Wl, double* Tm, int tid) {
  constexpr int B = 16;
  const int wave = tid >> 6, lane = tid & 63, r = tid >> 4, c = tid & 15;
  if (wave < 3) {
    const int o = B * wave, cc = lane < B ? lane : B - 1;
    double w[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      double s = (i == cc) ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < i; ++m) s -= L[(o + i) * TP + o + m] * w[m];
      w[i] = s * dinv[o + i];
    }
    if (lane < B) {
#pragma unroll
      for (int i = 0; i < B; ++i) Wl[(o + i) * TP + o + cc] = w[i];
    }
  }
  lds_barrier();
  // 16 x 16 block products on the matrix pipe, one wave each: (rows xr.., columns xo.. of X) times (rows yo.., columns
  // yc.. of Y), result in the MFMA layout (this lane: column lane & 15, rows (lane >> 4) + 4v)
  const int mi = lane & 15, mg = lane >> 4;
  auto mm16 = [&](const double* X, int xr, int xo, const double* Y, int yo, int yc, dbl4 acc) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(xr + mi) * TP + xo + 4 * kk + mg], Y[(yo + 4 * kk + mg) * TP + yc + mi], acc, 0, 0, 0);
    return acc;
  };
  auto put = [&](double* M, int rb, int cb, dbl4 v, double sign) {
#pragma unroll
    for (int q = 0; q < 4; ++q) M[(rb + mg + 4 * q) * TP + cb + mi] = sign * v[q];
  };
  const dbl4 zero = {0.0, 0.0, 0.0, 0.0};
  (void)r; (void)c;
  if (wave == 0) put(Tm, 16, 0, mm16(L, 16, 0, Wl, 0, 0, zero), 1.0);          // T10 = L10 W00
  else if (wave == 1) put(Tm, 32, 16, mm16(L, 32, 16, Wl, 16, 16, zero), 1.0);  // T21 = L21 W11
  else if (wave == 2) put(Tm, 32, 0, mm16(L, 32, 0, Wl, 0, 0, zero), 1.0);      // T20 = L20 W00
  lds_barrier();
  if (wave == 0) put(Wl, 16, 0, mm16(Wl, 16, 16, Tm, 16, 0, zero), -1.0);       // W10 = -W11 T10
  else if (wave == 1) put(Wl, 32, 16, mm16(Wl, 32, 32, Tm, 32, 16, zero), -1.0);   // W21 = -W22 T21
  lds_barrier();
  if (wave == 0) {                                                              // T20 += L21 W10 ; W20 = -W22 T20
    dbl4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = Tm[(32 + mg + 4 * q) * TP + mi];
    put(Tm, 32, 0, mm16(L, 32, 16, Wl, 16, 0, t), 1.0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    put(Wl, 32, 0, mm16(Wl, 32, 32, Tm, 32, 0, zero), -1.0);
  }
  lds_barrier();
}

// L = X W_j^T for the tile X held in LDS (pitch TP): wave I < 3 forms row block I into c[J] (MFMA result layout: rows
// 16I + (lane >> 4) + 4v, column 16J + (lane & 15)); W is lower triangular, so column block J only needs k < 16 (J + 1).
// W_j comes from the DIAG task of column j: its cells are read until they are all there.
template <bool DAG, bool EAGER = false>
__device__ __forceinline__ void times_inverse_transposed(const SolverDev& sv, int tile_j, const double* X, dbl4 c[3], int wave, int lane) {
  const int I = wave, r = lane & 15, g = lane >> 4;
  const double* Wg = sv.Winv + (size_t)tile_j * (T * T);
  double wv[3][12];   // B operand: W[16J + r][4kk + g]
  {
    bool late = false;
    // A SUB task's FIRST look at W_j goes through the caches, like the operand tiles of accumulate() (round 6): a dozen SUB tasks of a column
    // read the same 18 KB, and every one of them used to fetch it from the memory side (180 MB per C4 launch for a 4.6 MB array of inverses,
    // tools/chol_poll_split.py).  W_j is written once: what an ordinary load returns is final or empty — possibly a stale empty, which
    // sends the task down the coherent path below, as before.  The DIAG task of the next column on the chain (EAGER) polls for rows that
    // are being written: coherent from the start.
    bool cached_look = DAG && !EAGER && kCachedOperands;
    for (;;) {
      bool ok = true;
      if (cached_look) {
#pragma unroll
        for (int J = 0; J < 3; ++J)
#pragma unroll
          for (int kk = 0; kk < 4 * (J + 1); ++kk) { wv[J][kk] = gl(Wg + (16 * J + r) * T + 4 * kk + g); ok = ok && filled(wv[J][kk]); }
      } else {
#pragma unroll
        for (int J = 0; J < 3; ++J)
#pragma unroll
          for (int kk = 0; kk < 4 * (J + 1); ++kk) { wv[J][kk] = ld<DAG>(Wg + (16 * J + r) * T + 4 * kk + g); ok = ok && filled(wv[J][kk]); }
        if (DAG) CHOL_COUNT(3, 24);
      }
      if (!DAG || __ballot(!ok) == 0ull) break;
      if (cached_look) { cached_look = false; watch_cell<DAG>(Wg + (T * T - 1) - T * (blockIdx.x & 15)); continue; }   // (not there, or a stale empty: wait for it on the memory side, then read it from there)
      CHOL_COUNT(4, 24);
      late = true;
      // A SUB task watches one cell and reads again when it has landed (hundreds of them wait at a time).  A DIAG task is the
      // next link of the critical chain and only a handful wait for their W at any moment: it reads everything again straight
      // away — one memory round trip less per level of the elimination tree.
      if (EAGER) __builtin_amdgcn_s_sleep(2);
      else watch_cell<DAG>(Wg + (T * T - 1) - T * (blockIdx.x & 15));   // the last cell of one of the last sixteen rows (all of them land with the last stores): the SUB tasks of a column spread over sixteen cache lines instead of all polling the one the producer is about to write
    }
    if (late) note_late_input();
  }
  if (DAG && EAGER && threadIdx.x == 0 && s_trace_slot) s_trace_slot[5] = wall_clock64();   // trace: W_k* complete in registers
  c[0] = c[1] = c[2] = dbl4{0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < 12; ++kk) {
    const double a = X[(16 * I + r) * TP + 4 * kk + g];
#pragma unroll
    for (int J = 0; J < 3; ++J)
      if (kk < 4 * (J + 1)) c[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wv[J][kk], c[J], 0, 0, 0);
  }
}

// One column block of the same product for the DIAG task that follows column j on the critical chain: rows 16 I .., columns 16 Jc .. of
// L = X W_j^T (the instruction sequence of times_inverse_transposed for that block: same bits), polling only the rows 16 Jc .. of W_j it
// needs — the whole row block is read again straight away until it is complete (only a handful of DIAG tasks wait at any moment).
template <bool DAG>
__device__ __forceinline__ dbl4 times_inverse_block(const SolverDev& sv, int tile_j, const double* X, int Jc, int wave, int lane) {
  const int I = wave, r = lane & 15, g = lane >> 4;
  const double* Wg = sv.Winv + (size_t)tile_j * (T * T) + (16 * Jc + r) * T + g;
  double wv[12];
  bool late = false;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int kk = 0; kk < 12; ++kk) if (kk < 4 * (Jc + 1)) { wv[kk] = ld<DAG>(Wg + 4 * kk); ok = ok && filled(wv[kk]); }
    if (DAG) CHOL_COUNT(3, 4 * (Jc + 1));
    if (!DAG || __ballot(!ok) == 0ull) break;
    CHOL_COUNT(4, 4 * (Jc + 1));
    late = true;
    __builtin_amdgcn_s_sleep(2);
  }
  if (late) note_late_input();
  dbl4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 12; ++kk)
    if (kk < 4 * (Jc + 1)) c = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(16 * I + r) * TP + 4 * kk + g], wv[kk], c, 0, 0, 0);   // (one accumulator, k ascending: the bits of times_inverse_transposed)
  return c;
}

// DIAG task, early half: the updates formed so far are in acc / bz (K split over the waves), the tile and rhs (less the partial
// tiles) in sreg / breg — D = S_jj - updates (lower triangle) and b = rhs - updates go to LDS.  No barrier at the end.
__device__ __forceinline__ void diag_assemble(double sreg[9], double breg, Acc& acc, double bz[3], double* smem, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  double* D = smem;
  double* vec = smem + kVecOff; double* bvec = vec + 4 * T;
  acc.spill(smem + wave * kBuf, lane);
  spill_bz(bz, vec + wave * T, lane);
  lds_barrier();
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int e = tid + 256 * q, r = e / T, c = e % T, o = r * TP + c;
    const double upd = (smem[o] + smem[kBuf + o]) + (smem[2 * kBuf + o] + smem[3 * kBuf + o]);
    sreg[q] = (c <= r) ? sreg[q] - upd : 0.0;
  }
  if (tid < T) bvec[tid] = breg - ((vec[tid] + vec[T + tid]) + (vec[2 * T + tid] + vec[3 * T + tid]));
  lds_barrier();
#pragma unroll
  for (int q = 0; q < 9; ++q) { const int e = tid + 256 * q; D[(e / T) * TP + e % T] = sreg[q]; }
}

// DIAG task, the serial half: D (LDS, complete) = L_jj L_jj^T, W_j = L_jj^-1, z_j = W_j b
template <bool DAG>
__device__ __forceinline__ void diag_factor(const SolverDev& sv, int tile_j, double* smem, int tid) {
  double* D = smem; double* Wl = smem + kBuf;
  double* vec = smem + kVecOff; double* bvec = vec + 4 * T;
  arm_pivot_messages(smem, tid);   // (rows of buffers 2 and 3 that held X / L of the last contributor until the barrier in front of this call)
  lds_barrier();
  CHOL_STAMP(4);
  // L_jj itself is never formed: everything downstream uses W_j
  const bool ok = factor_invert_tile<false, false, DAG>(D, Wl, smem + 2 * kBuf, smem + 3 * kBuf, tid, nullptr, sv.Winv + (size_t)tile_j * (T * T));   // (W leaves by row blocks from inside)
  if (tid == 0 && !ok) atomicExch(sv.chol_fail, 1);
  CHOL_STAMP(6);
  if (tid < 4 * T) {   // z_j = W b: four lanes per row, twelve columns each (the next DIAG task of the chain waits for it right behind the last rows of W)
    const int row = tid >> 2, q = tid & 3;
    double s0 = 0.0;
#pragma unroll
    for (int m = 0; m < T / 4; ++m) { const int c = (T / 4) * q + m; if (c <= row) s0 += Wl[row * TP + c] * bvec[c]; }
    s0 += __shfl_xor(s0, 1, 64);
    s0 += __shfl_xor(s0, 2, 64);
    if (q == 0) st<DAG>(sv.zv + (size_t)tile_j * T + row, s0);
  }
}

template <bool DAG>
__device__ __forceinline__ void task_diag(const SolverDev& sv, const CholPlan& pl, int b, double* smem, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  const int32_t* info = pl.diag_info + 4 * b;
  const int slot_jj = gl(info), tile_j = gl(info + 1), part0 = gl(info + 2), nparts = gl(info + 3);
  // The last contributor k* (the column that finishes one level before this one) is formed HERE from W_k* instead of being
  // read back from the SUB task of tile (j, k*) (item fs).
  const int fs = gl(pl.diag_fuse + b);
  const int p0 = gl(pl.diag_own + b), p1 = gl(pl.diag_ptr + b + 1) - (fs >= 0 ? 1 : 0);   // the owner's share of the contributor list
  double* XB = smem + kXBuf; double* LB = smem + kLBuf;
  // X = S_jk* - (older updates) comes from the SUB task of that tile, which publishes it before it starts waiting for W_k*:
  // requested here, looked at when it is needed
  double xreg[9];
  const double* xs = sv.Xpub + (size_t)b * (T * T);
  if (fs >= 0) {
#pragma unroll
    for (int q = 0; q < 9; ++q) xreg[q] = ld<DAG>(xs + tid + 256 * q);
  }
  // the tile itself (left by the Schur kernels before this launch) travels while the updates are formed
  double sreg[9];
  const double* src = tile_ptr(sv, slot_jj);
#pragma unroll
  for (int q = 0; q < 9; ++q) sreg[q] = gl(src + tid + 256 * q);
  double breg = tid < T ? gl(sv.rhs + (size_t)tile_j * T + tid) : 0.0;
  // the early part of a long contributor list arrives pre-reduced (UPDATE tasks), well before the owner's own share
  subtract_partials<DAG, true>(sv, part0, nparts, sreg, breg, tid);
  Acc acc; acc.clear();
  double bz[3] = {0, 0, 0};
  accumulate<DAG, true>(sv, pl, pl.diag_list, p0, p1, acc, bz, wave, lane);
  CHOL_STAMP(3);
  diag_assemble(sreg, breg, acc, bz, smem, tid);   // everything that does not need column k*: off the critical path
  if (fs >= 0) {
    const int tile_k = gl(pl.sub_col + fs);
    double* D = smem;
    double* vec = smem + kVecOff; double* bvec = vec + 4 * T; double* zs = vec + 7 * T;
    {
      bool late = false;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 9; ++q) ok = ok && filled(xreg[q]);
        if (!DAG || ok) break;
        late = true;
        watch_cell<DAG>(xs + tid + 256 * 8);
#pragma unroll
        for (int q = 0; q < 9; ++q) xreg[q] = ld<DAG>(xs + tid + 256 * q);
      }
      if (late) note_late_input();
#pragma unroll
      for (int q = 0; q < 9; ++q) { const int e = tid + 256 * q; XB[(e / T) * TP + e % T] = xreg[q]; }
    }
    lds_barrier();   // X and D are in LDS
    // W_k* arrives by row blocks (factor_invert_tile): column block Jc of L = X W^T needs rows 16 Jc .. of W only, and D -= L L^T is a sum
    // over the column blocks of L — so each stage runs as its rows land, under the factorisation that produces the next ones, and only
    // the last stage (a third of both products) is left on the chain.  Stage Jc: waves 0-2 form their row block of L(:, Jc), barrier,
    // the six lower blocks of D take the rank-16 update (dealt to the four waves).  z_k* is stored behind the last rows of W.
#pragma unroll
    for (int Jc = 0; Jc < 3; ++Jc) {
      if (wave < 3) {
        const int r = lane & 15, g = lane >> 4;
        const dbl4 cj = times_inverse_block<DAG>(sv, tile_k, XB, Jc, wave, lane);
#pragma unroll
        for (int v = 0; v < 4; ++v) LB[(16 * wave + g + 4 * v) * TP + 16 * Jc + r] = cj[v];
      } else if (Jc == 2 && lane < T) {
        const double* zk = sv.zv + (size_t)tile_k * T + lane;
        double z = ld<DAG>(zk);
        while (DAG && !filled(z)) { __builtin_amdgcn_s_sleep(2); z = ld<DAG>(zk); }
        zs[lane] = z;
      }
      if (Jc == 2 && DAG && threadIdx.x == 0 && s_trace_slot) s_trace_slot[5] = wall_clock64();   // trace: the last rows of W_k* are in registers
      lds_barrier();
      const int mi = lane & 15, mg = lane >> 4;
#pragma unroll
      for (int blk = 0; blk < 6; ++blk) {
        if ((blk & 3) != wave) continue;
        const int I = blk < 1 ? 0 : (blk < 3 ? 1 : 2), J = blk - (I * (I + 1)) / 2;
        dbl4 a4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 4 * Jc; kk < 4 * Jc + 4; ++kk)
          a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(LB[(16 * I + mi) * TP + 4 * kk + mg], LB[(16 * J + mi) * TP + 4 * kk + mg], a4, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * I + mg + 4 * v, c = 16 * J + mi;
          if (c <= r) D[r * TP + c] -= a4[v];
        }
      }
    }
    // b -= L z_k* by two lanes per row of waves 2 and 3, which only have one block each
    {
      if (tid >= 128 && tid < 128 + 2 * T) {
        const int row = (tid - 128) >> 1, h = tid & 1;
        double s0 = 0.0;
#pragma unroll
        for (int m = 0; m < T / 2; ++m) s0 += LB[row * TP + (T / 2) * h + m] * zs[(T / 2) * h + m];
        s0 += __shfl_xor(s0, 1, 64);
        if (h == 0) bvec[row] -= s0;
      }
    }
  }
  lds_barrier();
  diag_factor<DAG>(sv, tile_j, smem, tid);
}

template <bool DAG>
__device__ __forceinline__ void task_sub(const SolverDev& sv, const CholPlan& pl, int b, double* smem, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  double* X = smem;
  const int32_t* info = pl.sub_info + 4 * b;
  const int slot_ij = gl(info), slot_jj = gl(info + 1), part0 = gl(info + 2), nparts = gl(info + 3), tile_j = gl(pl.sub_col + b);
  const int p0 = gl(pl.sub_own + b), p1 = gl(pl.sub_ptr + b + 1);
  double sreg[9];
  const double* src = tile_ptr(sv, slot_ij);
#pragma unroll
  for (int q = 0; q < 9; ++q) sreg[q] = gl(src + tid + 256 * q);
  { double none = 0.0; subtract_partials<DAG, false>(sv, part0, nparts, sreg, none, tid); }
  Acc acc; acc.clear();
  double bz[3] = {0, 0, 0};
  accumulate<DAG, false>(sv, pl, pl.sub_list, p0, p1, acc, bz, wave, lane);
  CHOL_STAMP(3);
  acc.spill(smem + wave * kBuf, lane);
  lds_barrier();
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int e = tid + 256 * q, o = (e / T) * TP + e % T;
    sreg[q] -= (smem[o] + smem[kBuf + o]) + (smem[2 * kBuf + o] + smem[3 * kBuf + o]);
  }
  lds_barrier();
#pragma unroll
  for (int q = 0; q < 9; ++q) { const int e = tid + 256 * q; X[(e / T) * TP + e % T] = sreg[q]; }
  if (const int pub = gl(pl.sub_pub + b); pub >= 0) {   // the DIAG task of row i multiplies this by W_j itself (look-ahead on the critical path)
    double* xp = sv.Xpub + (size_t)pub * (T * T);
#pragma unroll
    for (int q = 0; q < 9; ++q) st<DAG>(xp + tid + 256 * q, sreg[q]);
  }
  CHOL_STAMP(4);
  lds_barrier();
  CHOL_STAMP(5);
  // L_ij = X W_j^T
  if (wave < 3) {
    const int I = wave, r = lane & 15, g = lane >> 4;
    double* out = factor_ptr(sv, slot_ij);
    dbl4 c[3];
    times_inverse_transposed<DAG>(sv, tile_j, X, c, wave, lane);
#pragma unroll
    for (int J = 0; J < 3; ++J)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        st<DAG>(out + (16 * I + g + 4 * v) * T + 16 * J + r, c[J][v]);
      }
  }
  CHOL_STAMP(6);
  (void)slot_jj;
}

template <bool DAG>
__device__ __forceinline__ void task_back(const SolverDev& sv, const CholPlan& pl, int b, double* smem, int tid) {
  double* part = smem;            // [10][T]
  double* tvec = smem + 10 * T;   // [T]
  const int32_t* list = pl.back_list;
  const int c2 = tid % 24, rg = tid / 24;   // column pair, row group (rg < 10 for tid < 240)
  const int tile_j = gl(pl.back_info + 2 * b + 1);
  const int p0 = gl(pl.back_ptr + b), p1 = gl(pl.back_ptr + b + 1);
  const bool worker = rg < 10;
  // rows of a 48 x 48 tile handled by this thread: rg, rg + 10, .. ; columns 2 c2, 2 c2 + 1 (threads without work
  // read nothing).  Whether the cells were all there is asked where the values are USED (gathered_ok), never next to
  // the loads: a test there would make the compiler wait for them on the spot and undo the prefetch.
  auto gather = [&](const double* tile, const double* y, double2 v[5], double yy[5], bool LDSY) {
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int r = rg + 10 * u;
      if (worker && r < T) {
        const double* q = tile + (size_t)r * T + 2 * c2;
        v[u] = make_double2(ld<DAG>(q), ld<DAG>(q + 1));
        yy[u] = LDSY ? y[r] : ld<DAG>(y + r);
      } else { v[u] = make_double2(0.0, 0.0); yy[u] = 0.0; }
    }
  };
  auto gathered_ok = [&](const double2 v[5], const double yy[5]) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < 5; ++u) ok = ok && filled(v[u].x) && filled(v[u].y) && filled(yy[u]);
    return ok;
  };
  // W_j and z_j (from the DIAG task of this column, long finished as a rule) are requested NOW and looked at behind the y_i the
  // task really waits for: their round trips used to sit between the last y_i and the store of y_j, on the chain of the backward solve
  const double* Wg = sv.Winv + (size_t)tile_j * (T * T);
  double2 wpre[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int r = rg + 10 * u;
    if (worker && r < T) { const double* q = Wg + (size_t)r * T + 2 * c2; wpre[u] = make_double2(ld<DAG>(q), ld<DAG>(q + 1)); } else wpre[u] = make_double2(0.0, 0.0);
  }
  double zpre = tid < T ? ld<DAG>(sv.zv + (size_t)tile_j * T + tid) : 0.0;
  double z2pre = (tid < T && sv.zv2) ? ld<DAG>(sv.zv2 + (size_t)tile_j * T + tid) : 0.0;
  double s0 = 0.0, s1 = 0.0;
  if (p0 < p1) {
    // L_ij (forward phase) is long finished; y_i is what the task waits for.  The host lists the tiles of the
    // column bottom-up, the order in which the y_i become available; loads run one group of two tiles ahead and a
    // thread whose cells of a group are not all there yet reads the group again.
    constexpr int kGroup = 1;
    struct Group { double2 v[kGroup][5]; double yy[kGroup][5]; };
    Group cur, nxt;
    auto fetch_group = [&](Group& g, int p) {
#pragma unroll
      for (int u = 0; u < kGroup; ++u) { const int q = min(p + u, p1 - 1); gather(factor_ptr(sv, gl(list + 2 * q)), sv.yv + (size_t)gl(list + 2 * q + 1) * T, g.v[u], g.yy[u], false); }
    };
    auto group_ok = [&](const Group& g) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < kGroup; ++u) ok = ok && gathered_ok(g.v[u], g.yy[u]);
      return ok;
    };
    auto mac_group = [&](const Group& g, int p) {
#pragma unroll
      for (int u = 0; u < kGroup; ++u)
        if (p + u < p1) {
#pragma unroll
          for (int k = 0; k < 5; ++k) { s0 += g.v[u][k].x * g.yy[u][k]; s1 += g.v[u][k].y * g.yy[u][k]; }
        }
    };
    fetch_group(cur, p0);
    for (int p = p0; p < p1; p += kGroup) {
      const int pn = p + kGroup;
      if (pn < p1) fetch_group(nxt, pn);
      if (DAG) {
        bool late = false;
        while (!group_ok(cur)) {
          watch_cell<DAG>(sv.yv + (size_t)gl(list + 2 * (min(p + kGroup, p1) - 1) + 1) * T);   // y of the last tile of the group
          fetch_group(cur, p);
          late = true;
        }
        if (late) note_late_input();
      }
      mac_group(cur, p);
      if (pn < p1) cur = nxt;
    }
  }
  if (worker) { part[rg * T + 2 * c2] = s0; part[rg * T + 2 * c2 + 1] = s1; }
  lds_barrier();
  CHOL_STAMP(3);
  if (tid < T) {
    double t = zpre;   // z_j, from the DIAG task of this column
    while (DAG && !filled(t)) { __builtin_amdgcn_s_sleep(8); t = ld<DAG>(sv.zv + (size_t)tile_j * T + tid); }
    if (sv.zv2) {   // two right-hand sides through one backward solve: L^T y = z - (s eta) z2 (the ETA task leaves s eta in its cell)
#pragma clang fp contract(off)
      double c = ld<DAG>(sv.ceta), z2 = z2pre;
      while (DAG && !filled(c)) { __builtin_amdgcn_s_sleep(8); c = ld<DAG>(sv.ceta); }
      while (DAG && !filled(z2)) { __builtin_amdgcn_s_sleep(8); z2 = ld<DAG>(sv.zv2 + (size_t)tile_j * T + tid); }
      t = t - c * z2;
    }
#pragma unroll
    for (int g = 0; g < 10; ++g) t -= part[g * T + tid];
    tvec[tid] = t;
  }
  lds_barrier();
  // y = W^T t
  {
    double2 v[5]; double yy[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) { const int r = rg + 10 * u; v[u] = wpre[u]; yy[u] = (worker && r < T) ? tvec[r] : 0.0; }
    while (DAG && !gathered_ok(v, yy)) { __builtin_amdgcn_s_sleep(8); gather(Wg, tvec, v, yy, true); }
    s0 = 0.0; s1 = 0.0;
#pragma unroll
    for (int u = 0; u < 5; ++u) { s0 += v[u].x * yy[u]; s1 += v[u].y * yy[u]; }
  }
  lds_barrier();
  if (worker) { part[rg * T + 2 * c2] = s0; part[rg * T + 2 * c2 + 1] = s1; }
  lds_barrier();
  if (tid < T) {
    double y = 0.0;
#pragma unroll
    for (int g = 0; g < 10; ++g) y += part[g * T + tid];
    st<DAG>(sv.yv + (size_t)tile_j * T + tid, y);
  }
  CHOL_STAMP(4);
}

// (defined with the substitution-only driver further down)
template <bool DAG>
__device__ __forceinline__ void task_forward(const SolverDev& sv, const CholPlan& pl, int d, int p0, int p1, const double* __restrict__ b2, const double* __restrict__ minus, double* z2, double* smem, int tid,
                                             double* partial_out = nullptr);
template <bool DAG>
__device__ __forceinline__ void task_eta(const SolverDev& sv, const CholPlan& pl, double* smem, int tid);

template <bool DAG>
__device__ __forceinline__ void run_task(const SolverDev& sv, const CholPlan& pl, int kind, int item, double* smem, int tid) {
  switch (kind) {
    case kTaskUpdate: task_update<DAG>(sv, pl, item, smem, tid); break;
    case kTaskDiag: task_diag<DAG>(sv, pl, item, smem, tid); break;
    case kTaskSub: task_sub<DAG>(sv, pl, item, smem, tid); break;
    case kTaskFwd2: { const int tr = gl(pl.diag_toprow + item);
      task_forward<DAG>(sv, pl, item, gl(pl.fwd_range + 2 * item), gl(pl.fwd_range + 2 * item + 1), sv.border2, (pl.fwd2_minus && tr >= 0) ? pl.fwd2_minus + (size_t)tr * T : nullptr, sv.zv2, smem, tid); } break;
    case kTaskFwd2P: task_forward<DAG>(sv, pl, item, gl(pl.fwd_range + 2 * item), gl(pl.fwd_range + 2 * item + 1), sv.border2, nullptr, sv.zv2, smem, tid, pl.fwd2_partial + (size_t)gl(pl.diag_toprow + item) * T); break;
    case kTaskEta: task_eta<DAG>(sv, pl, smem, tid); break;
    default: task_back<DAG>(sv, pl, item, smem, tid); break;
  }
}

// one launch per (level, kind): items first .. first + gridDim.x of one kind
__global__ __launch_bounds__(256) void chol_level_kernel(const SolverDev sv, const CholPlan pl, int kind, int first) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  run_task<false>(sv, pl, kind, first + blockIdx.x, smem, threadIdx.x);
}

// the whole factorisation + both triangular solves in one persistent launch
__global__ __launch_bounds__(256, 2) void chol_dag_kernel(const DagArgs* __restrict__ args) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int s_ticket;
  const int tid = threadIdx.x;
  const CholPlan& pl = args->pl;
  if (lm_stopped(args->sv.ctl)) return;   // (device-side trust region: the solve is over, iterations enqueued ahead fall through)
  for (;;) {
    lds_barrier();   // the previous task's LDS is free, s_ticket has been read by everyone
    if (tid == 0) {
      const int tk = (int)__hip_atomic_fetch_add(pl.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_ticket = tk;
      s_trace_slot = (pl.trace && tk < pl.ntasks) ? pl.trace + 8 * (size_t)tk : nullptr;
      if (s_trace_slot) { s_trace_slot[0] = blockIdx.x; s_trace_slot[1] = s_trace_slot[2] = wall_clock64(); }
    }
    lds_barrier();
    const int t = s_ticket;
    if (t >= pl.ntasks) {
      // The last workgroup to leave re-arms the ticket counters for the next launch (a 16-byte fill per solve was a launch of its
      // own: ~6 us of kernel boundary on a solve of a few hundred): counter [3] counts the leavers.
      if (tid == 0 && __hip_atomic_fetch_add(pl.ticket + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __hip_atomic_store(pl.ticket + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pl.ticket + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    // every task recomputes what it derives from the thread index: left alone, the compiler hoists dozens of per-thread LDS
    // offsets out of this loop, keeps them live across all task bodies and spills them to scratch — whose reloads (vmcnt)
    // then stall on every prefetch in flight
    int task_tid = tid;
    asm volatile("" : "+v"(task_tid));
    run_task<true>(args->sv, pl, gl(pl.tasks + 2 * t), gl(pl.tasks + 2 * t + 1), smem, task_tid);
    if (tid == 0 && s_trace_slot) s_trace_slot[7] = wall_clock64();
  }
}

// ---- one more right-hand side through a finished factorisation -----------------------------------------------------
// S v = b2 with the factor tiles and the W_j the last solve left (the free interFrameRatio's border column, the columns of a
// covariance block): forward tasks z_j = W_j (b2_j - sum_k L_jk z_k) in the order of the DIAG items, then the BACK tasks of the
// factorisation on z2 / y2 — one persistent launch over write-once z / y cells of its own, no tile product anywhere.  It used to be a
// second factorisation (0.76 ms at 1k cameras for 0.1 ms of substitutions).
// z2_j = W_j (b2_j - minus_j - sum_{p in [p0, p1)} L_jk z2_k) over the contributors p of DIAG item d's list: thread (row r = tid % T, column
// group cg = tid / T < 5) takes ten columns of its row of every tile — 240 threads, ten loads each per contributor, the next contributor's
// tile travelling while this one is multiplied.  Everything it reads is a write-once cell: beside a running factorisation (FWD2 tasks of
// the persistent driver) the rows of L_jk, z2_k and W_j are polled; behind a finished one (chol_solve_kernel, the level schedule) they are
// simply there.  minus (may be null): what other ranks' columns contribute to this column (sharded factorisation: summed by the exchange).
template <bool DAG>
__device__ __forceinline__ void task_forward(const SolverDev& sv, const CholPlan& pl, int d, int p0, int p1, const double* __restrict__ b2, const double* __restrict__ minus, double* z2,
                                             double* smem, int tid, double* partial_out) {
  constexpr int CG = 10;        // columns per thread
  double* zb = smem;            // [T] z2 of the contributor being multiplied
  double* sp = smem + T;        // [5][T] partial sums
  double* tv = smem + 6 * T;    // [T]
  const int cg = tid / T, r = tid % T, c0 = CG * cg;
  const bool worker = cg < 5;
  const int tile_j = gl(pl.diag_info + 4 * d + 1);
  auto load_row = [&](const double* tile, double v[CG]) {
#pragma unroll
    for (int m = 0; m < CG; ++m) v[m] = (worker && c0 + m < T) ? ld<DAG>(tile + (size_t)r * T + c0 + m) : 0.0;
  };
  auto row_ok = [&](const double v[CG]) { bool ok = true;
#pragma unroll
    for (int m = 0; m < CG; ++m) ok = ok && filled(v[m]);
    return ok; };
  double s = 0.0;
  double cur[CG], nxt[CG];
  if (p0 < p1) load_row(factor_ptr(sv, gl(pl.diag_list + 2 * p0)), cur);
  for (int p = p0; p < p1; ++p) {
    if (p + 1 < p1) load_row(factor_ptr(sv, gl(pl.diag_list + 2 * (p + 1))), nxt);
    if (tid < T) {
      const double* zk = z2 + (size_t)gl(pl.diag_list + 2 * p + 1) * T + tid;
      double z = ld<DAG>(zk);
      while (DAG && !filled(z)) { __builtin_amdgcn_s_sleep(2); z = ld<DAG>(zk); }
      zb[tid] = z;
    }
    while (DAG && !row_ok(cur)) { __builtin_amdgcn_s_sleep(2); load_row(factor_ptr(sv, gl(pl.diag_list + 2 * p)), cur); }
    lds_barrier();
#pragma unroll
    for (int m = 0; m < CG; ++m) if (worker && c0 + m < T) s += cur[m] * zb[c0 + m];
    lds_barrier();
#pragma unroll
    for (int m = 0; m < CG; ++m) cur[m] = nxt[m];
  }
  if (partial_out) {   // FWD2P: only the sum over this rank's part — it travels (plain stores: read behind this launch)
    if (worker) sp[cg * T + r] = s;
    lds_barrier();
    if (tid < T) { double t = 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) t += sp[k * T + r];
      partial_out[r] = t; }
    return;
  }
  // W_j: requested before the sums meet (the DIAG task of this column has just produced it, or is about to)
  const double* Wg = sv.Winv + (size_t)tile_j * (T * T);
  double w[CG];
  load_row(Wg, w);
  if (worker) sp[cg * T + r] = s;
  lds_barrier();
  if (tid < T) {
    double t = gl(b2 + (size_t)tile_j * T + r);
    if (minus) t -= gl(minus + r);
#pragma unroll
    for (int k = 0; k < 5; ++k) t -= sp[k * T + r];
    tv[r] = t;
  }
  while (DAG && !row_ok(w)) { __builtin_amdgcn_s_sleep(2); load_row(Wg, w); }
  lds_barrier();
  s = 0.0;
#pragma unroll
  for (int m = 0; m < CG; ++m) if (worker && c0 + m < T && c0 + m <= r) s += w[m] * tv[c0 + m];   // (W is lower triangular: exact zeros right of the diagonal)
  lds_barrier();
  if (worker) sp[cg * T + r] = s;
  lds_barrier();
  if (tid < T) {
    double z = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) z += sp[k * T + r];
    st<DAG>(z2 + (size_t)tile_j * T + r, z);
  }
}

// The ratio's step from the two forward solves (solver_state.hpp): d1 = z2.z, d2 = z2.z2 over the columns [t0, t1) of this plan (+ what
// `extra` carries: sharded factorisation — the other ranks' parts, summed by the exchange), eta = (g_s - s d1) / (h_s + D - s^2 d2); s eta goes to
// the cell the BACK tasks wait for.  One workgroup, sums in a fixed order.
template <bool DAG>
__device__ __forceinline__ void task_eta(const SolverDev& sv, const CholPlan& pl, double* smem, int tid) {
#pragma clang fp contract(off)
  double a = 0.0, b = 0.0;
  for (int64_t t = tid; t < sv.npad; t += 256) {
    if (pl.eta_tiles && !gl(pl.eta_tiles + t / T)) continue;   // (a sharded factorisation: this launch's columns only)
    double z = ld<DAG>(sv.zv + t), w = ld<DAG>(sv.zv2 + t);
    while (DAG && !(filled(z) && filled(w))) { __builtin_amdgcn_s_sleep(8); z = ld<DAG>(sv.zv + t); w = ld<DAG>(sv.zv2 + t); }
    a += w * z; b += w * w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
  if ((tid & 63) == 0) { smem[tid >> 6] = a; smem[4 + (tid >> 6)] = b; }
  lds_barrier();
  if (tid == 0) {
    double d1 = (smem[0] + smem[1]) + (smem[2] + smem[3]), d2 = (smem[4] + smem[5]) + (smem[6] + smem[7]);
    if (pl.eta_partial) { pl.eta_partial[0] = d1; pl.eta_partial[1] = d2; return; }   // launch A of a sharded factorisation: this rank's part, to be summed by the exchange
    if (pl.eta_extra) { d1 += gl(pl.eta_extra); d2 += gl(pl.eta_extra + 1); }        // launch B: the separators' share (every rank alike) + all parts'
    const double sc = sv.rt[kRtScale];
    const double eta = (sv.rt[kRtGs] - sc * d1) / (sv.rt[kRtDiagTerm] - sc * sc * d2);
    const double c = sc * eta;
    sv.rt[kRtDot1] = d1; sv.rt[kRtDot2] = d2; sv.rt[kRtEta] = eta; sv.rt[kRtC] = c;
    st<DAG>(sv.ceta, c);
  }
}

// the same tasks one launch per level (no polling): what the persistent form is checked against and falls back to
__global__ __launch_bounds__(256) void chol_solve_level_kernel(const SolverDev sv, const CholPlan pl, int backward, int first, const double* __restrict__ b2, double* zy2) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (!backward) { const int d = first + blockIdx.x; task_forward<false>(sv, pl, d, pl.diag_ptr[d], pl.diag_ptr[d + 1], b2, nullptr, zy2, smem, threadIdx.x); }
  else {
    SolverDev sv2 = sv;
    sv2.zv = zy2; sv2.yv = zy2 + sv2.npad; sv2.zv2 = nullptr;
    task_back<false>(sv2, pl, first + blockIdx.x, smem, threadIdx.x);
  }
}

__global__ __launch_bounds__(256) void chol_solve_kernel(const DagArgs* __restrict__ args, const double* __restrict__ b2, double* zy2, unsigned int* ticket) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int s_ticket;
  const int tid = threadIdx.x;
  const CholPlan& pl = args->pl;
  const int nd = pl.ndiag;
  for (;;) {
    lds_barrier();
    if (tid == 0) { s_ticket = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_trace_slot = nullptr; }
    lds_barrier();
    const int t = s_ticket;
    if (t >= 2 * nd) return;
    int task_tid = tid;
    asm volatile("" : "+v"(task_tid));
    if (t < nd) task_forward<true>(args->sv, pl, t, gl(pl.diag_ptr + t), gl(pl.diag_ptr + t + 1), b2, nullptr, zy2, smem, task_tid);
    else {
      SolverDev sv2 = args->sv;
      sv2.zv = zy2; sv2.yv = zy2 + sv2.npad; sv2.zv2 = nullptr;
      task_back<true>(sv2, pl, gl(pl.tasks + 2 * (pl.ntasks - 2 * nd + t) + 1), smem, task_tid);   // (the BACK tasks close the task list, in their order)
    }
  }
}

// ---- verification of the persistent driver's result -------------------------------------------------------------
// The task-DAG kernel has no safety net inside: its hand-offs are write-once cells found by polling.  What it returns is
// therefore checked against the system it was asked to solve: res = rhs - S y over the packed tiles (S symmetric, lower
// tiles stored), with den = |rhs| + |S| |y| as the yardstick of a backward-stable solve.  A stale or torn cell anywhere
// in the factorisation shows up here as a residual many orders above rounding; the solver then repeats the solve on the
// level schedule (solver.hip).  Sums are fp64 atomics in arbitrary order: this is a check, not a result.
// res / den start out zero (the check kernel re-arms them after it has looked): one workgroup per packed tile adds what the
// tile contributes — thread (r, q) = (tid / 4, tid % 4) takes a quarter of row r resp. column r.
__global__ __launch_bounds__(256) void chol_residual_kernel(const SolverDev sv, const int32_t* __restrict__ slot_tiles, const double* __restrict__ b_rhs, double* res, double* den) {
  __shared__ double tile[T * TP];
  __shared__ double yj[T], yi[T];
  const int slot = blockIdx.x, tid = threadIdx.x, r = tid >> 2, q = tid & 3;
  const int tile_i = slot_tiles[2 * slot], tile_j = slot_tiles[2 * slot + 1];
  auto quad_sum = [](double v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); return v; };
  if (tid < T) { yj[tid] = sv.yv[(size_t)tile_j * T + tid]; yi[tid] = sv.yv[(size_t)tile_i * T + tid]; }
  const double* St = tile_ptr(sv, slot);
  for (int e = tid; e < T * T; e += 256) tile[(e / T) * TP + e % T] = St[e];
  __syncthreads();
  if (r >= T) return;
  if (tile_i == tile_j) {   // a diagonal tile through its lower triangle (what the factorisation reads); the rhs enters here
    double s = 0.0, a = 0.0;
    for (int c = 12 * q; c < 12 * q + 12; ++c) { const double v = tile[max(r, c) * TP + min(r, c)]; s += v * yj[c]; a += fabs(v) * fabs(yj[c]); }
    s = quad_sum(s); a = quad_sum(a);
    if (q == 0) { const double b = b_rhs[(size_t)tile_j * T + r]; atomicAdd(res + (size_t)tile_j * T + r, b - s); atomicAdd(den + (size_t)tile_j * T + r, fabs(b) + a); }
    return;
  }
  double s = 0.0, a = 0.0, st = 0.0, at = 0.0;
  for (int c = 12 * q; c < 12 * q + 12; ++c) {
    const double v = tile[r * TP + c]; s += v * yj[c]; a += fabs(v) * fabs(yj[c]);          // row r of tile (i, j) against y_j
    const double w = tile[c * TP + r]; st += w * yi[c]; at += fabs(w) * fabs(yi[c]);        // column r of it against y_i
  }
  s = quad_sum(s); a = quad_sum(a); st = quad_sum(st); at = quad_sum(at);
  if (q == 0) {
    atomicAdd(res + (size_t)tile_i * T + r, -s); atomicAdd(den + (size_t)tile_i * T + r, a);
    atomicAdd(res + (size_t)tile_j * T + r, -st); atomicAdd(den + (size_t)tile_j * T + r, at);
  }
}
// flag = 1 when some |res_t| exceeds tol * den_t (NaN included) although no pivot failed; res / den are zeroed for the next solve.
// The flag is STICKY: a clean solve leaves it alone, so that several solves between two host reads (the free interFrameRatio's
// second right-hand side, the columns of a covariance block) cannot mask each other; the host clears it before the first one.
__global__ __launch_bounds__(1024) void chol_residual_check_kernel(const SolverDev sv, double* res, double* den, double tol, double* flag, const uint8_t* __restrict__ row_mine) {
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  bool bad = false;
  for (int64_t t = threadIdx.x; t < sv.npad; t += 1024) {
    const double rr = fabs(res[t]), d = den[t];
    if (!(rr <= tol * d) && !(rr == 0.0) && (!row_mine || row_mine[t / T])) bad = true;
    res[t] = 0.0; den[t] = 0.0;
  }
  if (bad) s_bad = 1;
  __syncthreads();
  if (threadIdx.x == 0 && s_bad && *sv.chol_fail == 0) *flag = 1.0;   // a failed pivot is reported through chol_fail, not here
}

}  // namespace

hipError_t launch_chol_verify(const SolverDev& sv, const int32_t* slot_tiles, const double* b_rhs, double* res, double* den, double tol, double* flag, hipStream_t st, const uint8_t* row_mine) {
  hipLaunchKernelGGL(chol_residual_kernel, dim3(sv.nslots), dim3(256), 0, st, sv, slot_tiles, b_rhs, res, den);
  hipLaunchKernelGGL(chol_residual_check_kernel, dim3(1), dim3(1024), 0, st, sv, res, den, tol, flag, row_mine);
  return hipGetLastError();
}

hipError_t launch_chol_level(const SolverDev& sv, const CholPlan& pl, int kind, int first, int count, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  hipError_t e = allow_dynamic_lds(chol_level_kernel, kCholLds * sizeof(double));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(chol_level_kernel, dim3(count), dim3(256), kCholLds * sizeof(double), st, sv, pl, kind, first);
  return hipGetLastError();
}

// zy2: [2][npad] doubles (z | y of this right-hand side; y = the solution), ticket: one counter of the caller's
hipError_t launch_chol_solve(const SolverDev& sv, const CholPlan& pl, const DagArgs* device_args, const double* b2, double* zy2, unsigned int* ticket, int workgroups, hipStream_t st) {
  if (pl.ndiag <= 0) return hipSuccess;
  hipError_t e = allow_dynamic_lds(chol_solve_kernel, (size_t)84 * 1024);   // (one workgroup per CU, as the factorisation: polling waves do not share SIMDs with the waves they wait for)
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(zy2, 0xFF, 2 * (size_t)sv.npad * sizeof(double), st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(ticket, 0, sizeof(unsigned int), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(chol_solve_kernel, dim3(workgroups), dim3(256), (size_t)84 * 1024, st, device_args, b2, zy2, ticket);
  return hipGetLastError();
}

hipError_t launch_chol_solve_level(const SolverDev& sv, const CholPlan& pl, bool backward, int first, int count, const double* b2, double* zy2, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  hipError_t e = allow_dynamic_lds(chol_solve_level_kernel, kCholLds * sizeof(double));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(chol_solve_level_kernel, dim3(count), dim3(256), kCholLds * sizeof(double), st, sv, pl, backward ? 1 : 0, first, b2, zy2);
  return hipGetLastError();
}

hipError_t launch_chol_dag(const SolverDev& sv, const CholPlan& pl, const DagArgs* device_args, int workgroups, bool one_per_cu, hipStream_t st) {
  if (pl.ntasks <= 0) return hipSuccess;
  // The kernel needs 78 KB of LDS and would fit a CU twice — and two workgroups sharing a CU (polling waves on the SIMDs of the
  // waves they wait for) slow the solve down by orders of magnitude.  The production launch therefore asks for more than half
  // of a CU's 160 KB: the dispatcher cannot co-locate two of them, whatever else runs on the device.
  const size_t lds_bytes = one_per_cu ? (size_t)84 * 1024 : kCholLds * sizeof(double);
  static_assert(kCholLds * sizeof(double) <= (size_t)84 * 1024, "LDS map of the Cholesky tasks");
  hipError_t e = allow_dynamic_lds(chol_dag_kernel, lds_bytes);
  if (e != hipSuccess) return e;
  // (the ticket counters are zero: set so when the plan was made, and re-armed by the last workgroup of every launch; the caller has
  // re-armed the write-once cells: one memset over their common allocation, solver.hip)
  hipLaunchKernelGGL(chol_dag_kernel, dim3(workgroups), dim3(256), lds_bytes, st, device_args);
  return hipGetLastError();
}

}  // namespace rsba

#ifdef RSBA_TEST_HOOKS
// instrumented build only (not part of include/rsba_amd.h): the coherent-read counters of the persistent Cholesky kernel, since the last reset
extern "C" int32_t rsba_debug_chol_coherent(unsigned long long out[8], int32_t reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rsba::g_chol_coherent), 8 * sizeof(unsigned long long)) != hipSuccess) return RSBA_ERR_HIP;
  if (reset) { const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rsba::g_chol_coherent), zero, sizeof zero) != hipSuccess) return RSBA_ERR_HIP; }
  return RSBA_OK;
}
#endif

