// Probe for the register-blocked form of the Schur kernel's loop (Next 1): does a wave that keeps the accumulators of a
// 2 x 2 block of tile pairs — 36 blocks of 16 x 16, 288 registers, ONE wave per SIMD — keep the fp64 matrix pipe busier than the
// shipped loop (one tile pair per wave, 9 blocks, two waves per SIMD)?  Both loops here take FACTORED groups (80 doubles: the layout of
// solver_state.hpp, kGroupFactored) from a buffer of the C4 size (585 k groups, 374 MB) through tables of group offsets staged in LDS,
// form the operand rows in registers and issue every MFMA behind a scalar test of a mask bit, like kernels_normal.hip: schur_chunk.
//   pair  form: per group of four entries 2 sides x 8 loaded doubles -> 27 MFMAs   (1.69 MFMAs per loaded double, 24 fp64 vector ops)
//   block form: per group of four entries 4 sides x 8 loaded doubles -> 108 MFMAs  (3.38 MFMAs per loaded double, 48 fp64 vector ops)
// The same number of MFMAs in total.  Prints time, TFLOP/s, cycles per MFMA of a wave and the shader clock inside the loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/schur_block_probe.hip -o tools/schur_block_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double dbl4 __attribute__((ext_vector_type(4)));
constexpr int kGroup = 80;

struct Lane { int main, left, tau_m, tau_l; double a0, b0, a1, b1, a2, b2; };
__device__ __forceinline__ Lane lane_consts(int r) {
  Lane f; const int sl = 16 + (r & 7);
  f.main = r; f.left = 48 + (r & 7); f.tau_m = 72 + r / 6; f.tau_l = 72 + sl / 6;
  f.a0 = 1.0; f.b0 = -1.0; f.a1 = 0.0; f.b1 = 1.0;
  const bool p1 = (r >> 3) != 0;
  f.a2 = p1 ? 0.0 : 1.0; f.b2 = p1 ? 1.0 : -1.0;
  return f;
}
struct Side { double qm[3], ql[3], tm, tl; };
__device__ __forceinline__ void fetch_side(const double* __restrict__ p, const Lane& fl, Side& S) {
  S.tm = p[fl.tau_m]; S.tl = p[fl.tau_l];
#pragma unroll
  for (int t = 0; t < 3; ++t) { S.qm[t] = p[fl.main + 16 * t]; S.ql[t] = p[fl.left + 8 * t]; }
}
__device__ __forceinline__ void weights(const Side& S, const Lane& fl, double w[3]) {
  w[0] = __builtin_fma(fl.b0, S.tm, fl.a0); w[1] = __builtin_fma(fl.b1, S.tm, fl.a1); w[2] = __builtin_fma(fl.b2, S.tl, fl.a2);
}
template <int ABL = 0>
__device__ __forceinline__ void operands(const Side& S, int t, const double w[3], double out[3]) {
  if constexpr (ABL & 2) { out[0] = S.qm[t]; out[1] = S.tm; out[2] = S.ql[t]; }
  else { out[0] = S.qm[t] * w[0]; out[1] = S.qm[t] * w[1]; out[2] = S.ql[t] * w[2]; }
}

// NS sides per entry: 2 (pair form: acc[1][1]) or 4 (block form: acc[2][2]); table [nq][4 lane groups][NS] offsets + [nq] masks per wave, staged in LDS
// ASM: every MFMA as inline assembly with the accumulator's register file fixed — 32 of the block form's 36 accumulators in AGPRs (all 256 of them), the last four in
// VGPRs: left to itself the compiler puts all 36 into AGPR form and moves the four that do not fit in and out around every group (v_accvgpr_read / write)
template <int NB, int ASM>
__device__ __forceinline__ void mfma(dbl4& c, double a, double b) {
  if constexpr (ASM == 0) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  else if constexpr (NB < 32) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int NX, int ASM, class Acc>
__device__ __forceinline__ void mfma_at(Acc& acc, int x, int y, int I, int J, double a, double b) {
  // (x, y, I, J are compile-time constants after unrolling; the switch folds)
#define RSBA_CASE(n) case n: mfma<n, ASM>(acc[(n) / 18 % NX][(n) / 9 % NX][(n) / 3 % 3][(n) % 3], a, b); break;
  switch (((x * 2 + y) * 3 + I) * 3 + J) {
    RSBA_CASE(0) RSBA_CASE(1) RSBA_CASE(2) RSBA_CASE(3) RSBA_CASE(4) RSBA_CASE(5) RSBA_CASE(6) RSBA_CASE(7) RSBA_CASE(8) RSBA_CASE(9) RSBA_CASE(10) RSBA_CASE(11)
    RSBA_CASE(12) RSBA_CASE(13) RSBA_CASE(14) RSBA_CASE(15) RSBA_CASE(16) RSBA_CASE(17) RSBA_CASE(18) RSBA_CASE(19) RSBA_CASE(20) RSBA_CASE(21) RSBA_CASE(22) RSBA_CASE(23)
    RSBA_CASE(24) RSBA_CASE(25) RSBA_CASE(26) RSBA_CASE(27) RSBA_CASE(28) RSBA_CASE(29) RSBA_CASE(30) RSBA_CASE(31) RSBA_CASE(32) RSBA_CASE(33) RSBA_CASE(34) RSBA_CASE(35)
  }
#undef RSBA_CASE
}
// ASM (mode): 0 = builtin MFMAs; 1 = pinned accumulators; 2 = pinned + the operand arithmetic of a coordinate kept together in front of its MFMAs (scheduling barriers:
// left alone, the scheduler spreads the fp64 multiplies between the MFMAs, and fp64 vector instructions and fp64 MFMAs share one datapath); 3 = pinned + ALL operands
// of a group of four entries formed in one batch.  epi > 0: after every `epi` groups of a wave the four waves' accumulators meet in LDS tile pair by tile pair and the
// partial tiles are stored (the chunk epilogue of the real kernel), then the accumulators start from zero.
template <int NS, int kDepth, int kWaves, int ASM = 0, int ABL = 0>
__global__ __launch_bounds__(256, kWaves) void loop_kernel(const double* __restrict__ Pm, const uint32_t* __restrict__ tab, const unsigned long long* __restrict__ msk, int nq, double* __restrict__ out, long long* clk, int epi, double* __restrict__ part) {
  constexpr int NX = NS / 2;   // tiles per side of the block
  extern __shared__ uint32_t s_tab[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const size_t wg = (size_t)blockIdx.x * 4 + wave;
  uint32_t* my = s_tab + (size_t)wave * nq * 4 * NS;
  for (int i = lane; i < nq * 4 * NS; i += 64) my[i] = tab[wg * nq * 4 * NS + i];
  unsigned long long* mym = reinterpret_cast<unsigned long long*>(s_tab + (size_t)4 * nq * 4 * NS) + (size_t)wave * nq;
  for (int i = lane; i < nq; i += 64) mym[i] = msk[wg * nq + i];
  __syncthreads();
  const Lane fl = lane_consts(r);
  dbl4 acc[NX][NX][3][3];
#pragma unroll
  for (int x = 0; x < NX; ++x)
#pragma unroll
    for (int y = 0; y < NX; ++y)
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J) acc[x][y][I][J] = dbl4{0, 0, 0, 0};
  struct Group { Side s[NS]; };
  Group ring[kDepth];
  auto fetch = [&](int q, Group& G) {
    const int qq = q < nq ? q : nq - 1;
    const uint32_t* o = my + ((size_t)qq * 4 + g) * NS;
    if constexpr (NS == 2) { const uint2 v = *reinterpret_cast<const uint2*>(o); fetch_side(Pm + v.x, fl, G.s[0]); fetch_side(Pm + v.y, fl, G.s[1]); }
    else { const uint4 v = *reinterpret_cast<const uint4*>(o); fetch_side(Pm + v.x, fl, G.s[0]); fetch_side(Pm + v.y, fl, G.s[1]); fetch_side(Pm + v.z, fl, G.s[2]); fetch_side(Pm + v.w, fl, G.s[3]); }
  };
  const long long w0 = wall_clock64(), c0 = clock64();
#pragma unroll
  for (int d = 0; d < kDepth; ++d) fetch(d, ring[d]);
  const int per = epi > 0 ? epi : nq;   // groups of a wave per chunk (a multiple of kDepth)
  for (int c0 = 0; c0 < nq; c0 += per) {
  for (int base = c0; base < c0 + per; base += kDepth) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      if (base + d < nq) {
        const unsigned long long m = mym[base + d];
        // the nine bits of every tile pair once per coordinate: every MFMA tests a bit of its own (kernels_normal.hip: the same bit tested three times is kept as a lane mask)
        unsigned m27[NX][NX];
#pragma unroll
        for (int x = 0; x < NX; ++x)
#pragma unroll
          for (int y = 0; y < NX; ++y) m27[x][y] = ((unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((m >> (9 * (2 * x + y))) & 0x1FFu))) * 0x40201u;
        const Group& G = ring[d];
        double w[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s) { if constexpr (ABL & 2) { w[s][0] = w[s][1] = w[s][2] = 0; } else weights(G.s[s], fl, w[s]); }
        double a[3][NX][3], b[3][NX][3];
        if constexpr (ASM == 3) {
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int x = 0; x < NX; ++x) { operands<ABL>(G.s[x], t, w[x], a[t][x]); operands<ABL>(G.s[NX + x], t, w[NX + x], b[t][x]); }
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 1" ::: "memory");
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          if constexpr (ASM != 3) {
            if constexpr (ASM == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < NX; ++x) { operands<ABL>(G.s[x], t, w[x], a[t][x]); operands<ABL>(G.s[NX + x], t, w[NX + x], b[t][x]); }
            if constexpr (ASM == 2) __builtin_amdgcn_sched_barrier(0);
            if constexpr (ASM != 0) asm volatile("s_nop 1" ::: "memory");   // (VALU write -> MFMA read of the operands: the compiler cannot see into the asm statements)
          }
#pragma unroll
          for (int x = 0; x < NX; ++x)
#pragma unroll
            for (int y = 0; y < NX; ++y)
#pragma unroll
              for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int J = 0; J < 3; ++J)
                  if ((ABL & 4) || __builtin_expect(((m27[x][y] >> (9 * t + 3 * I + J)) & 1u) != 0u, 1)) mfma_at<NX, ASM>(acc, x, y, I, J, a[t][x][I], b[t][y][J]);
        }
      }
      if constexpr (!(ABL & 1)) fetch(base + d + kDepth, ring[d]);
    }
  }
  if (epi > 0) {   // the chunk's epilogue
    if constexpr (ASM != 0) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double* red = reinterpret_cast<double*>(s_tab + 16384);   // behind the tables
#pragma unroll
    for (int x = 0; x < NX; ++x)
#pragma unroll
      for (int y = 0; y < NX; ++y) {
        __syncthreads();
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
          for (int J = 0; J < 3; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) { red[wave * (48 * 49) + (16 * I + g + 4 * v) * 49 + 16 * J + r] = acc[x][y][I][J][v]; acc[x][y][I][J][v] = 0.0; }
        __syncthreads();
        double* dst = part + ((size_t)blockIdx.x * 4 + 2 * x + y) * 2304;
        for (int e = tid; e < 2304; e += 256) { const int o = (e / 48) * 49 + e % 48; dst[e] = (red[o] + red[48 * 49 + o]) + (red[2 * 48 * 49 + o] + red[3 * 48 * 49 + o]); }
      }
  }
  }
  const long long w1 = wall_clock64(), c1 = clock64();
  if constexpr (ASM != 0) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (MFMA write -> VALU read of the accumulators)
  double s = 0;
#pragma unroll
  for (int x = 0; x < NX; ++x)
#pragma unroll
    for (int y = 0; y < NX; ++y)
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J) s += acc[x][y][I][J][0] + acc[x][y][I][J][1] + acc[x][y][I][J][2] + acc[x][y][I][J][3];
  out[(size_t)blockIdx.x * 256 + tid] = s;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = w1 - w0; clk[1] = c1 - c0; }
}


// The block form scheduled by hand: every MFMA an asm statement that also clobbers memory, so that the loads written between two of
// them STAY there — one load behind every second or fourth MFMA instead of 32 in a clump at the end of the group (a wave issues in
// order: a clump of loads is ~500 cycles in which the matrix pipe runs ONE MFMA).  The group two steps ahead lands in the ring slot
// that is being consumed, field by field as the fields die: taus and coordinate 0 during the MFMAs of coordinate 0, and so on.
// Table cells (group offsets, masks) are read from LDS one group ahead.  FAST: a tile pair whose nine blocks are all present runs its
// nine MFMAs of a coordinate without the per-MFMA tests.
template <int NB>
__device__ __forceinline__ void mfma_mem(dbl4& c, double a, double b) {
  if constexpr (NB < 32) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b) : "memory");
  else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "memory");
}
// the load that follows MFMA k (0..35) of coordinate T: a field of this slot that is dead by now, for the group two steps ahead
template <int T, int K>
__device__ __forceinline__ void load_behind(Side* G, const double* const p[4], const Lane& fl) {
  if constexpr (T == 0) {
    if constexpr (K % 2 == 0 && K < 32) {
      constexpr int i = K / 2, sd = i / 4, wh = i % 4;
      if constexpr (wh == 0) G[sd].tm = p[sd][fl.tau_m]; else if constexpr (wh == 1) G[sd].tl = p[sd][fl.tau_l]; else if constexpr (wh == 2) G[sd].qm[0] = p[sd][fl.main]; else G[sd].ql[0] = p[sd][fl.left];
    }
  } else if constexpr (K % 4 == 0 && K < 32) {
    constexpr int i = K / 4, sd = i / 2, wh = i % 2;
    if constexpr (wh == 0) G[sd].qm[T] = p[sd][fl.main + 16 * T]; else G[sd].ql[T] = p[sd][fl.left + 8 * T];
  }
}
template <int T, int XY, int IJ, bool TESTED, class Acc>
__device__ __forceinline__ void one_mfma(Acc& acc, unsigned m27, const double a[2][3], const double b[2][3], Side* G, const double* const p[4], const Lane& fl) {
  constexpr int x = XY / 2, y = XY % 2, I = IJ / 3, J = IJ % 3;
  if (!TESTED || __builtin_expect(((m27 >> (9 * T + IJ)) & 1u) != 0u, 1)) mfma_mem<18 * x + 9 * y + IJ>(acc[x][y][I][J], a[x][I], b[y][J]);
  load_behind<T, 9 * XY + IJ>(G, p, fl);
}
template <int T, int XY, bool TESTED, class Acc>
__device__ __forceinline__ void nine_mfmas(Acc& acc, unsigned m27, const double a[2][3], const double b[2][3], Side* G, const double* const p[4], const Lane& fl) {
  one_mfma<T, XY, 0, TESTED>(acc, m27, a, b, G, p, fl); one_mfma<T, XY, 1, TESTED>(acc, m27, a, b, G, p, fl); one_mfma<T, XY, 2, TESTED>(acc, m27, a, b, G, p, fl);
  one_mfma<T, XY, 3, TESTED>(acc, m27, a, b, G, p, fl); one_mfma<T, XY, 4, TESTED>(acc, m27, a, b, G, p, fl); one_mfma<T, XY, 5, TESTED>(acc, m27, a, b, G, p, fl);
  one_mfma<T, XY, 6, TESTED>(acc, m27, a, b, G, p, fl); one_mfma<T, XY, 7, TESTED>(acc, m27, a, b, G, p, fl); one_mfma<T, XY, 8, TESTED>(acc, m27, a, b, G, p, fl);
}
template <int T, bool FAST, class Acc>
__device__ __forceinline__ void coordinate(Acc& acc, const unsigned m9[4], const double w[4][3], Side* G, const double* const p[4], const Lane& fl) {
  double a[2][3], b[2][3];
#pragma unroll
  for (int x = 0; x < 2; ++x) { operands<0>(G[x], T, w[x], a[x]); operands<0>(G[2 + x], T, w[2 + x], b[x]); }
  asm volatile("s_nop 1" ::: "memory");
#define RSBA_PAIR(XY) if (FAST && m9[XY] == 0x1FFu) nine_mfmas<T, XY, false>(acc, 0u, a, b, G, p, fl); else nine_mfmas<T, XY, true>(acc, m9[XY] * 0x40201u, a, b, G, p, fl);
  RSBA_PAIR(0) RSBA_PAIR(1) RSBA_PAIR(2) RSBA_PAIR(3)
#undef RSBA_PAIR
}
template <bool FAST>
__global__ __launch_bounds__(256, 1) void block_kernel2(const double* __restrict__ Pm, const uint32_t* __restrict__ tab, const unsigned long long* __restrict__ msk, int nq, double* __restrict__ out, long long* clk, int epi, double* __restrict__ part) {
  extern __shared__ uint32_t s_tab[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const size_t wg = (size_t)blockIdx.x * 4 + wave;
  uint32_t* my = s_tab + (size_t)wave * nq * 16;
  for (int i = lane; i < nq * 16; i += 64) my[i] = tab[wg * nq * 16 + i];
  unsigned long long* mym = reinterpret_cast<unsigned long long*>(s_tab + (size_t)4 * nq * 16) + (size_t)wave * nq;
  for (int i = lane; i < nq; i += 64) mym[i] = msk[wg * nq + i];
  __syncthreads();
  const Lane fl = lane_consts(r);
  dbl4 acc[2][2][3][3];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J) acc[x][y][I][J] = dbl4{0, 0, 0, 0};
  Side ring[2][4];
  auto offsets = [&](int q) { const int qq = q < nq ? q : nq - 1; return *reinterpret_cast<const uint4*>(my + ((size_t)qq * 4 + g) * 4); };
  const long long w0 = wall_clock64(), c0 = clock64();
#pragma unroll
  for (int d = 0; d < 2; ++d) { const uint4 o = offsets(d); fetch_side(Pm + o.x, fl, ring[d][0]); fetch_side(Pm + o.y, fl, ring[d][1]); fetch_side(Pm + o.z, fl, ring[d][2]); fetch_side(Pm + o.w, fl, ring[d][3]); }
  uint4 onext = offsets(2);                 // the group that lands during the first group's MFMAs
  unsigned long long mnext = mym[0];
  const int per = epi > 0 ? epi : nq;
  for (int cc = 0; cc < nq; cc += per) {
  for (int base = cc; base < cc + per; base += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int q = base + d;
      unsigned m9[4];
#pragma unroll
      for (int xy = 0; xy < 4; ++xy) m9[xy] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((mnext >> (9 * xy)) & 0x1FFu));
      Side* G = ring[d];
      double w[4][3];
#pragma unroll
      for (int s = 0; s < 4; ++s) weights(G[s], fl, w[s]);
      const double* const p[4] = {Pm + onext.x, Pm + onext.y, Pm + onext.z, Pm + onext.w};
      coordinate<0, FAST>(acc, m9, w, G, p, fl);
      coordinate<1, FAST>(acc, m9, w, G, p, fl);
      coordinate<2, FAST>(acc, m9, w, G, p, fl);
      onext = offsets(q + 3);
      mnext = mym[q + 1 < nq ? q + 1 : nq - 1];
    }
  }
  if (epi > 0) {   // the chunk's epilogue
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double* red = reinterpret_cast<double*>(s_tab + 16384);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        __syncthreads();
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
          for (int J = 0; J < 3; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) { red[wave * (48 * 49) + (16 * I + g + 4 * v) * 49 + 16 * J + r] = acc[x][y][I][J][v]; acc[x][y][I][J][v] = 0.0; }
        __syncthreads();
        double* dst = part + ((size_t)blockIdx.x * 4 + 2 * x + y) * 2304;
        for (int e = tid; e < 2304; e += 256) { const int oo = (e / 48) * 49 + e % 48; dst[e] = (red[oo] + red[48 * 49 + oo]) + (red[2 * 48 * 49 + oo] + red[3 * 48 * 49 + oo]); }
      }
  }
  }
  const long long w1 = wall_clock64(), c1 = clock64();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  double sum = 0;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J) sum += acc[x][y][I][J][0] + acc[x][y][I][J][1] + acc[x][y][I][J][2] + acc[x][y][I][J][3];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = w1 - w0; clk[1] = c1 - c0; }
}

template <int NS, int kDepth, int kWaves, int ASM = 0, int ABL = 0, int KERN = 0>
static void run(const char* what, const double* Pm, size_t ngroups, int wgs, long long entries_total, double* out, long long* clk, bool cached, int epi = 0) {
  // entries_total entries of NS sides, shared out over wgs x 4 waves in groups of four
  const int waves = wgs * 4;
  const int nq = (int)(entries_total / 4 / waves);
  std::vector<uint32_t> tab((size_t)waves * nq * 4 * NS);
  std::vector<unsigned long long> msk((size_t)waves * nq, 0xFFFFFFFFFull);
  // the entries of a chunk (= a workgroup) follow the points: consecutive entries take consecutive groups of NS regions of the buffer (the tiles' groups of
  // neighbouring points lie together); a wave takes every fourth group of four, like the shipped kernel
  srand(7);
  for (int b = 0; b < wgs; ++b) {
    size_t region[NS];
    for (int s = 0; s < NS; ++s) region[s] = cached ? (size_t)s * 4096 : (size_t)((double)rand() / RAND_MAX * (double)(ngroups - (size_t)nq * 16 - 64));
    for (int w = 0; w < 4; ++w)
      for (int q = 0; q < nq; ++q)
        for (int g = 0; g < 4; ++g)
          for (int s = 0; s < NS; ++s) {
            const size_t k = (size_t)(4 * q + w) * 4 + g;   // entry of the chunk
            tab[(((size_t)(b * 4 + w) * nq + q) * 4 + g) * NS + s] = (uint32_t)((region[s] + (cached ? k % 64 : k)) * kGroup);
          }
  }
  uint32_t* d_tab; unsigned long long* d_msk;
  hipMalloc(&d_tab, tab.size() * 4); hipMalloc(&d_msk, msk.size() * 8);
  hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_msk, msk.data(), msk.size() * 8, hipMemcpyHostToDevice);
  const size_t lds = epi > 0 ? 16384 * 4 + 4 * 48 * 49 * 8 : (size_t)4 * nq * 4 * NS * 4 + (size_t)4 * nq * 8;
  if (epi > 0 && (size_t)4 * nq * 4 * NS * 4 + (size_t)4 * nq * 8 > 16384 * 4) { printf("tables too long for the epilogue variant\n"); return; }
  double* part; hipMalloc(&part, (size_t)wgs * 4 * 2304 * 8);
  auto kern = KERN == 1 ? block_kernel2<false> : KERN == 2 ? block_kernel2<true> : loop_kernel<NS, kDepth, kWaves, ASM, ABL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) kern<<<wgs, 256, lds>>>(Pm, d_tab, d_msk, nq, out, clk, epi, part);
  hipDeviceSynchronize();
  float best = 1e9f, sum = 0;
  for (int i = 0; i < 10; ++i) {
    hipEventRecord(e0); kern<<<wgs, 256, lds>>>(Pm, d_tab, d_msk, nq, out, clk, epi, part); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
  }
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mfmas = (double)waves * nq * 27.0 * (NS == 4 ? 4 : 1);
  const hipError_t err = hipGetLastError();
  printf("%-64s %4d wgs, %5d groups of four per wave: best %.3f ms (mean %.3f), %5.1f TFLOP/s, %.1f cycles per MFMA of a wave, loop clock %4.0f MHz%s\n", what, wgs, nq, best, sum / 10,
         mfmas * 2048.0 / best * 1e-9, (double)h[1] / ((double)nq * 27.0 * (NS == 4 ? 4 : 1)), h[1] / (h[0] * 0.01), err == hipSuccess ? "" : hipGetErrorString(err));
  hipFree(d_tab); hipFree(d_msk); hipFree(part);
}

int main() {
  const size_t ngroups = 585000;
  double* Pm; hipMalloc(&Pm, (ngroups + 64) * kGroup * 8);
  std::vector<double> hp((ngroups + 64) * kGroup);
  for (size_t i = 0; i < hp.size(); ++i) hp[i] = 1e-3 * (double)((i * 2654435761u) & 1023) - 0.5;
  hipMemcpy(Pm, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
  double* out; hipMalloc(&out, 8ull * 256 * 2048);
  long long* clk; hipMalloc(&clk, 16);
  const long long pair_entries = 2048000;   // C4: 2.04 M (point, tile pair) entries
  for (int cached = 0; cached < 2; ++cached) {
    printf(cached ? "--- operands out of the caches (64 groups per side) ---\n" : "--- operands from a 374 MB buffer, chunk by chunk ---\n");
    run<2, 2, 2>("pair form, two groups in flight, two waves per SIMD (shipped)", Pm, ngroups, 512, pair_entries, out, clk, cached);
    run<2, 3, 1>("pair form, three groups in flight, one wave per SIMD", Pm, ngroups, 256, pair_entries, out, clk, cached);
    run<4, 1, 1>("block form 2 x 2, one group in flight, one wave per SIMD", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1>("block form 2 x 2, two groups in flight, one wave per SIMD", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 1, 1, 1>("block form 2 x 2, accumulators pinned (32 AGPR + 4 VGPR blocks), one group", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1, 1>("block form 2 x 2, accumulators pinned (32 AGPR + 4 VGPR blocks), two groups", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1, 1, 0, 1>("block 2 x 2 by hand: a load behind every 2nd / 4th MFMA, tables a group ahead", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1, 1, 0, 2>("  + tile pairs with all nine blocks present run without tests", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1, 1, 0, 2>("  + epilogue every 16 groups (256 block entries)", Pm, ngroups, 256, pair_entries / 4, out, clk, cached, 16);
    run<4, 2, 1, 1, 0, 2>("  + epilogue every 24 groups (384 block entries)", Pm, ngroups, 256, pair_entries / 4, out, clk, cached, 24);
    run<4, 2, 1, 1, 1>("  ablation: no loads in the loop", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1, 1, 2>("  ablation: no operand arithmetic", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1, 1, 4>("  ablation: no mask tests", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1, 1, 7>("  ablation: none of the three (MFMAs + the table reads)", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1, 1, 6>("  ablation: loads only (no arithmetic, no tests)", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<2, 2, 2, 0>("pair form (shipped), epilogue every 32 groups of a wave (512 entries)", Pm, ngroups, 512, pair_entries, out, clk, cached, 32);
    run<4, 2, 1, 1>("block 2 x 2 pinned, two groups, epilogue every 16 groups (256 block entries)", Pm, ngroups, 256, pair_entries / 4, out, clk, cached, 16);
    run<4, 2, 1, 1>("block 2 x 2 pinned, two groups, epilogue every 32 groups (512 block entries)", Pm, ngroups, 256, pair_entries / 4, out, clk, cached, 32);
    run<4, 2, 1, 2>("block 2 x 2 pinned, per-coordinate pieces, epilogue every 16 groups", Pm, ngroups, 256, pair_entries / 4, out, clk, cached, 16);
  }
  return 0;
}
