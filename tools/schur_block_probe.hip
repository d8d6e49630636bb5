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
__device__ __forceinline__ void operands(const Side& S, int t, const double w[3], double out[3]) { out[0] = S.qm[t] * w[0]; out[1] = S.qm[t] * w[1]; out[2] = S.ql[t] * w[2]; }

// NS sides per entry: 2 (pair form: acc[1][1]) or 4 (block form: acc[2][2]); table [nq][4 lane groups][NS] offsets + [nq] masks per wave, staged in LDS
// ASM: every MFMA as inline assembly with the accumulator's register file fixed — 32 of the block form's 36 accumulators in AGPRs (all 256 of them), the last four in
// VGPRs: left to itself the compiler puts all 36 into AGPR form and moves the four that do not fit in and out around every group (v_accvgpr_read / write)
template <int NB, bool ASM>
__device__ __forceinline__ void mfma(dbl4& c, double a, double b) {
  if constexpr (!ASM) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  else if constexpr (NB < 32) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int NX, bool ASM, class Acc>
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
template <int NS, int kDepth, int kWaves, bool ASM = false>
__global__ __launch_bounds__(256, kWaves) void loop_kernel(const double* __restrict__ Pm, const uint32_t* __restrict__ tab, const unsigned long long* __restrict__ msk, int nq, double* __restrict__ out, long long* clk) {
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
  for (int base = 0; base < nq; base += kDepth) {
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
        for (int s = 0; s < NS; ++s) weights(G.s[s], fl, w[s]);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          double a[NX][3], b[NX][3];
#pragma unroll
          for (int x = 0; x < NX; ++x) { operands(G.s[x], t, w[x], a[x]); operands(G.s[NX + x], t, w[NX + x], b[x]); }
          if constexpr (ASM) asm volatile("s_nop 1" ::: "memory");   // (VALU write -> MFMA read of the operands: the compiler cannot see into the asm statements)
#pragma unroll
          for (int x = 0; x < NX; ++x)
#pragma unroll
            for (int y = 0; y < NX; ++y)
#pragma unroll
              for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int J = 0; J < 3; ++J)
                  if (__builtin_expect(((m27[x][y] >> (9 * t + 3 * I + J)) & 1u) != 0u, 1)) mfma_at<NX, ASM>(acc, x, y, I, J, a[x][I], b[y][J]);
        }
      }
      fetch(base + d + kDepth, ring[d]);
    }
  }
  const long long w1 = wall_clock64(), c1 = clock64();
  if constexpr (ASM) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (MFMA write -> VALU read of the accumulators)
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

template <int NS, int kDepth, int kWaves, bool ASM = false>
static void run(const char* what, const double* Pm, size_t ngroups, int wgs, long long entries_total, double* out, long long* clk, bool cached) {
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
  const size_t lds = (size_t)4 * nq * 4 * NS * 4 + (size_t)4 * nq * 8;
  auto kern = loop_kernel<NS, kDepth, kWaves, ASM>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) kern<<<wgs, 256, lds>>>(Pm, d_tab, d_msk, nq, out, clk);
  hipDeviceSynchronize();
  float best = 1e9f, sum = 0;
  for (int i = 0; i < 10; ++i) {
    hipEventRecord(e0); kern<<<wgs, 256, lds>>>(Pm, d_tab, d_msk, nq, out, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
  }
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mfmas = (double)waves * nq * 27.0 * (NS == 4 ? 4 : 1);
  const hipError_t err = hipGetLastError();
  printf("%-64s %4d wgs, %5d groups of four per wave: best %.3f ms (mean %.3f), %5.1f TFLOP/s, %.1f cycles per MFMA of a wave, loop clock %4.0f MHz%s\n", what, wgs, nq, best, sum / 10,
         mfmas * 2048.0 / best * 1e-9, (double)h[1] / ((double)nq * 27.0 * (NS == 4 ? 4 : 1)), h[1] / (h[0] * 0.01), err == hipSuccess ? "" : hipGetErrorString(err));
  hipFree(d_tab); hipFree(d_msk);
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
    run<4, 1, 1, true>("block form 2 x 2, accumulators pinned (32 AGPR + 4 VGPR blocks), one group", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
    run<4, 2, 1, true>("block form 2 x 2, accumulators pinned (32 AGPR + 4 VGPR blocks), two groups", Pm, ngroups, 256, pair_entries / 4, out, clk, cached);
  }
  return 0;
}
