// Do fp64 MFMAs of one wave and fp64 vector instructions of ANOTHER wave on the same SIMD overlap?  (cholesky.hip's ldl_probe shows that
// inside one wave their times add up; the Schur kernel runs two waves per SIMD, one of which is usually in its MFMAs while the other forms
// operands.)  256 workgroups x 512 threads = two waves per SIMD; waves 0-3 of a workgroup (one per SIMD) issue MFMAs, waves 4-7 (the second wave of every SIMD)
// dependent-free v_fma_f64 — mix = 0: everyone MFMAs, 1: everyone FMAs, 2: half and half.  Times per wave-instruction at the clock the kernel ran at.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(double* out, long long* clk, int iters, int mix, int nfma) {
  const bool do_mfma = mix == 0 || (mix == 2 && (threadIdx.x >> 8) == 0);   // waves 0-3 (one per SIMD) against waves 4-7 (the second wave of every SIMD)
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  dbl4 c[9];
  double f[16];
  for (int i = 0; i < 9; ++i) c[i] = dbl4{0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) f[i] = i * 1e-3;
  const long long w0 = wall_clock64(), s0 = clock64();
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 9; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < nfma; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], b, a);
    }
  }
  const long long w1 = wall_clock64(), s1 = clock64();
  double s = 0;
  for (int i = 0; i < 9; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) { clk[2 * (threadIdx.x >> 8)] = w1 - w0; clk[2 * (threadIdx.x >> 8) + 1] = s1 - s0; }
}
int main() {
  double* d; hipMalloc(&d, 8ull * 256 * 8192);
  long long* c; hipMalloc(&c, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;            // 36 000 MFMAs per wave = 2.3 M cycles alone
  for (int nfma : {9000, 18000, 36000}) {   // 16 nfma v_fma_f64 per wave
    printf("-- %d v_fma_f64 per FMA wave against %d MFMAs per MFMA wave\n", 16 * nfma, 9 * iters);
    for (int mix : {0, 1, 2}) {
      for (int w = 0; w < 20; ++w) k<<<256, 512>>>(d, c, iters, mix, nfma);
      hipDeviceSynchronize();
      hipEventRecord(e0); k<<<256, 512>>>(d, c, iters, mix, nfma); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[4]; hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
      const char* what = mix == 0 ? "every wave MFMAs" : mix == 1 ? "every wave FMAs" : "waves 0-3 MFMA, waves 4-7 FMA";
      printf("%-38s kernel %.3f ms; wave 0: %.1f us, %.1f cycles per instruction; wave 4: %.1f us, %.1f cycles per instruction (clock %.0f MHz)\n", what, ms,
             h[0] * 0.01, (double)h[1] / (mix == 1 ? 16.0 * nfma : 9.0 * iters), h[2] * 0.01, (double)h[3] / (mix == 0 ? 9.0 * iters : 16.0 * nfma), h[1] / (h[0] * 0.01));
    }
  }
  return 0;
}
