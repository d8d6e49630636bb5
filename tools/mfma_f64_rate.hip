// Sustained rate of v_mfma_f64_16x16x4_f64 with every SIMD of the chip busy (is the fp64 matrix peak reachable,
// or does the clock drop under it?).  hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  dbl4 c[9];
  for (int i = 0; i < 9; ++i) c[i] = dbl4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 9; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, 8ull * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {256, 512, 1024}) {
    const int iters = 4000;
    for (int w = 0; w < 60; ++w) k<<<wgs, 256>>>(d, iters);   // let the clocks settle
    hipDeviceSynchronize();
    hipEventRecord(e0); k<<<wgs, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * 9 * 2048.0;
    const double waves_per_simd = wgs * 4 / 1024.0;
    printf("%4d workgroups: %.3f ms, %.1f TFLOP/s, %.0f cycles per MFMA per SIMD at 2.4 GHz\n", wgs, ms, flops / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / (iters * 9.0 * (waves_per_simd < 1 ? 1 : waves_per_simd)));
  }
  return 0;
}
