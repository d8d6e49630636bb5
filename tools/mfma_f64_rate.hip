// Sustained rate of v_mfma_f64_16x16x4_f64 (is the fp64 matrix peak reachable, or what holds it back?).  Per launch shape: time, TFLOP/s,
// the shader clock during the kernel (s_memtime against the 100 MHz wall clock) and the cycles one SIMD spends per MFMA at THAT clock.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* clk, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  dbl4 c[9];
  for (int i = 0; i < 9; ++i) c[i] = dbl4{0, 0, 0, 0};
  const long long w0 = wall_clock64(), s0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  const long long w1 = wall_clock64(), s1 = clock64();
  double s = 0;
  for (int i = 0; i < 9; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = w1 - w0; clk[1] = s1 - s0; }
}
int main() {
  double* d; hipMalloc(&d, 8ull * 256 * 8192);
  long long* c; hipMalloc(&c, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Shape { int wgs, threads; const char* what; };
  const Shape shapes[] = {{256, 64, "one wave per CU (one SIMD of four)"}, {256, 128, "two SIMDs per CU"}, {256, 256, "one wave per SIMD"},
                          {512, 256, "two waves per SIMD"}, {1024, 256, "four waves per SIMD"}, {64, 256, "one wave per SIMD on a quarter of the CUs"}};
  for (const Shape& sh : shapes) {
    const int iters = 4000;
    for (int w = 0; w < 40; ++w) k<<<sh.wgs, sh.threads>>>(d, c, iters);   // let the clocks settle
    hipDeviceSynchronize();
    hipEventRecord(e0); k<<<sh.wgs, sh.threads>>>(d, c, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double waves = (double)sh.wgs * sh.threads / 64, flops = waves * iters * 9 * 2048.0;
    const double mhz = h[1] / (h[0] * 0.01);           // shader ticks per microsecond
    printf("%-44s %5d x %3d: %.3f ms, %5.1f TFLOP/s, shader clock %4.0f MHz, %.0f shader cycles per MFMA of a wave\n", sh.what, sh.wgs, sh.threads, ms, flops / ms * 1e-9, mhz,
           (double)h[1] / (iters * 9.0));
  }
  return 0;
}
