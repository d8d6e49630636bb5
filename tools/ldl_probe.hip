// What one pivot of the MFMA-pivot LDL^T step (rsba_amd/csrc/cholesky.hip, ldl16_eliminate / ldl16_follow) costs, piece by piece:
// one wave, 16 pivots, variants with parts of the per-pivot work switched off (fp64 MFMA and fp64 vector time add up), then the
// step split over two waves that talk through LDS.  Prints clock64 ticks and wall_clock64 (100 MHz) time
// per 16-pivot block.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ldl_probe.hip -o tools/ldl_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// FLAGS: 1 = second MFMA (W'), 2 = reciprocal chain, 4 = look-ahead lane reads, 8 = operand selects from the accumulator
template <int FLAGS>
__device__ __forceinline__ dbl4_t block(dbl4_t d, dbl4_t& wout, int lane) {
  const int c = lane & 15, g = lane >> 4;
  dbl4_t w;
#pragma unroll
  for (int v = 0; v < 4; ++v) w[v] = (g + 4 * v == c) ? 1.0 : 0.0;
  double piv = readlane_f64(d[0], 0), rinv = 1.0 / piv;
  double fixed_mul = -0.001 * (c + 1), fixed_row = 0.002 * (g + 1);
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int gg = jj & 3, vv = jj >> 2;
    const bool grp = g == gg;
    double row, mul, sr = 0.25, sd = 4.0;
    if (FLAGS & 8) {
      const double dv = d[vv];
      row = grp ? dv : 0.0;
      mul = (grp && c > jj) ? -(dv * rinv) : 0.0;
      if ((FLAGS & 4) && jj < 15) { sr = readlane_f64(dv, 16 * gg + jj + 1); sd = readlane_f64(d[(jj + 1) >> 2], 16 * ((jj + 1) & 3) + jj + 1); }
    } else { row = fixed_row; mul = fixed_mul * rinv; }
    __builtin_amdgcn_sched_barrier(0);
    d = __builtin_amdgcn_mfma_f64_16x16x4f64(mul, row, d, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    double y0 = 0.0, e = 0.0;
    if ((FLAGS & 2) && jj < 15) { piv = fma(-(sr * rinv), sr, sd); y0 = __builtin_amdgcn_rcp(piv); e = fma(-piv, y0, 1.0); }
    if (FLAGS & 1) {
      const double wrow = (FLAGS & 8) ? (grp ? w[vv] : 0.0) : fixed_row;
      __builtin_amdgcn_sched_barrier(0);
      w = __builtin_amdgcn_mfma_f64_16x16x4f64(mul, wrow, w, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((FLAGS & 2) && jj < 15) rinv = fma(y0, fma(e, e, e), y0);
    __builtin_amdgcn_sched_barrier(0);
  }
  wout = w;
  return d;
}

template <int FLAGS>
__global__ __launch_bounds__(64) void probe(const double* in, double* out, long long* t, int reps) {
  const int lane = threadIdx.x;
  dbl4_t d0;
  for (int v = 0; v < 4; ++v) d0[v] = in[lane * 4 + v];
  dbl4_t acc = {0, 0, 0, 0}, w;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < reps; ++it) {
    dbl4_t d = d0;
    asm volatile("" : "+v"(d));
    d = block<FLAGS>(d, w, lane);
    acc += d + w;
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  for (int v = 0; v < 4; ++v) out[lane * 4 + v] = acc[v];
  if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

template <int FLAGS>
void run(const char* name, const double* din, double* dout, long long* dt) {
  const int reps = 2000;
  long long h[2];
  for (int pass = 0; pass < 2; ++pass) { hipLaunchKernelGGL(probe<FLAGS>, dim3(1), dim3(64), 0, 0, din, dout, dt, reps); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(h, dt, sizeof h, hipMemcpyDeviceToHost);
  std::printf("%-58s %7.1f clock64 ticks / pivot, %6.1f ns / pivot, %6.3f us / 16-pivot block\n", name, h[0] / (16.0 * reps), h[1] * 10.0 / (16.0 * reps), h[1] * 0.01 / reps);
}

// ---- two waves: leader eliminates and posts messages in LDS, follower applies them (cholesky.hip ldl16_eliminate / ldl16_follow)
constexpr int kMsg = 66;
__device__ __forceinline__ bool filled(double v) { return __double_as_longlong(v) != -1ll; }
typedef __attribute__((address_space(3))) double* lds_ptr;
typedef const volatile __attribute__((address_space(3))) double* lds_cvptr;

__device__ __forceinline__ dbl4_t leader(dbl4_t d, double* msg, int lane) {
  const int c = lane & 15, g = lane >> 4;
  lds_ptr lm = (lds_ptr)msg;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int gg = jj & 3, vv = jj >> 2;
    const bool grp = g == gg;
    const double dv = d[vv];
    const double piv = readlane_f64(dv, 16 * gg + jj);
    const double y0 = __builtin_amdgcn_rcp(piv), e = fma(-piv, y0, 1.0), t = dv * y0;
    const double q = fma(t, fma(e, e, e), t);
    const double row = grp ? dv : 0.0;
    const double mul = (grp && c > jj) ? -q : 0.0;
    lm[jj * kMsg + lane] = mul;
    lm[jj * kMsg + 64] = piv;
    if (jj < 15) d = __builtin_amdgcn_mfma_f64_16x16x4f64(mul, row, d, 0, 0, 0);
  }
  return d;
}
template <int SLEEP>
__device__ __forceinline__ dbl4_t follower(const double* msg, int lane, int& polls) {
  const int c = lane & 15, g = lane >> 4;
  lds_cvptr vm = (lds_cvptr)msg;
  dbl4_t w;
#pragma unroll
  for (int v = 0; v < 4; ++v) w[v] = (g + 4 * v == c) ? 1.0 : 0.0;
  double pv[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int gg = jj & 3, vv = jj >> 2;
    const bool grp = g == gg;
    double mul, piv;
    for (;;) {
      mul = vm[jj * kMsg + lane]; piv = vm[jj * kMsg + 64];
      if (__ballot(!(filled(mul) && filled(piv))) == 0ull) break;
      ++polls;
      if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    }
    const double wrow = grp ? w[vv] : 0.0;
    if (jj < 15) w = __builtin_amdgcn_mfma_f64_16x16x4f64(mul, wrow, w, 0, 0, 0);
    pv[vv] = grp ? piv : pv[vv];
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) w[v] *= pv[v];
  return w;
}

// MODE 0: both waves together; 1: leader alone; 2: follower alone on messages already posted
template <int MODE, int SLEEP>
__global__ __launch_bounds__(256) void pair_probe(const double* in, double* out, long long* t, int reps) {
  __shared__ double msg[16 * kMsg];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  dbl4_t d0;
  for (int v = 0; v < 4; ++v) d0[v] = in[lane * 4 + v];
  dbl4_t acc = {0, 0, 0, 0};
  long long lead = 0, foll = 0, total = 0; int polls = 0;
  for (int it = 0; it < reps; ++it) {
    if (MODE != 2 || it == 0) for (int e = tid; e < 16 * kMsg; e += 256) msg[e] = __longlong_as_double(-1ll);
    __syncthreads();
    if (MODE == 2 && it == 0) { if (wave == 0) acc += leader(d0, msg, lane); __syncthreads(); }
    const long long c0 = clock64();
    if (wave == 0 && MODE != 2) { dbl4_t d = d0; asm volatile("" : "+v"(d)); acc += leader(d, msg, lane); lead += clock64() - c0; }
    if (wave == 1 && MODE != 1) { acc += follower<SLEEP>(msg, lane, polls); foll += clock64() - c0; }
    __syncthreads();
    total += clock64() - c0;
  }
  for (int v = 0; v < 4; ++v) out[tid * 4 + v] = acc[v];
  if (tid == 0) { t[0] = lead; t[2] = total; }
  if (tid == 64) { t[1] = foll; t[3] = polls; }
}
template <int MODE, int SLEEP>
void run_pair(const char* name, const double* din, double* dout, long long* dt) {
  const int reps = 2000;
  long long h[4];
  for (int pass = 0; pass < 2; ++pass) { hipLaunchKernelGGL((pair_probe<MODE, SLEEP>), dim3(1), dim3(256), 0, 0, din, dout, dt, reps); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(h, dt, sizeof h, hipMemcpyDeviceToHost);
  std::printf("%-44s leader %7.1f  follower %7.1f  both + barrier %7.1f ticks per 16-pivot block; failed polls per block %.1f\n", name, h[0] / (double)reps, h[1] / (double)reps,
              h[2] / (double)reps, h[3] / (double)reps);
}

int main() {
  double h[256];
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) { const int r = (l >> 4) + 4 * v, c = l & 15; h[l * 4 + v] = (r == c) ? 20.0 + r : 1.0 / (1 + r + c); }
  double *din, *dout; long long* dt;
  (void)hipMalloc(&din, sizeof h); (void)hipMalloc(&dout, 256 * 4 * 8); (void)hipMalloc(&dt, 32);
  (void)hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  run<0>("one MFMA per pivot, operands fixed (bare dependent MFMAs)", din, dout, dt);
  run<1>("two MFMAs per pivot, operands fixed", din, dout, dt);
  run<8>("one MFMA, operands selected from the accumulator", din, dout, dt);
  run<9>("two MFMAs, operands selected from the accumulator", din, dout, dt);
  run<8 | 2>("one MFMA, selects, reciprocal chain", din, dout, dt);
  run<8 | 4 | 2>("one MFMA, selects, lane reads, reciprocal chain", din, dout, dt);
  run<15>("everything (the step as built)", din, dout, dt);
  run<2>("one MFMA, fixed operands scaled by the reciprocal chain", din, dout, dt);
  run_pair<1, 0>("leader alone", din, dout, dt);
  run_pair<2, 0>("follower alone (messages already there)", din, dout, dt);
  run_pair<0, 0>("leader + follower, tight poll", din, dout, dt);
  run_pair<0, 1>("leader + follower, s_sleep 1 between polls", din, dout, dt);
  run_pair<0, 4>("leader + follower, s_sleep 4 between polls", din, dout, dt);
  return 0;
}
