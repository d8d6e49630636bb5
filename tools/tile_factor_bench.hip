// NOTE (end of round 3): this file times the code as the compiler lays it out for THESE kernels; inside the solver's persistent kernel
// the same source is laid out, allocated and fetched differently (tools/chol_trace.py is the measurement that counts): the rolled
// three-block loop of factor_invert_tile made the tile 0.9 us faster there (8.3 -> 7.4 us) and 1.2 us slower here.
// Micro-benchmark + check of the serial core of the tile Cholesky (rsba_amd/csrc/cholesky.hip): W = chol(D)^-1 of one
// 48 x 48 tile by ONE 256-thread workgroup — the step that sits 42 times on the critical path of a 1k-camera solve.
// Variants: the lane-per-row blocked potrf + blocked triangular inverse (round 1) and the MFMA-pivot LDL^T form.
// Checks |W D W^T - I| and |W - W_ref| against a host factorisation, prints microseconds per tile (wall_clock64, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/tile_factor_bench.hip -o tools/tile_factor_bench
#include "../rsba_amd/csrc/cholesky.hip"

#include <cmath>
#include <cstdio>
#include <vector>

namespace rsba {
hipError_t allow_dynamic_lds_impl(const void* kernel, size_t bytes) { return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); }   // (capi.hip's, for the launchers pulled in above)
namespace {

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void tile_kernel(const double* __restrict__ Din, double* __restrict__ Wout, long long* ticks, int reps, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  double* D = smem; double* Wl = smem + kBuf; double* Tm = smem + 2 * kBuf; double* Lp = smem + 3 * kBuf;
  double* dinv = smem + kVecOff + 5 * T;
  long long total = 0;
  for (int it = 0; it < reps; ++it) {
    for (int e = tid; e < T * T; e += 256) { const int r = e / T, c = e % T; D[r * TP + c] = (c <= r) ? Din[e] : 0.0; }
    arm_pivot_messages(smem, tid);
    __syncthreads();
    const long long t0 = wall_clock64();
    if (VARIANT == 0) {
      potrf_blocked(D, tid);
      if (tid < T) dinv[tid] = 1.0 / D[tid * TP + tid];
      lds_barrier();
      invert_lower_blocked(D, dinv, Wl, Tm, tid);
    } else {
      if (VARIANT == 2) factor_invert_tile<true>(D, Wl, Tm, Lp, tid, stamps);   // 2: with phase stamps (every repetition: warm code)
      else if (VARIANT == 3) factor_invert_tile<false, true>(D, Wl, Tm, Lp, tid);   // 3: round 2's second wave (rank-1 MFMAs on the identity)
      else if (VARIANT == 4) factor_invert_tile<false, false, true>(D, Wl, Tm, Lp, tid, nullptr, Wout + T * T);   // 4: as the persistent driver runs it — W published to write-once cells by row blocks
      else factor_invert_tile(D, Wl, Tm, Lp, tid);
    }
    const long long t1 = wall_clock64();
    total += t1 - t0;
    __syncthreads();
  }
  for (int e = tid; e < T * T; e += 256) { const int r = e / T, c = e % T; Wout[e] = (c <= r) ? Wl[r * TP + c] : 0.0; }
  if (tid == 0) ticks[0] = total;
}


// the two waves of one 16 x 16 diagonal block step on their own: MODE 0 both, 1 the eliminating wave alone, 2 the following wave alone
// on messages that are already there.  FW = which wave of the workgroup follows.
// What slows the following wave down when the eliminating wave runs beside it (both: 366 cycles per pivot; alone on ready messages: 204)?
// A STAND-IN for the eliminating wave posts the same messages (taken from a first, real elimination) at the eliminating wave's pace —
// with nothing but s_sleep between them (PACER 1), or with the real wave's sixteen dependent rank-1 MFMAs on a dummy block in between
// (PACER 2): if the follower keeps pace with 1 and not with 2, it is the matrix pipe's traffic that costs it, not the waiting.
template <int PACER>
__global__ __launch_bounds__(256) void paced_kernel(const double* __restrict__ Din, double* __restrict__ out, long long* t, int reps, int sleep_units) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double* D = smem; double* msg = smem;
  double* saved = smem + 4 * kBuf + 64;   // [16][64] the messages of a real elimination (behind the task's buffers)
  for (int e = tid; e < T * T; e += 256) { const int r = e / T, c = e % T; D[r * TP + c] = (c <= r) ? Din[e] : 0.0; }
  __syncthreads();
  const dbl4_t d0 = load_sym16(D, 0, lane);
  arm_pivot_messages(smem, tid);
  __syncthreads();
  if (wave == 0) ldl16_eliminate(d0, msg, lane);
  __syncthreads();
  if (wave == 0) for (int jj = 0; jj < 16; ++jj) saved[jj * 64 + lane] = msg[msg_off(jj) + lane];
  __syncthreads();
  dbl4_t acc = {0, 0, 0, 0}, dummy = d0;
  long long lead = 0, foll = 0, total = 0;
  for (int it = 0; it < reps; ++it) {
    arm_pivot_messages(smem, tid);
    __syncthreads();
    const long long c0 = clock64();
    if (wave == 0) {
      for (int jj = 0; jj < 16; ++jj) {
        if (PACER == 2) dummy = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-30 * dummy[0], 1e-30, dummy, 0, 0, 0);   // (a dependent chain, like the real wave's)
        for (int u = 0; u < sleep_units; ++u) __builtin_amdgcn_s_sleep(1);
        msg[msg_off(jj) + lane] = saved[jj * 64 + lane];
      }
      lead += clock64() - c0;
    }
    if (wave == 1) { ldl16_follow_rows(msg, smem + kBuf, 0, lane); acc[0] += smem[kBuf + lane]; foll += clock64() - c0; }
    __syncthreads();
    total += clock64() - c0;
  }
  for (int v = 0; v < 4; ++v) out[tid * 4 + v] = acc[v] + dummy[v];
  if (tid == 0) { t[0] = lead; t[2] = total; }
  if (tid == 64) t[1] = foll;
}

template <int MODE, int FW, bool ROWS = false>
__global__ __launch_bounds__(256) void pair_kernel(const double* __restrict__ Din, double* __restrict__ out, long long* t, int reps, long long* stamps = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double* D = smem; double* msg = smem;
  for (int e = tid; e < T * T; e += 256) { const int r = e / T, c = e % T; D[r * TP + c] = (c <= r) ? Din[e] : 0.0; }
  __syncthreads();
  const dbl4_t d0 = load_sym16(D, 0, lane);
  dbl4_t acc = {0, 0, 0, 0};
  long long lead = 0, foll = 0, total = 0;
  for (int it = 0; it < reps; ++it) {
    if (MODE != 2 || it == 0) arm_pivot_messages(smem, tid);
    __syncthreads();
    if (MODE == 2 && it == 0) { if (wave == 0) ldl16_eliminate(d0, msg, lane); __syncthreads(); }
    const long long c0 = clock64();
    const bool last = stamps && it == reps - 1;
    if (last && tid == 0) stamps[0] = c0;
    if (wave == 0 && MODE != 2) { dbl4_t d = d0; asm volatile("" : "+v"(d)); const bool ok = ldl16_eliminate(d, msg, lane, last ? stamps + 1 : nullptr); acc[0] += ok; lead += clock64() - c0; }
    if (wave == FW && MODE != 1) {
      if (ROWS) { ldl16_follow_rows(msg, smem + kBuf, 0, lane, last ? stamps + 17 : nullptr); acc[0] += smem[kBuf + lane]; } else acc += ldl16_follow(msg, lane);
      foll += clock64() - c0;
    }
    __syncthreads();
    total += clock64() - c0;
  }
  for (int v = 0; v < 4; ++v) out[tid * 4 + v] = acc[v];
  if (tid == 0) { t[0] = lead; t[2] = total; }
  if (tid == 64 * FW) t[1] = foll;
}

}  // namespace
}  // namespace rsba

int main() {
  using namespace rsba;
  constexpr int n = T;
  // SPD tile with the conditioning of a damped reduced camera block: A = B B^T + 0.1 I
  std::vector<double> B(n * n), A(n * n, 0.0), L(n * n, 0.0), Wref(n * n, 0.0);
  unsigned long long st = 12345;
  auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return ((st >> 11) * (1.0 / 9007199254740992.0)) - 0.5; };
  for (double& b : B) b = rnd();
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += B[i * n + k] * B[j * n + k]; A[i * n + j] = s + (i == j ? 0.1 : 0.0); }
  for (int j = 0; j < n; ++j) {   // host Cholesky and inverse of the factor
    double d = A[j * n + j]; for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = std::sqrt(d);
    for (int i = j + 1; i < n; ++i) { double s = A[i * n + j]; for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k]; L[i * n + j] = s / L[j * n + j]; }
  }
  for (int c = 0; c < n; ++c) for (int i = c; i < n; ++i) { double s = (i == c) ? 1.0 : 0.0; for (int k = c; k < i; ++k) s -= L[i * n + k] * Wref[k * n + c]; Wref[i * n + c] = s / L[i * n + i]; }
  double *dA, *dW; long long *dT, *dS;
  hipMalloc(&dS, 64 * 8); hipMemset(dS, 0, 64 * 8);
  hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dW, 2 * sizeof(double) * n * n); hipMalloc(&dT, 8);
  hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
  const size_t lds = kCholLds * sizeof(double);
  hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int reps = 2000;
  {   // what the instruction cache costs: the factorisation is ~20 KB of straight-line code that a DIAG task of the solver runs ONCE
    auto once = [&](int r, const char* what) {
      long long ticks = 0;
      hipLaunchKernelGGL(tile_kernel<1>, dim3(1), dim3(256), lds, 0, dA, dW, dT, r, dS); (void)hipDeviceSynchronize();
      (void)hipMemcpy(&ticks, dT, 8, hipMemcpyDeviceToHost);
      std::printf("MFMA-pivot LDL^T, %s: %.3f us for %d tile(s)\n", what, ticks * 0.01, r);
    };
    once(1, "first launch of the process, one tile (cold code)");
    once(1, "second launch, one tile");
    once(2, "third launch, two tiles");
    once(3, "fourth launch, three tiles");
  }
  for (int variant = 0; variant < 5; ++variant) {
    for (int pass = 0; pass < 2; ++pass) {   // first pass warms up
      if (variant == 0) hipLaunchKernelGGL(tile_kernel<0>, dim3(1), dim3(256), lds, 0, dA, dW, dT, reps, dS);
      else if (variant == 1) hipLaunchKernelGGL(tile_kernel<1>, dim3(1), dim3(256), lds, 0, dA, dW, dT, reps, dS);
      else if (variant == 2) hipLaunchKernelGGL(tile_kernel<2>, dim3(1), dim3(256), lds, 0, dA, dW, dT, reps, dS);
      else if (variant == 3) hipLaunchKernelGGL(tile_kernel<3>, dim3(1), dim3(256), lds, 0, dA, dW, dT, reps, dS);
      else hipLaunchKernelGGL(tile_kernel<4>, dim3(1), dim3(256), lds, 0, dA, dW, dT, reps, dS);
      hipDeviceSynchronize();
    }
    std::vector<double> W(n * n); long long ticks = 0;
    hipMemcpy(W.data(), dW, sizeof(double) * n * n, hipMemcpyDeviceToHost); hipMemcpy(&ticks, dT, 8, hipMemcpyDeviceToHost);
    double err_id = 0, err_ref = 0, wmax = 0;
    std::vector<double> WA(n * n, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += W[i * n + k] * A[k * n + j]; WA[i * n + j] = s; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
      double s = 0; for (int k = 0; k < n; ++k) s += WA[i * n + k] * W[j * n + k];
      err_id = std::fmax(err_id, std::fabs(s - (i == j ? 1.0 : 0.0)));
      err_ref = std::fmax(err_ref, std::fabs(W[i * n + j] - Wref[i * n + j])); wmax = std::fmax(wmax, std::fabs(Wref[i * n + j]));
    }
    std::printf("%s: %.3f us per tile (%d reps), |W A W^T - I| = %.2e, |W - W_host| = %.2e (|W| = %.2e), %s\n",
                variant == 0 ? "lane-per-row potrf + blocked inverse" : variant == 1 ? "MFMA-pivot LDL^T, W by rows (round 3)" : variant == 2 ? "  the same with phase stamps        " : variant == 3 ? "MFMA-pivot LDL^T, W by MFMAs (round 2)" : "  round 3 + W published by row blocks ", ticks * 0.01 / reps, reps, err_id, err_ref, wmax,
                hipGetErrorString(hipGetLastError()));
  }
  {
    double* dO; long long* dP; (void)hipMalloc(&dO, 256 * 4 * 8); (void)hipMalloc(&dP, 32);
    auto run_pair = [&](auto kern, const char* name) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      long long h[3];
      for (int pass = 0; pass < 2; ++pass) { hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, 0, dA, dO, dP, reps, (long long*)nullptr); (void)hipDeviceSynchronize(); }
      (void)hipMemcpy(h, dP, sizeof h, hipMemcpyDeviceToHost);
      std::printf("%-48s eliminating wave %7.1f  following wave %7.1f  both + barrier %7.1f ticks per 16-pivot block\n", name, h[0] / (double)reps, h[1] / (double)reps, h[2] / (double)reps);
    };
    run_pair(pair_kernel<1, 1>, "eliminating wave alone");
    run_pair(pair_kernel<2, 1>, "following wave alone (messages already there)");
    run_pair(pair_kernel<0, 1>, "both, wave 1 follows");
    run_pair(pair_kernel<0, 2>, "both, wave 2 follows");
    run_pair(pair_kernel<0, 3>, "both, wave 3 follows");
    run_pair(pair_kernel<2, 1, true>, "W by rows: following wave alone (messages there)");
    run_pair(pair_kernel<0, 1, true>, "W by rows: both, wave 1 follows");
    for (int sl : {0, 1, 2, 3, 4}) {
      auto run_paced = [&](auto kern, const char* name) {
        const size_t lds2 = lds + (64 + 16 * 64) * sizeof(double);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        long long h[3];
        for (int pass = 0; pass < 2; ++pass) { hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds2, 0, dA, dO, dP, reps, sl); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(h, dP, sizeof h, hipMemcpyDeviceToHost);
        std::printf("%-48s posting wave %7.1f  following wave %7.1f  both + barrier %7.1f ticks per 16-pivot block  (%d x s_sleep 1 per pivot)\n", name, h[0] / (double)reps, h[1] / (double)reps, h[2] / (double)reps, sl);
      };
      run_paced(paced_kernel<1>, "stand-in posts ready messages, sleeps between");
      run_paced(paced_kernel<2>, "stand-in posts ready messages + dependent MFMAs");
    }
    {   // when each pivot message is posted (eliminating wave), seen, and its row stored (following wave): the last repetition of the pair above
      long long* dQ; (void)hipMalloc(&dQ, 128 * 8); (void)hipMemset(dQ, 0, 128 * 8);
      hipLaunchKernelGGL((pair_kernel<0, 1, true>), dim3(1), dim3(256), lds, 0, dA, dO, dP, 50, dQ); (void)hipDeviceSynchronize();
      long long q[128]; (void)hipMemcpy(q, dQ, sizeof q, hipMemcpyDeviceToHost);
      std::printf("pivot k: message posted | seen | next look + multipliers requested | updates of pivot k done | row k stored (shader cycles since the block started)\n");
      for (int k = 0; k < 16; ++k) std::printf("  %2d: %6lld | %6lld | %6lld | %6lld | %6lld\n", k, q[1 + k] - q[0], q[17 + 4 * k] - q[0], q[17 + 4 * k + 1] - q[0], q[17 + 4 * k + 2] - q[0], q[17 + 4 * k + 3] - q[0]);
    }
  }
  long long hs[64]; hipMemcpy(hs, dS, sizeof hs, hipMemcpyDeviceToHost);
  std::printf("phases of the last MFMA-pivot tile (clock64 ticks since entry; pairs = before / after each barrier):");
  for (int k = 1; k < 13; ++k) std::printf(" %lld", hs[k] - hs[0]);
  std::printf("\nfollower wave: pivot message k seen at tick (since entry):");
  for (int k = 0; k < 16; ++k) std::printf(" %lld", hs[16 + k] - hs[0]);
  std::printf("\n");
  return 0;
}
