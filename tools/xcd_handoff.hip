// How long does a tile take from one workgroup to another through the write-once cells of the Cholesky (rsba_amd/csrc/cholesky.hip),
// when both sit on the SAME XCD and the consumer reads at workgroup scope (sc0: served by the XCD's L2) instead of agent scope (sc1: from
// the memory side)?  One producer stores 48 x 48 doubles (agent-scope atomic stores: write-through, what every other XCD needs) after
// a delay; consumers poll the whole tile (36 loads per lane, like the DIAG tasks) and stamp when it is complete.
//   hipcc --offload-arch=gfx950 -O2 tools/xcd_handoff.hip -o tools/xcd_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GL __attribute__((address_space(1)))
__device__ __forceinline__ bool filled(double v) { return __double_as_longlong(v) != -1ll; }
template <int SCOPE> __device__ __forceinline__ double ld(const double* p) { return __hip_atomic_load((const GL double*)p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE> __device__ __forceinline__ void st(double* p, double v) { __hip_atomic_store((GL double*)p, v, __ATOMIC_RELAXED, SCOPE); }
// a load that bypasses the CU's vector cache (sc0) but may be served by the XCD's L2 (no sc1)
__device__ __forceinline__ double ld_sc0(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// block 0 of the launch's XCD 0 produces; mode: 0 = consumers read at agent scope, 1 = workgroup scope, 2 = workgroup scope with every fourth look at agent scope
template <int MODE>
__global__ __launch_bounds__(256) void k(double* tile, long long* out, unsigned* sync, int delay_us) {
  const int tid = threadIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID
  __shared__ unsigned s_role;
  if (tid == 0) s_role = atomicAdd(sync + 1 + xcc, 1u);   // arrival order on this XCD
  __syncthreads();
  const unsigned role = s_role;
  const bool producer = xcc == 0 && role == 0;
  const bool consumer = role == 1 && (xcc == 0 || xcc == 3);   // one consumer next to the producer, one on another XCD
  if (!producer && !consumer) return;
  if (producer) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 100ll * delay_us) __builtin_amdgcn_s_sleep(10);
    const long long t1 = wall_clock64();
    if (MODE >= 5) for (int q = 0; q < 9; ++q) { const double val = 1.0 + tid + 256 * q; asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(tile + tid + 256 * q), "v"(val) : "memory"); }   // a plain store first: lands (dirty) in this XCD's L2
    if (MODE >= 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int q = 0; q < 9; ++q) st<__HIP_MEMORY_SCOPE_AGENT>(tile + tid + 256 * q, 1.0 + tid + 256 * q);
    if (tid == 0) out[0] = t1;
    return;
  }
  double v[9];
  int looks = 0;
  for (;;) {
    bool ok = true;
    const bool agent = MODE == 0 || ((MODE == 2 || MODE == 4 || MODE == 6) && (looks & 3) == 3) || (MODE >= 5 && xcc != 0);
    if (MODE >= 7 && !agent) asm volatile("buffer_inv sc0" ::: "memory");   // drop this CU's vector cache: the loads below go to the XCD's L2
    for (int q = 0; q < 9; ++q) {
      v[q] = (MODE >= 3 && !agent) ? ld_sc0(tile + tid + 256 * q) : agent ? ld<__HIP_MEMORY_SCOPE_AGENT>(tile + tid + 256 * q) : ld<__HIP_MEMORY_SCOPE_WORKGROUP>(tile + tid + 256 * q);
      ok = ok && filled(v[q]);
    }
    ++looks;
    if (__syncthreads_and(ok)) break;
    __builtin_amdgcn_s_sleep(2);
    if (looks > 20000) break;   // (never hang the box)
  }
  if (tid == 0) { out[1 + (xcc != 0)] = wall_clock64(); out[3 + (xcc != 0)] = looks; }
}

int main() {
  double* tile; long long* out; unsigned* sync;
  hipMalloc(&tile, 48 * 48 * 8); hipMalloc(&out, 64); hipMalloc(&sync, 64);
  const char* names[8] = {"agent-scope loads (today)", "workgroup-scope loads", "workgroup scope, every 4th look agent", "sc0 loads (L1 bypass, L2 may serve)", "sc0 loads, every 4th look agent", "plain + agent stores; sc0 loads next to the producer, agent loads elsewhere", "the same, every 4th local look agent", "plain + agent stores; local looks: buffer_inv sc0 + sc0 loads"};
  for (int mode = 0; mode < 8; ++mode) {
    double sum[2] = {0, 0}; long long worst[2] = {0, 0}; long long looks[2] = {0, 0}; int reps = 40, bad = 0;
    for (int r = 0; r < reps; ++r) {
      hipMemset(tile, 0xFF, 48 * 48 * 8); hipMemset(out, 0, 64); hipMemset(sync, 0, 64);
      if (mode == 0) k<0><<<256, 256>>>(tile, out, sync, 30); else if (mode == 1) k<1><<<256, 256>>>(tile, out, sync, 30); else if (mode == 2) k<2><<<256, 256>>>(tile, out, sync, 30); else if (mode == 3) k<3><<<256, 256>>>(tile, out, sync, 30); else if (mode == 4) k<4><<<256, 256>>>(tile, out, sync, 30); else if (mode == 5) k<5><<<256, 256>>>(tile, out, sync, 30); else if (mode == 6) k<6><<<256, 256>>>(tile, out, sync, 30); else k<7><<<256, 256>>>(tile, out, sync, 30);
      hipDeviceSynchronize();
      long long h[5]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
      for (int c = 0; c < 2; ++c) { const long long d = h[1 + c] - h[0]; if (h[1 + c] == 0 || h[3 + c] > 19999) { ++bad; continue; } sum[c] += d; if (d > worst[c]) worst[c] = d; looks[c] += h[3 + c]; }
    }
    printf("%-40s same XCD: mean %.2f us (worst %.2f), other XCD: mean %.2f us (worst %.2f); looks per hand-off %.0f / %.0f; timed out or missing: %d\n", names[mode],
           sum[0] * 0.01 / reps, worst[0] * 0.01, sum[1] * 0.01 / reps, worst[1] * 0.01, (double)looks[0] / reps, (double)looks[1] / reps, bad); fflush(stdout);
  }
  return 0;
}
