// Layout check of v_mfma_f64_16x16x4_f64: D = A(16x4) * B(4x16) with asymmetric inputs, prints which
// (row, col) each lane's 4 results hold.  hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_check.hip -o /tmp/mfma_check
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* out, long long* cyc) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];      // A[i = l&15][k = l>>4]
  const double b = B[(l >> 4) * 16 + (l & 15)];     // B[k = l>>4][j = l&15]
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
  // issue-rate probe: 64 dependent-free MFMAs on 4 accumulators
  double4_t c0 = c, c1 = c, c2 = c, c3 = c;
  asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  const long long t0 = clock64();
  asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  for (int i = 0; i < 16; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  const long long t1 = clock64();
  if (l == 0) { cyc[0] = t1 - t0; }
  out[256 + l] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  double hA[64], hB[64], hD[16][16] = {}, ho[256 + 64];
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = 1 + i + 0.1 * kk;
  for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = 100 * (kk + 1) + 7 * j * j + j;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int kk = 0; kk < 4; ++kk) hD[i][j] += hA[i * 4 + kk] * hB[kk * 16 + j];
  double *dA, *dB, *dO; long long* dC; long long hc = 0;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dO, sizeof ho); hipMalloc(&dC, 8);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dO, dC);
  hipMemcpy(ho, dO, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(&hc, dC, 8, hipMemcpyDeviceToHost);
  int okA = 1, okB = 1;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const double got = ho[l * 4 + v];
    if (got != hD[(l >> 4) + 4 * v][l & 15]) okA = 0;       // row = (lane>>4) + 4*reg
    if (got != hD[4 * (l >> 4) + v][l & 15]) okB = 0;       // row = 4*(lane>>4) + reg
  }
  for (int l = 0; l < 64; l += 7) for (int v = 0; v < 4; ++v) {
    int fi = -1, fj = -1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (hD[i][j] == ho[l * 4 + v]) { fi = i; fj = j; }
    printf("lane %2d reg %d -> (%d,%d) got %.3f\n", l, v, fi, fj, ho[l * 4 + v]);
  }
  printf("layout row=(lane>>4)+4*reg: %s ; row=4*(lane>>4)+reg: %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
  printf("64 independent-ish MFMAs: %lld clock64 ticks -> %.1f ticks per MFMA\n", hc, hc / 64.0);
  return 0;
}
