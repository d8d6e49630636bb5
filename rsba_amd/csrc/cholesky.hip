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
// One launch per tile column k (right-looking with the trailing update of column k-1 deferred by one
// step, so it overlaps the next panel instead of sitting on the critical path):
//   panel workgroups (1 + sub-diagonal tiles of column k): apply the pending step-(k-1) update to the
//     diagonal tile and to their own tile, factor S_kk = L L^T in LDS (redundantly — 48^3/3 flops — so
//     no inter-workgroup hand-off is needed); workgroup 0 stores L_kk and forward-substitutes the
//     right-hand-side tile, workgroup t>0 solves its tile  L_ik = S_ik L_kk^-T;
//   trailing workgroups: the remaining step-(k-1) updates  S_ij -= L_i,k-1 L_j,k-1^T  (j > k), the (i,i)
//     ones also carry  rhs_i -= L_i,k-1 z_k-1  (the forward solve rides along with the factorisation).
// The backward solve L^T y = z is one persistent workgroup walking the tile columns in reverse.
#include "solver_state.hpp"

namespace rsba {

namespace {

constexpr int T = kTile;
constexpr int TP = T + 1;   // LDS row pitch (doubles): odd pitch keeps column walks conflict-free

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1/sqrt(x) to full fp64 accuracy: hardware estimate + two Newton steps (avoids the long fp64
// sqrt-then-divide dependency chain on the factorisation's critical path)
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

__device__ __forceinline__ double* tile_ptr(const SolverDev& sv, int slot) { return sv.S + (size_t)slot * (T * T); }

__device__ __forceinline__ void load_tile(double* dst, const double* src, int tid, bool lower_only) {
  for (int e = tid; e < T * T; e += 256) {
    const int r = e / T, c = e % T;
    dst[r * TP + c] = (!lower_only || c <= r) ? src[e] : 0.0;
  }
}
__device__ __forceinline__ void store_tile(double* dst, const double* src, int tid) {
  for (int e = tid; e < T * T; e += 256) dst[e] = src[(e / T) * TP + e % T];
}

// C -= A B^T on T x T tiles in LDS; 256 threads as 16 x 16, each a 3 x 3 micro-tile over K = 48
__device__ __forceinline__ void tile_gemm_sub(double* C, const double* A, const double* B, int tid) {
  const int ty = tid >> 4, tx = tid & 15;
  double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll 4
  for (int m = 0; m < T; ++m) {
    double a[3], b[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) { a[u] = A[(ty * 3 + u) * TP + m]; b[u] = B[(tx * 3 + u) * TP + m]; }
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) acc[u][v] += a[u] * b[v];
  }
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) C[(ty * 3 + u) * TP + tx * 3 + v] -= acc[u][v];
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

constexpr int NB = 12;   // inner block: one rolling-shutter camera block

// Factor the T x T tile held in LDS (lower triangle, pitch TP) in place: blocked right-looking Cholesky.
// Each 12-column panel is factored by the first wave entirely in registers (lane = row, pivot rows
// broadcast with v_readlane, 1/sqrt instead of sqrt + divide: no LDS round trips or barriers on the
// serial chain); the rank-12 trailing update is spread over all 256 threads.  All threads must call.
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
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < jj; ++m) s += a[m] * readlane_f64(a[m], jj);
        const double t = a[jj] - s;
        const double djj = readlane_f64(t, jj);
        ok = ok && (djj > 0.0) && isfinite(djj);
        const double rinv = rsqrt_nr(djj);
        a[jj] = (tid == jj) ? djj * rinv : t * rinv;
      }
      if (live) {
#pragma unroll
        for (int m = 0; m < NB; ++m) if (c0 + m <= row) A[row * TP + c0 + m] = a[m];
      }
    }
    __syncthreads();
    const int n1 = T - (c0 + NB);
    if (n1 > 0) {
      for (int e = tid; e < n1 * n1; e += 256) {
        const int r = c0 + NB + e / n1, c = c0 + NB + e % n1;
        if (c <= r) {
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int m = 0; m < NB; m += 2) { s0 += A[r * TP + c0 + m] * A[c * TP + c0 + m]; s1 += A[r * TP + c0 + m + 1] * A[c * TP + c0 + m + 1]; }
          A[r * TP + c] -= s0 + s1;
        }
      }
      __syncthreads();
    }
  }
  return ok;
}

// X <- X L^-T for the T x T tiles X and L (lower, factored) in LDS, dinv = 1 / diag(L): blocked forward
// substitution along the rows; 12-column solves by one thread per row, rank-12 updates by all threads.
__device__ __forceinline__ void trsm_blocked(double* X, const double* L, const double* dinv, int tid) {
#pragma unroll
  for (int bc = 0; bc < T / NB; ++bc) {
    const int c0 = NB * bc;
    if (tid < T) {
      double x[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) x[m] = X[tid * TP + c0 + m];
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < jj; ++m) s += x[m] * L[(c0 + jj) * TP + c0 + m];
        x[jj] = (x[jj] - s) * dinv[c0 + jj];
      }
#pragma unroll
      for (int m = 0; m < NB; ++m) X[tid * TP + c0 + m] = x[m];
    }
    __syncthreads();
    const int n1 = T - (c0 + NB);
    if (n1 > 0) {
      for (int e = tid; e < T * n1; e += 256) {
        const int r = e / n1, c = c0 + NB + e % n1;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int m = 0; m < NB; m += 2) { s0 += X[r * TP + c0 + m] * L[c * TP + c0 + m]; s1 += X[r * TP + c0 + m + 1] * L[c * TP + c0 + m + 1]; }
        X[r * TP + c] -= s0 + s1;
      }
      __syncthreads();
    }
  }
}

struct StepArgs {
  int k;
  int npanel;                 // 1 + number of sub-diagonal tiles of column k
  const int32_t* panel_slot;  // [npanel] slot of tile (i,k); entry 0 is the diagonal tile (k,k)
  const int32_t* prev_slot;   // [npanel] slot of tile (i,k-1) if it exists (pending update), else -1; entry 0 is (k,k-1)
  const int32_t* trail;       // [ntrail][4]: slots of (i,k-1), (j,k-1), (i,j) and the tile row i if i == j else -1
};

__global__ __launch_bounds__(256) void chol_step_kernel(const SolverDev sv, const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, k = a.k;
  if ((int)blockIdx.x >= a.npanel) {
    // ---- trailing update of step k-1 ----
    double* A = smem; double* B = smem + T * TP; double* C = smem + 2 * T * TP;
    const int32_t* t4 = a.trail + 4 * (blockIdx.x - a.npanel);
    double* sij = tile_ptr(sv, t4[2]);
    load_tile(A, tile_ptr(sv, t4[0]), tid, false);
    load_tile(B, tile_ptr(sv, t4[1]), tid, false);
    load_tile(C, sij, tid, false);
    __syncthreads();
    tile_gemm_sub(C, A, B, tid);
    __syncthreads();
    store_tile(sij, C, tid);
    if (t4[3] >= 0 && tid < T) {
      const double* z = sv.rhs + (size_t)(k - 1) * T;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
      for (int m = 0; m < T; m += 2) { s0 += A[tid * TP + m] * z[m]; s1 += A[tid * TP + m + 1] * z[m + 1]; }
      sv.rhs[(size_t)t4[3] * T + tid] -= s0 + s1;
    }
    return;
  }
  // ---- panel of column k ----
  double* D = smem;                 // diagonal tile S_kk -> L_kk
  double* X = smem + T * TP;        // own tile S_ik
  double* Lp = smem + 2 * T * TP;   // L_k,k-1
  double* Lq = smem + 3 * T * TP;   // L_i,k-1, later the pivot reciprocals
  int* s_okp = reinterpret_cast<int*>(smem + 4 * T * TP);   // all LDS in the one dynamic region (16-B aligned base)
#define s_ok (*s_okp)
  const int b = blockIdx.x;
  const bool prev_k = a.prev_slot[0] >= 0, prev_i = b > 0 && prev_k && a.prev_slot[b] >= 0;
  load_tile(D, tile_ptr(sv, a.panel_slot[0]), tid, true);
  if (b > 0) load_tile(X, tile_ptr(sv, a.panel_slot[b]), tid, false);
  if (prev_k) load_tile(Lp, tile_ptr(sv, a.prev_slot[0]), tid, false);
  if (prev_i) load_tile(Lq, tile_ptr(sv, a.prev_slot[b]), tid, false);
  if (tid == 0) s_ok = 1;
  __syncthreads();
  if (prev_k) {
    tile_gemm_sub(D, Lp, Lp, tid);                 // S_kk -= L_k,k-1 L_k,k-1^T
    if (prev_i) tile_gemm_sub(X, Lq, Lp, tid);     // S_ik -= L_i,k-1 L_k,k-1^T
    __syncthreads();
  }
  const bool ok = potrf_blocked(D, tid);
  if (tid < 64 && !ok) s_ok = 0;
  if (tid < T) Lq[tid] = 1.0 / D[tid * TP + tid];   // pivot reciprocals (Lq is free after the pending update)
  __syncthreads();
  if (b == 0) {
    if (!s_ok && tid == 0) atomicExch(sv.chol_fail, 1);
    double* out = tile_ptr(sv, a.panel_slot[0]);
    for (int e = tid; e < T * T; e += 256) { const int r = e / T, c = e % T; if (c <= r) out[e] = D[r * TP + c]; }
    // forward substitution of the right-hand-side tile, first wave, lane r owns b_r:
    //   b_k -= L_k,k-1 z_k-1 (pending),  z_k = L_kk^-1 b_k
    if (tid < 64) {
      const int r = tid < T ? tid : T - 1;
      double bb = sv.rhs[(size_t)k * T + r];
      if (prev_k) {
        const double* z = sv.rhs + (size_t)(k - 1) * T;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int m = 0; m < T; m += 2) { s0 += Lp[r * TP + m] * z[m]; s1 += Lp[r * TP + m + 1] * z[m + 1]; }
        bb -= s0 + s1;
      }
      double col[T];
#pragma unroll
      for (int c = 0; c < T; ++c) col[c] = D[r * TP + c];     // row r of L_kk
      const double dinv = Lq[r];
#pragma unroll
      for (int c = 0; c < T; ++c) {
        const double zc = __shfl(bb * dinv, c, 64);             // z_c = b_c / L_cc, broadcast from lane c
        if (tid == c) bb = zc; else if (tid > c) bb -= col[c] * zc;
      }
      if (tid < T) sv.rhs[(size_t)k * T + tid] = bb;
    }
  } else {
    trsm_blocked(X, D, Lq, tid);                   // L_ik = S_ik L_kk^-T
    store_tile(tile_ptr(sv, a.panel_slot[b]), X, tid);
  }
#undef s_ok
}

// L^T y = z, one persistent workgroup: for k = nt-1 .. 0:  t = z_k - sum_{i>k} L_ik^T y_i ; solve L_kk^T y_k = t
// col_ptr / col_slot / col_row list, per tile column k, the diagonal tile first and then the tiles (i,k).
__global__ __launch_bounds__(256) void chol_backsolve_kernel(const SolverDev sv, const int32_t* col_ptr, const int32_t* col_slot, const int32_t* col_row) {
  __shared__ double A[T * TP];
  __shared__ double part[10][T];
  const int tid = threadIdx.x;
  const int c2 = tid % 24, rg = tid / 24;   // column pair, row group (rg < 10 for tid < 240)
  for (int k = sv.nt - 1; k >= 0; --k) {
    const int p0 = col_ptr[k], p1 = col_ptr[k + 1];
    load_tile(A, tile_ptr(sv, col_slot[p0]), tid, true);
    double s0 = 0.0, s1 = 0.0;
    if (rg < 10) {
      for (int p = p0 + 1; p < p1; ++p) {
        const double* lik = tile_ptr(sv, col_slot[p]) + 2 * c2;
        const double* yi = sv.rhs + (size_t)col_row[p] * T;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int r = rg + 10 * u;
          if (r < T) {
            const double2 v = *reinterpret_cast<const double2*>(lik + (size_t)r * T);
            const double y = yi[r];
            s0 += v.x * y; s1 += v.y * y;
          }
        }
      }
      part[rg][2 * c2] = s0; part[rg][2 * c2 + 1] = s1;
    }
    __syncthreads();
    if (tid < 64) {
      const int r = tid < T ? tid : T - 1;
      double t = sv.rhs[(size_t)k * T + r];
#pragma unroll
      for (int g = 0; g < 10; ++g) t -= part[g][r];
      double colr[T];   // column r of L_kk = row r of L_kk^T : L[cc][r] for cc >= r
#pragma unroll
      for (int cc = 0; cc < T; ++cc) colr[cc] = A[cc * TP + r];
      const double dinv = 1.0 / A[r * TP + r];
#pragma unroll
      for (int cc = T - 1; cc >= 0; --cc) {
        const double y = __shfl(t * dinv, cc, 64);
        if (tid == cc) t = y; else if (tid < cc) t -= colr[cc] * y;
      }
      if (tid < T) sv.rhs[(size_t)k * T + tid] = t;
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace

hipError_t launch_chol_step(const SolverDev& sv, int k, int npanel, const int32_t* panel_slot, const int32_t* prev_slot,
                            const int32_t* trail, int ntrail, hipStream_t st) {
  StepArgs a{k, npanel, panel_slot, prev_slot, trail};
  const size_t lds = (size_t)4 * T * TP * sizeof(double) + 16;   // 75 KB of the CU's 160 KB
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chol_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = true;
  }
  hipLaunchKernelGGL(chol_step_kernel, dim3(npanel + ntrail), dim3(256), lds, st, sv, a);
  return hipGetLastError();
}
hipError_t launch_chol_backsolve(const SolverDev& sv, const int32_t* col_ptr, const int32_t* col_slot, const int32_t* col_row, hipStream_t st) {
  hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1), dim3(256), 0, st, sv, col_ptr, col_slot, col_row);
  return hipGetLastError();
}

}  // namespace rsba
