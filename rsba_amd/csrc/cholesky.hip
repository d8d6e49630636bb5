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
// A band factored column after column is a serial chain of ~n/48 dependent steps (250 at 1k cameras),
// each a few tens of microseconds of latency: the GPU idles.  So the host reorders the tile columns by
// nested dissection (BFS-level separators cut the band into independent segments), computes the levels of
// the resulting elimination structure, and the factorisation runs level by level: two launches per level
// (diagonal tiles, then sub-diagonal tiles), every tile of a level in its own workgroup; the forward solve
// rides along, the backward solve walks the levels in reverse.  1k cameras: ~50 levels instead of 250 steps.
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
// A tile in flight: each of the 256 threads holds 9 of its 2304 doubles, so the HBM/L2 latency of the next
// contributor overlaps the 48^3 product on the current one.
struct TileRegs {
  double v[9];
  __device__ __forceinline__ void fetch(const double* src, int tid) {
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = src[tid + 256 * k];
  }
  __device__ __forceinline__ void commit(double* lds, int tid) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int e = tid + 256 * k; lds[(e / T) * TP + e % T] = v[k]; }
  }
};

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

// ---- level-scheduled left-looking kernels -----------------------------------------------------------
// The host orders the tile columns by nested dissection and groups them into levels of the elimination
// structure; all columns of a level are independent.  Left-looking: a tile pulls every update it needs
// from finished columns (no two workgroups ever write the same tile, so no atomics and a fixed summation
// order), then the diagonal tile is factored / the sub-diagonal tile is solved.

// one workgroup per chunk of a long contributor list: partial = sum_{k in chunk} L_ik L_jk^T (and, for a
// diagonal tile, sum L_jk z_k) to scratch; upd item = {kind, list begin, list end, scratch slot}
__global__ __launch_bounds__(256) void chol_update_kernel(const SolverDev sv, const int32_t* upd, const int32_t* diag_list, const int32_t* sub_list) {
  __shared__ double A[T * TP], B[T * TP];
  const int tid = threadIdx.x;
  const int32_t* u = upd + 4 * blockIdx.x;
  const bool diag = u[0] == 0;
  const int32_t* list = diag ? diag_list : sub_list;
  double* out = sv.chol_part + (size_t)u[3] * (T * T + T);
  const int ty = tid >> 4, tx = tid & 15;
  double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double bacc = 0.0;
  TileRegs ra, rb;
  ra.fetch(tile_ptr(sv, list[2 * u[1]]), tid);
  if (!diag) rb.fetch(tile_ptr(sv, list[2 * u[1] + 1]), tid);
  for (int p = u[1]; p < u[2]; ++p) {
    __syncthreads();
    ra.commit(A, tid);
    if (!diag) rb.commit(B, tid);
    __syncthreads();
    if (p + 1 < u[2]) {   // next contributor's tiles travel while this one is multiplied
      ra.fetch(tile_ptr(sv, list[2 * (p + 1)]), tid);
      if (!diag) rb.fetch(tile_ptr(sv, list[2 * (p + 1) + 1]), tid);
    }
    const double* Bm = diag ? A : B;
#pragma unroll 4
    for (int m = 0; m < T; ++m) {
      double a[3], b[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) { a[q] = A[(ty * 3 + q) * TP + m]; b[q] = Bm[(tx * 3 + q) * TP + m]; }
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int v = 0; v < 3; ++v) acc[q][v] += a[q] * b[v];
    }
    if (diag && tid < T) {
      const double* z = sv.rhs + (size_t)list[2 * p + 1] * T;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
      for (int m = 0; m < T; m += 2) { s0 += A[tid * TP + m] * z[m]; s1 += A[tid * TP + m + 1] * z[m + 1]; }
      bacc += s0 + s1;
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int v = 0; v < 3; ++v) out[(ty * 3 + q) * T + tx * 3 + v] = acc[q][v];
  if (diag && tid < T) out[T * T + tid] = bacc;
}

// one workgroup per column j of the level:
//   S_jj -= sum_k L_jk L_jk^T ;  b_j -= sum_k L_jk z_k ;  S_jj = L_jj L_jj^T ;  z_j = L_jj^-1 b_j
__global__ __launch_bounds__(256) void chol_diag_kernel(const SolverDev sv, const int32_t* info, const int32_t* ptr, const int32_t* list) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* D = smem; double* A = smem + T * TP; double* bvec = smem + 2 * T * TP; int* s_okp = reinterpret_cast<int*>(bvec + T);
  const int tid = threadIdx.x, b = blockIdx.x;
  const int slot_jj = info[4 * b], tile_j = info[4 * b + 1], part0 = info[4 * b + 2], nparts = info[4 * b + 3];
  load_tile(D, tile_ptr(sv, slot_jj), tid, true);
  if (tid < T) bvec[tid] = sv.rhs[(size_t)tile_j * T + tid];
  if (tid == 0) *s_okp = 1;
  __syncthreads();
  for (int c = 0; c < nparts; ++c) {   // long contributor lists arrive pre-reduced (chol_update_kernel)
    const double* part = sv.chol_part + (size_t)(part0 + c) * (T * T + T);
    for (int e = tid; e < T * T; e += 256) D[(e / T) * TP + e % T] -= part[e];
    if (tid < T) bvec[tid] -= part[T * T + tid];
  }
  if (nparts > 0) __syncthreads();
  TileRegs ra;
  if (nparts == 0 && ptr[b] < ptr[b + 1]) ra.fetch(tile_ptr(sv, list[2 * ptr[b]]), tid);
  for (int p = ptr[b]; nparts == 0 && p < ptr[b + 1]; ++p) {
    ra.commit(A, tid);
    __syncthreads();
    if (p + 1 < ptr[b + 1]) ra.fetch(tile_ptr(sv, list[2 * (p + 1)]), tid);
    tile_gemm_sub(D, A, A, tid);
    if (tid < T) {
      const double* z = sv.rhs + (size_t)list[2 * p + 1] * T;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
      for (int m = 0; m < T; m += 2) { s0 += A[tid * TP + m] * z[m]; s1 += A[tid * TP + m + 1] * z[m + 1]; }
      bvec[tid] -= s0 + s1;
    }
    __syncthreads();
  }
  const bool ok = potrf_blocked(D, tid);
  if (tid < 64 && !ok) *s_okp = 0;
  __syncthreads();
  if (!*s_okp && tid == 0) atomicExch(sv.chol_fail, 1);
  double* out = tile_ptr(sv, slot_jj);
  for (int e = tid; e < T * T; e += 256) { const int r = e / T, c = e % T; if (c <= r) out[e] = D[r * TP + c]; }
  if (tid < 64) {
    const int r = tid < T ? tid : T - 1;
    double bb = bvec[r];
    double colv[T];
#pragma unroll
    for (int c = 0; c < T; ++c) colv[c] = D[r * TP + c];     // row r of L_jj
    const double dinv = 1.0 / D[r * TP + r];
#pragma unroll
    for (int c = 0; c < T; ++c) {
      const double zc = __shfl(bb * dinv, c, 64);               // z_c = b_c / L_cc, broadcast from lane c
      if (tid == c) bb = zc; else if (tid > c) bb -= colv[c] * zc;
    }
    if (tid < T) sv.rhs[(size_t)tile_j * T + tid] = bb;
  }
}

// one workgroup per sub-diagonal tile (i,j) of the level's columns:
//   S_ij -= sum_k L_ik L_jk^T ;  L_ij = S_ij L_jj^-T
__global__ __launch_bounds__(256) void chol_sub_kernel(const SolverDev sv, const int32_t* info, const int32_t* ptr, const int32_t* list) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* X = smem; double* A = smem + T * TP; double* B = smem + 2 * T * TP; double* dinv = smem + 3 * T * TP;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int slot_ij = info[4 * b], slot_jj = info[4 * b + 1], part0 = info[4 * b + 2], nparts = info[4 * b + 3];
  load_tile(X, tile_ptr(sv, slot_ij), tid, false);
  __syncthreads();
  for (int c = 0; c < nparts; ++c) {
    const double* part = sv.chol_part + (size_t)(part0 + c) * (T * T + T);
    for (int e = tid; e < T * T; e += 256) X[(e / T) * TP + e % T] -= part[e];
  }
  if (nparts > 0) __syncthreads();
  TileRegs ra, rb;
  if (nparts == 0 && ptr[b] < ptr[b + 1]) { ra.fetch(tile_ptr(sv, list[2 * ptr[b]]), tid); rb.fetch(tile_ptr(sv, list[2 * ptr[b] + 1]), tid); }
  for (int p = ptr[b]; nparts == 0 && p < ptr[b + 1]; ++p) {
    ra.commit(A, tid); rb.commit(B, tid);
    __syncthreads();
    if (p + 1 < ptr[b + 1]) { ra.fetch(tile_ptr(sv, list[2 * (p + 1)]), tid); rb.fetch(tile_ptr(sv, list[2 * (p + 1) + 1]), tid); }
    tile_gemm_sub(X, A, B, tid);
    __syncthreads();
  }
  load_tile(A, tile_ptr(sv, slot_jj), tid, true);   // L_jj, finished by chol_diag_kernel of this level
  __syncthreads();
  if (tid < T) dinv[tid] = 1.0 / A[tid * TP + tid];
  __syncthreads();
  trsm_blocked(X, A, dinv, tid);
  store_tile(tile_ptr(sv, slot_ij), X, tid);
}

// backward solve, levels in reverse; one workgroup per column j:  y_j = L_jj^-T ( z_j - sum_i L_ij^T y_i )
__global__ __launch_bounds__(256) void chol_back_kernel(const SolverDev sv, const int32_t* info, const int32_t* ptr, const int32_t* list) {
  __shared__ double A[T * TP];
  __shared__ double part[10][T];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int c2 = tid % 24, rg = tid / 24;   // column pair, row group (rg < 10 for tid < 240)
  const int slot_jj = info[2 * b], tile_j = info[2 * b + 1];
  load_tile(A, tile_ptr(sv, slot_jj), tid, true);
  double s0 = 0.0, s1 = 0.0;
  if (rg < 10) {
    for (int p = ptr[b]; p < ptr[b + 1]; ++p) {
      const double* lij = tile_ptr(sv, list[2 * p]) + 2 * c2;
      const double* yi = sv.rhs + (size_t)list[2 * p + 1] * T;
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int r = rg + 10 * u;
        if (r < T) {
          const double2 v = *reinterpret_cast<const double2*>(lij + (size_t)r * T);
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
    double t = sv.rhs[(size_t)tile_j * T + r];
#pragma unroll
    for (int g = 0; g < 10; ++g) t -= part[g][r];
    double colr[T];   // column r of L_jj = row r of L_jj^T
#pragma unroll
    for (int cc = 0; cc < T; ++cc) colr[cc] = A[cc * TP + r];
    const double dinv = 1.0 / A[r * TP + r];
#pragma unroll
    for (int cc = T - 1; cc >= 0; --cc) {
      const double y = __shfl(t * dinv, cc, 64);
      if (tid == cc) t = y; else if (tid < cc) t -= colr[cc] * y;
    }
    if (tid < T) sv.rhs[(size_t)tile_j * T + tid] = t;
  }
}

}  // namespace

namespace {
template <class K>
hipError_t set_lds(K kernel, size_t bytes, bool& configured) {
  if (configured) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) configured = true;
  return e;
}
}  // namespace

hipError_t launch_chol_update(const SolverDev& sv, int nitem, const int32_t* upd, const int32_t* diag_list, const int32_t* sub_list, hipStream_t st) {
  if (nitem <= 0) return hipSuccess;
  hipLaunchKernelGGL(chol_update_kernel, dim3(nitem), dim3(256), 0, st, sv, upd, diag_list, sub_list);
  return hipGetLastError();
}
hipError_t launch_chol_diag(const SolverDev& sv, int ncol, const int32_t* info, const int32_t* ptr, const int32_t* list, hipStream_t st) {
  if (ncol <= 0) return hipSuccess;
  const size_t lds = (size_t)(2 * T * TP + T) * sizeof(double) + 16;
  static bool configured = false;
  hipError_t e = set_lds(chol_diag_kernel, lds, configured);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(chol_diag_kernel, dim3(ncol), dim3(256), lds, st, sv, info, ptr, list);
  return hipGetLastError();
}
hipError_t launch_chol_sub(const SolverDev& sv, int ntile, const int32_t* info, const int32_t* ptr, const int32_t* list, hipStream_t st) {
  if (ntile <= 0) return hipSuccess;
  const size_t lds = (size_t)(3 * T * TP + T) * sizeof(double) + 16;
  static bool configured = false;
  hipError_t e = set_lds(chol_sub_kernel, lds, configured);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(chol_sub_kernel, dim3(ntile), dim3(256), lds, st, sv, info, ptr, list);
  return hipGetLastError();
}
hipError_t launch_chol_back(const SolverDev& sv, int ncol, const int32_t* info, const int32_t* ptr, const int32_t* list, hipStream_t st) {
  if (ncol <= 0) return hipSuccess;
  hipLaunchKernelGGL(chol_back_kernel, dim3(ncol), dim3(256), 0, st, sv, info, ptr, list);
  return hipGetLastError();
}

}  // namespace rsba
