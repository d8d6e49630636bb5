// K6  reduced camera system  S y = rhs  by a tile-sparse right-looking Cholesky factorisation, fp64.
//
// Replaces the CHOLMOD sparse Cholesky behind Ceres' SPARSE_SCHUR (reference call site
// /root/reference/src/rsba/CeresHandler.h:403,419).  S (npad x npad, lower triangle, row-major, ld)
// is treated as a grid of 48x48 tiles; the symbolic phase (host, once per problem) marks the tiles that
// are structurally non-zero after fill-in, and only those are touched.  rsba's problems are video:
// frames only share points with frames a few dozen positions away, so S is block-banded, fill stays
// inside the band, and the factorisation is O(n b^2) instead of O(n^3).
//
// Per tile column k, two launches:
//   panel : every workgroup factors the diagonal tile S_kk = L L^T in LDS (redundantly — 48^3/3 flops —
//           so no inter-workgroup hand-off is needed); workgroup 0 stores L_kk and forward-substitutes
//           the right-hand side tile, workgroup t>0 solves one sub-diagonal tile  L_ik = S_ik L_kk^-T.
//   update: one workgroup per tile pair (i >= j > k) of column k:  S_ij -= L_ik L_jk^T, the (i,i) ones
//           also carry  rhs_i -= L_ik z_k  (the forward solve rides along with the factorisation).
// The backward solve L^T y = z is one persistent workgroup walking the tile columns in reverse.
#include "solver_state.hpp"

namespace rsba {

namespace {

constexpr int T = kTile;
constexpr int TP = T + 1;   // LDS row pitch (doubles): odd pitch keeps column walks conflict-free

// Factor the T x T tile held in LDS (lower triangle, pitch TP) in place, by the first wave of the
// workgroup: lane i owns row i; column j needs dot products of rows i and j over the finished columns
// m < j, taken with four independent partial sums.  Returns false on a non-positive pivot.
__device__ __forceinline__ bool potrf_lds(double* A, int tid) {
  bool ok = true;
  if (tid < 64) {
    const int i = tid;
    for (int j = 0; j < T; ++j) {
      double s0 = 0.0, s1 = 0.0, d0 = 0.0, d1 = 0.0;
      if (i < T) {
        const double* ri = A + i * TP;
        const double* rj = A + j * TP;
        int m = 0;
        for (; m + 1 < j; m += 2) {
          const double a0 = rj[m], a1 = rj[m + 1];
          s0 += ri[m] * a0; s1 += ri[m + 1] * a1;
          d0 += a0 * a0; d1 += a1 * a1;
        }
        if (m < j) { const double a0 = rj[m]; s0 += ri[m] * a0; d0 += a0 * a0; }
      }
      const double djj = (i < T ? A[j * TP + j] : 1.0) - (d0 + d1);
      ok = ok && (djj > 0.0) && isfinite(djj);
      const double ljj = sqrt(djj);
      if (i < T && i >= j) {
        const double v = (i == j) ? ljj : (A[i * TP + j] - (s0 + s1)) / ljj;
        A[i * TP + j] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  return ok;
}

__global__ __launch_bounds__(256) void chol_panel_kernel(const SolverDev sv, int k, const int32_t* trsm_i) {
  __shared__ double A[T * TP];
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  const double* skk = sv.S + ((size_t)k * T) * sv.ld + (size_t)k * T;
  for (int e = tid; e < T * T; e += 256) { const int r = e / T, c = e % T; A[r * TP + c] = (c <= r) ? skk[(size_t)r * sv.ld + c] : 0.0; }
  if (tid == 0) s_ok = 1;
  __syncthreads();
  const bool ok = potrf_lds(A, tid);
  if (tid < 64 && !ok) s_ok = 0;
  __syncthreads();
  if (blockIdx.x == 0) {
    if (!s_ok && tid == 0) atomicExch(sv.chol_fail, 1);
    double* out = sv.S + ((size_t)k * T) * sv.ld + (size_t)k * T;
    for (int e = tid; e < T * T; e += 256) { const int r = e / T, c = e % T; if (c <= r) out[(size_t)r * sv.ld + c] = A[r * TP + c]; }
    // forward substitution of the right-hand side tile: z_k = L_kk^-1 b_k (first wave, lane r owns b_r)
    if (tid < 64) {
      double b = (tid < T) ? sv.rhs[(size_t)k * T + tid] : 0.0;
      for (int c = 0; c < T; ++c) {
        const double zc = __shfl(b, c, 64) / A[c * TP + c];
        if (tid == c) b = zc;
        else if (tid > c && tid < T) b -= A[tid * TP + c] * zc;
      }
      if (tid < T) sv.rhs[(size_t)k * T + tid] = b;
    }
  } else {
    // L_ik = S_ik L_kk^-T : thread r < T owns row r of the tile (forward substitution along the row)
    const int i = trsm_i[blockIdx.x - 1];
    double* sik = sv.S + ((size_t)i * T) * sv.ld + (size_t)k * T;
    if (tid < T) {
      double x[T];
      double* row = sik + (size_t)tid * sv.ld;
#pragma unroll
      for (int c = 0; c < T; ++c) x[c] = row[c];
#pragma unroll
      for (int c = 0; c < T; ++c) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const double* lc = A + c * TP;
#pragma unroll
        for (int m = 0; m < c; ++m) {
          const double t = x[m] * lc[m];
          if ((m & 3) == 0) s0 += t; else if ((m & 3) == 1) s1 += t; else if ((m & 3) == 2) s2 += t; else s3 += t;
        }
        x[c] = (x[c] - ((s0 + s1) + (s2 + s3))) / lc[c];
      }
#pragma unroll
      for (int c = 0; c < T; ++c) row[c] = x[c];
    }
  }
}

// S_ij -= L_ik L_jk^T ; 256 threads as 16 x 16, each a 3 x 3 micro-tile over K = 48
__global__ __launch_bounds__(256) void chol_update_kernel(const SolverDev sv, int k, const int32_t* upd_i, const int32_t* upd_j) {
  __shared__ double A[T * TP], B[T * TP];
  const int tid = threadIdx.x;
  const int i = upd_i[blockIdx.x], j = upd_j[blockIdx.x];
  const double* lik = sv.S + ((size_t)i * T) * sv.ld + (size_t)k * T;
  const double* ljk = sv.S + ((size_t)j * T) * sv.ld + (size_t)k * T;
  for (int e = tid; e < T * T; e += 256) {
    const int r = e / T, c = e % T;
    A[r * TP + c] = lik[(size_t)r * sv.ld + c];
    B[r * TP + c] = ljk[(size_t)r * sv.ld + c];
  }
  __syncthreads();
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
  double* sij = sv.S + ((size_t)i * T) * sv.ld + (size_t)j * T;
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) sij[(size_t)(ty * 3 + u) * sv.ld + tx * 3 + v] -= acc[u][v];
  if (i == j && tid < T) {
    // forward solve rides along: rhs_i -= L_ik z_k
    const double* zk = sv.rhs + (size_t)k * T;
    double s = 0.0;
    for (int m = 0; m < T; ++m) s += A[tid * TP + m] * zk[m];
    sv.rhs[(size_t)i * T + tid] -= s;
  }
}

// L^T y = z, one persistent workgroup: for k = nt-1 .. 0:  t = z_k - sum_{i>k} L_ik^T y_i ; solve L_kk^T y_k = t
__global__ __launch_bounds__(256) void chol_backsolve_kernel(const SolverDev sv, const int32_t* col_ptr, const int32_t* col_i) {
  __shared__ double A[T * TP];
  __shared__ double part[5][T];
  __shared__ double yk[T];
  const int tid = threadIdx.x;
  const int c = tid % T, g = tid / T;   // g < 5 for tid < 240
  for (int k = sv.nt - 1; k >= 0; --k) {
    const double* skk = sv.S + ((size_t)k * T) * sv.ld + (size_t)k * T;
    for (int e = tid; e < T * T; e += 256) { const int r = e / T, cc = e % T; A[r * TP + cc] = (cc <= r) ? skk[(size_t)r * sv.ld + cc] : 0.0; }
    double s = 0.0;
    if (g < 5) {
      for (int p = col_ptr[k]; p < col_ptr[k + 1]; ++p) {
        const int i = col_i[p];
        const double* lik = sv.S + ((size_t)i * T) * sv.ld + (size_t)k * T;
        const double* yi = sv.rhs + (size_t)i * T;
        for (int r = g; r < T; r += 5) s += lik[(size_t)r * sv.ld + c] * yi[r];
      }
      part[g][c] = s;
    }
    __syncthreads();
    if (tid < 64) {
      double t = (tid < T) ? sv.rhs[(size_t)k * T + tid] - (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid] + part[4][tid]) : 0.0;
      for (int cc = T - 1; cc >= 0; --cc) {
        const double y = __shfl(t, cc, 64) / A[cc * TP + cc];
        if (tid == cc) t = y;
        else if (tid < cc) t -= A[cc * TP + tid] * y;
      }
      if (tid < T) { yk[tid] = t; }
    }
    __syncthreads();
    if (tid < T) sv.rhs[(size_t)k * T + tid] = yk[tid];
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace

hipError_t launch_chol_panel(const SolverDev& sv, int k, const int32_t* trsm_i, int ntrsm, hipStream_t st) {
  hipLaunchKernelGGL(chol_panel_kernel, dim3(1 + ntrsm), dim3(256), 0, st, sv, k, trsm_i);
  return hipGetLastError();
}
hipError_t launch_chol_update(const SolverDev& sv, int k, const int32_t* upd_i, const int32_t* upd_j, int nupd, hipStream_t st) {
  if (nupd == 0) return hipSuccess;
  hipLaunchKernelGGL(chol_update_kernel, dim3(nupd), dim3(256), 0, st, sv, k, upd_i, upd_j);
  return hipGetLastError();
}
hipError_t launch_chol_backsolve(const SolverDev& sv, const int32_t* col_ptr, const int32_t* col_i, hipStream_t st) {
  hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1), dim3(256), 0, st, sv, col_ptr, col_i);
  return hipGetLastError();
}

}  // namespace rsba
