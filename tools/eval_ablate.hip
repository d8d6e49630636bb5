// Ablation of the residual+Jacobian kernel on a synthetic frame-major problem (no scene generator needed:
// timing only).  Variants: full kernel / no stores (results kept alive) / stores only (no math, same loads).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Irsba_amd/csrc tools/eval_ablate.hip rsba_amd/csrc/kernels_eval.hip -o tools/eval_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "device_state.hpp"
#include "obs_math.hpp"
using namespace rsba;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VAR>
__global__ __launch_bounds__(256) void k(const DeviceProblem dp) {
  const int tid = threadIdx.x; const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  const int64_t ic = i < dp.N ? i : dp.N - 1;
  const double2 xy = dp.xy[ic]; const int f = dp.obs_frame[ic], j = dp.obs_point[ic];
  double pose[12], X[3], cam[9];
  for (int q = 0; q < 12; ++q) pose[q] = dp.poses[(size_t)f * 12 + q];
  for (int q = 0; q < 3; ++q) X[q] = dp.points[(size_t)j * 3 + q];
  for (int q = 0; q < 9; ++q) cam[q] = dp.intr[q];
  ObsOut<true, 2> o;
  if (VAR != 2) { const Model m = {dp.shutter, dp.scan0, dp.scan1, dp.interp_rotation}; eval_observation<true, 2, true>(m, cam, pose, X, xy.x, xy.y, o); }
  else { o.r[0] = xy.x + X[0]; o.r[1] = xy.y + pose[3]; for (int q = 0; q < 15; ++q) { o.J[0][q] = X[q % 3] + q; o.J[1][q] = pose[q % 12] - q; } }
  if (VAR == 1) { double s = o.r[0] + o.r[1]; for (int q = 0; q < 15; ++q) s += o.J[0][q] * o.J[1][q]; if (s == 1.234e300) dp.res[0] = s; return; }
  double* rt = dp.res + (size_t)blockIdx.x * 512 + tid; rt[0] = o.r[0]; rt[256] = o.r[1];
  double* jt = dp.jac + (size_t)blockIdx.x * (30 * 256) + tid;
  for (int q = 0; q < 30; ++q) jt[q * 256] = o.J[q / 15][q % 15];
}

int main() {
  const int F = 1000, M = 100000; const int64_t N = 2041284;
  std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0, 1);
  std::vector<double> poses(F * 12), pts(M * 3), xy(2 * N), cam = {800, 800, -0.05, 0.01, 1e-3, -1e-3, 2e-3, 640, 360};
  std::vector<int32_t> of(N), op(N);
  for (int f = 0; f < F; ++f) for (int q = 0; q < 12; ++q) poses[f * 12 + q] = (q % 6 < 3) ? 0.05 * U(rng) : (q % 6 == 3 ? 0.8 * f : 0.1 * U(rng));
  for (int j = 0; j < M; ++j) { pts[3 * j] = 800.0 * j / M; pts[3 * j + 1] = 4 * U(rng) - 2; pts[3 * j + 2] = 8 + 6 * U(rng); }
  for (int64_t i = 0; i < N; ++i) { of[i] = (int)(i * F / N); op[i] = (int)std::min<int64_t>(M - 1, (int64_t)of[i] * M / F + (i % 2041) % 2000 - 1000 < 0 ? 0 : (int64_t)of[i] * M / F + (i % 2041) % 2000 - 1000); xy[2 * i] = 1280 * U(rng); xy[2 * i + 1] = 720 * U(rng); }
  DeviceProblem dp{}; dp.shutter = 1; dp.scan0 = 0; dp.scan1 = 1280; dp.interp_rotation = 1; dp.calibrated = 1; dp.P = 2; dp.F = F; dp.M = M; dp.NI = 1; dp.N = N; dp.K = 15;
  dp.ntiles = (N + 255) / 256;
  double *d_xy, *d_p, *d_X, *d_c, *d_res, *d_jac, *d_cp, *d_fp; int32_t *d_of, *d_op; int* d_fail;
  CK(hipMalloc(&d_xy, xy.size() * 8)); CK(hipMalloc(&d_p, poses.size() * 8)); CK(hipMalloc(&d_X, pts.size() * 8)); CK(hipMalloc(&d_c, 72));
  CK(hipMalloc(&d_of, N * 4)); CK(hipMalloc(&d_op, N * 4)); CK(hipMalloc(&d_res, dp.ntiles * 512 * 8)); CK(hipMalloc(&d_jac, dp.ntiles * 30 * 256 * 8));
  CK(hipMalloc(&d_cp, dp.ntiles * 8)); CK(hipMalloc(&d_fp, dp.ntiles * 8)); CK(hipMalloc(&d_fail, 4));
  CK(hipMemcpy(d_xy, xy.data(), xy.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_p, poses.data(), poses.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_X, pts.data(), pts.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_c, cam.data(), 72, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_of, of.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_op, op.data(), N * 4, hipMemcpyHostToDevice));
  dp.xy = (double2*)d_xy; dp.obs_frame = d_of; dp.obs_point = d_op; dp.poses = d_p; dp.points = d_X; dp.intr = d_c; dp.res = d_res; dp.jac = d_jac;
  dp.cost_partial = d_cp; dp.fixed_partial = d_fp; dp.fail_partial = d_fp; dp.fail_count = d_fail;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) { float ms; for (int i = 0; i < 3; ++i) launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("%-28s %.1f us\n", name, ms / 20 * 1e3); };
  const int grid = (int)dp.ntiles;
  hipStream_t nb; CK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking));
  auto run_s = [&](const char* name, auto launch) { float ms; for (int i = 0; i < 3; ++i) launch(); hipStreamSynchronize(nb); hipEventRecord(e0, nb); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1, nb); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("%-28s %.1f us\n", name, ms / 20 * 1e3); };
  for (int rep = 0; rep < 3; ++rep) run_s("product, non-blocking stream", [&] { launch_eval(dp, kRawJacobian, nb); });
  // extra allocations like the library's solver state (1.7 GB) to see whether placement matters
  void* extra[4]; for (int q = 0; q < 4; ++q) CK(hipMalloc(&extra[q], (size_t)450 << 20));
  double* jac2; CK(hipMalloc(&jac2, dp.ntiles * 30 * 256 * 8));
  { DeviceProblem d2 = dp; d2.jac = jac2; for (int rep = 0; rep < 2; ++rep) run("product, later allocation", [&] { launch_eval(d2, kRawJacobian, 0); }); }
  for (int rep = 0; rep < 2; ++rep) {
    run("product kernel", [&] { launch_eval(dp, kRawJacobian, 0); });
    run("simple: math + 8B stores", [&] { k<0><<<grid, 256>>>(dp); });
    run("simple: math, no stores", [&] { k<1><<<grid, 256>>>(dp); });
    run("simple: stores, no math", [&] { k<2><<<grid, 256>>>(dp); });
    run("residual-only product", [&] { launch_eval(dp, kResidualOnly, 0); });
  }
  return 0;
}
