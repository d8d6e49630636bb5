// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this repo uses
// (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 of a wide coalesced stream; other widths uncalibrated).
// Each kernel moves a known byte count; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void read_f64x2(const double2* __restrict__ a, size_t n, double* out) {
  double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
__global__ void read_f64(const double* __restrict__ a, size_t n, double* out) {
  double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 1.2345e300) out[0] = s;
}
__global__ void read_i32(const int* __restrict__ a, size_t n, double* out) {
  long s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 123456789012345L) out[0] = (double)s;
}
__global__ void write_f64(double* a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}
__global__ void write_f64x2(double2* a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_double2((double)i, 1.0);
}
// 32 component streams of 2 KB per workgroup tile (the tiled component-major Jacobian layout)
__global__ void write_tiled(double* a, size_t ntiles) {
  double* t = a + (size_t)blockIdx.x * 32 * 256 + threadIdx.x;
#pragma unroll
  for (int c = 0; c < 32; ++c) t[c * 256] = (double)c;
}
__global__ void write_tiled_nt(double* a, size_t ntiles) {
  double* t = a + (size_t)blockIdx.x * 32 * 256 + threadIdx.x;
#pragma unroll
  for (int c = 0; c < 32; ++c) __builtin_nontemporal_store((double)c, &t[c * 256]);
}
// tile written as 16-B stores: lane handles 2 adjacent observations of one component
__global__ void write_tiled_x2(double2* a, size_t ntiles) {
  double2* t = a + (size_t)blockIdx.x * 32 * 256 + threadIdx.x;   // 512-observation tile, 256 lanes x double2
#pragma unroll
  for (int c = 0; c < 32; ++c) t[c * 256] = make_double2((double)c, 1.0);
}
__global__ void write_tiled_x2_nt(double2* a, size_t ntiles) {
  double2* t = a + (size_t)blockIdx.x * 32 * 256 + threadIdx.x;
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d* tv = reinterpret_cast<v2d*>(t);
#pragma unroll
  for (int c = 0; c < 32; ++c) { v2d v = {(double)c, 1.0}; __builtin_nontemporal_store(v, &tv[c * 256]); }
}
__global__ void write_f64_nt(double* a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store((double)i, &a[i]);
}
// persistent variant of the tiled write: 2048 workgroups walk the tiles
__global__ void write_tiled_persist(double* a, size_t ntiles) {
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    double* t = a + tile * 32 * 256 + threadIdx.x;
#pragma unroll
    for (int c = 0; c < 32; ++c) t[c * 256] = (double)c;
  }
}
__global__ void copy_f64x2(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
  const size_t bytes = 1ull << 30;   // 1 GiB per kernel, far beyond the 256 MiB infinity cache
  void *a, *o; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&o, 64)); CK(hipMemset(a, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) { float ms; launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("%-12s %.1f GB/s (%zu bytes per launch)\n", name, bytes / (ms / 5 * 1e-3) / 1e9, bytes); };
  run("read_f64x2", [&] { read_f64x2<<<4096, 256>>>((double2*)a, bytes / 16, (double*)o); });
  run("read_f64", [&] { read_f64<<<4096, 256>>>((double*)a, bytes / 8, (double*)o); });
  run("read_i32", [&] { read_i32<<<4096, 256>>>((int*)a, bytes / 4, (double*)o); });
  run("write_f64", [&] { write_f64<<<4096, 256>>>((double*)a, bytes / 8); });
  run("write_f64x2", [&] { write_f64x2<<<4096, 256>>>((double2*)a, bytes / 16); });
  run("write_tiled", [&] { write_tiled<<<bytes / (32 * 256 * 8), 256>>>((double*)a, bytes / (32 * 256 * 8)); });
  run("tiled_nt", [&] { write_tiled_nt<<<bytes / (32 * 256 * 8), 256>>>((double*)a, 0); });
  run("tiled_x2", [&] { write_tiled_x2<<<bytes / (32 * 256 * 16), 256>>>((double2*)a, 0); });
  run("tiled_x2_nt", [&] { write_tiled_x2_nt<<<bytes / (32 * 256 * 16), 256>>>((double2*)a, 0); });
  run("write_f64_nt", [&] { write_f64_nt<<<4096, 256>>>((double*)a, bytes / 8); });
  run("tiled_pers", [&] { write_tiled_persist<<<2048, 256>>>((double*)a, bytes / (32 * 256 * 8)); });
  run("tiled_pers1k", [&] { write_tiled_persist<<<1024, 256>>>((double*)a, bytes / (32 * 256 * 8)); });
  run("copy_x2(r+w)", [&] { copy_f64x2<<<4096, 256>>>((double2*)a, (double2*)a + bytes / 32, bytes / 32); });
  return 0;
}
