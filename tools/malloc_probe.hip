// What hipMalloc / hipFree cost by block size on this box, and pageable against pinned-staged against registered uploads of 256 MiB (round 6: the
// one-arena plan the review asked for was not built after this: one 60 GiB hipMalloc is 0.3 ms, sixty of 1 GiB 1.1 ms — and one of them 5.3 s right after
// 960 frees; rsba_create's upload went to a pinned ring instead: tools/../rsba_amd/csrc/devmem.hip, dev_upload_staged).
//   hipcc --offload-arch=gfx950 -O2 tools/malloc_probe.hip -o tools/_tmp/malloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  for (int rep = 0; rep < 2; ++rep) {
    { double t0 = now(); void* p; hipError_t e = hipMalloc(&p, 60ull << 30); double t1 = now(); hipMemset(p, 0, 1 << 20); hipDeviceSynchronize(); double t2 = now(); hipFree(p); double t3 = now();
      printf("one block of 60 GiB: hipMalloc %.1f ms (%s), first touch %.1f ms, hipFree %.1f ms\n", t1 - t0, hipGetErrorString(e), t2 - t1, t3 - t2); }
    { std::vector<void*> v; double t0 = now(); for (int i = 0; i < 60; ++i) { void* p; hipMalloc(&p, 1ull << 30); v.push_back(p); } double t1 = now(); for (void* p : v) hipFree(p); double t2 = now();
      printf("60 blocks of 1 GiB:  hipMalloc %.1f ms, hipFree %.1f ms\n", t1 - t0, t2 - t1); }
    { std::vector<void*> v; double t0 = now(); for (int i = 0; i < 960; ++i) { void* p; hipMalloc(&p, 64ull << 20); v.push_back(p); } double t1 = now(); for (void* p : v) hipFree(p); double t2 = now();
      printf("960 blocks of 64 MiB: hipMalloc %.1f ms, hipFree %.1f ms\n", t1 - t0, t2 - t1); }
    { std::vector<void*> v; double t0 = now(); for (int i = 0; i < 90; ++i) { void* p; hipMalloc(&p, 1ull << 20); v.push_back(p); } double t1 = now(); for (void* p : v) hipFree(p); double t2 = now();
      printf("90 blocks of 1 MiB: hipMalloc %.2f ms, hipFree %.2f ms\n", t1 - t0, t2 - t1); }
    // pageable vs pinned-staged upload of 256 MB
    { size_t n = 256ull << 20; char* src = (char*)malloc(n); for (size_t i = 0; i < n; i += 4096) src[i] = 1; void* d; hipMalloc(&d, n);
      double t0 = now(); hipMemcpy(d, src, n, hipMemcpyHostToDevice); double t1 = now();
      void* pin; hipHostMalloc(&pin, n); double t2 = now(); memcpy(pin, src, n); double t3 = now(); hipMemcpy(d, pin, n, hipMemcpyHostToDevice); double t4 = now();
      double t5 = now(); hipHostRegister(src, n, hipHostRegisterDefault); double t6 = now(); hipMemcpy(d, src, n, hipMemcpyHostToDevice); double t7 = now(); hipHostUnregister(src); double t8 = now();
      printf("256 MiB upload: pageable hipMemcpy %.1f ms | memcpy to pinned %.1f + DMA %.1f ms | hipHostRegister %.1f + DMA %.1f + unregister %.1f ms\n", t1 - t0, t3 - t2, t4 - t3, t6 - t5, t7 - t6, t8 - t7);
      hipHostFree(pin); hipFree(d); free(src); }
  }
  return 0;
}
