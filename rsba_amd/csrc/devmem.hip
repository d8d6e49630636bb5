#include "devmem.hpp"

#include <cstdlib>
#include <cstring>
#include <thread>
#include <map>
#include <mutex>
#include <algorithm>
#include <unordered_map>
#include <vector>

namespace rsba {

namespace {

struct Block { size_t bytes; int device; };
struct Cache {
  std::mutex m;
  std::unordered_map<void*, Block> live;                    // blocks handed out (their rounded size)
  std::map<int, std::multimap<size_t, void*>> free_blocks;   // per device, by size
  size_t cached_bytes = 0, cap_bytes = 0;
  bool cap_read = false;
  std::map<int, std::vector<hipStream_t>> streams;           // idle, per device
  std::map<int, std::vector<hipEvent_t>> events[2];          // [timing]
  std::map<int, std::vector<void*>> pinned;                  // 4 KB blocks
  std::map<int, void*> staging;                              // the pinned ring of dev_upload_staged (one per device, kStageBytes)
  std::mutex staging_m;                                      // one staged upload at a time (a second caller takes hipMemcpy)
};
constexpr int kStageThreads = 4;
constexpr size_t kStageChunk = 8u << 20, kStageBytes = (size_t)kStageThreads * 2 * kStageChunk;
constexpr size_t kMaxStreams = 16, kMaxEvents = 64, kMaxPinned = 8, kPinnedBytes = 4096;
Cache& cache() { static Cache* c = new Cache(); return *c; }   // (never destroyed: handles may outlive static destructors)

// sizes in steps of 1/8 octave above 64 KB (a problem a frame longer than the last one still finds its blocks), 256 B below
size_t round_size(size_t bytes) {
  if (bytes <= 256) return 256;
  if (bytes <= (64u << 10)) return (bytes + 255) & ~(size_t)255;
  size_t step = 1;
  while ((step << 4) <= bytes) step <<= 1;   // step = 2^(floor(log2 bytes) - 3)
  return (bytes + step - 1) & ~(step - 1);
}

void read_cap(Cache& c) {   // (under the lock)
  if (c.cap_read) return;
  c.cap_read = true;
  size_t mb = 2048;
  if (const char* e = std::getenv("RSBA_DEVICE_CACHE_MB")) mb = (size_t)std::strtoull(e, nullptr, 10);
  c.cap_bytes = mb << 20;
}
// hipFree every cached block of every device (the pools of streams, events and pinned blocks are not touched)
void release_blocks(Cache& c) {
  std::map<int, std::multimap<size_t, void*>> take;
  { std::lock_guard<std::mutex> lk(c.m); take.swap(c.free_blocks); c.cached_bytes = 0; }
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  for (auto& d : take) {
    if (d.second.empty()) continue;
    (void)hipSetDevice(d.first);
    for (auto& kv : d.second) (void)hipFree(kv.second);
  }
  if (have) (void)hipSetDevice(cur);
}

}  // namespace

hipError_t dev_malloc(void** p, size_t bytes) {
  Cache& c = cache();
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  // (a block larger than the whole cache can never be cached: no point in rounding it up by up to an eighth — that was up to 12 % of HBM
  // wasted per large buffer of a 20 - 80 M observation scene)
  size_t cap = 0;
  { std::lock_guard<std::mutex> lk(c.m); read_cap(c); cap = c.cap_bytes; }
  const size_t want = bytes > cap ? ((bytes + 255) & ~(size_t)255) : round_size(bytes);
  {
    std::lock_guard<std::mutex> lk(c.m);
    auto& fl = c.free_blocks[dev];
    auto it = fl.lower_bound(want);
    if (it != fl.end() && it->first <= want + want / 4) {   // (at most a quarter wasted)
      *p = it->second;
      c.live[*p] = Block{it->first, dev};
      c.cached_bytes -= it->first;
      fl.erase(it);
      return hipSuccess;
    }
  }
  e = hipMalloc(p, want);
  if (e == hipErrorOutOfMemory) {   // give the cached BLOCKS back (the pooled streams / events / pinned blocks stay) and try once more
    release_blocks(c);
    e = hipMalloc(p, want);
  }
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(c.m);
  c.live[*p] = Block{want, dev};
  return hipSuccess;
}

void dev_free(void* p) {
  if (!p) return;
  Cache& c = cache();
  {
    std::lock_guard<std::mutex> lk(c.m);
    read_cap(c);
    auto it = c.live.find(p);
    if (it != c.live.end()) {
      const Block b = it->second;
      c.live.erase(it);
      if (c.cached_bytes + b.bytes <= c.cap_bytes) { c.free_blocks[b.device].emplace(b.bytes, p); c.cached_bytes += b.bytes; return; }
    }
  }
  (void)hipFree(p);
}

void dev_release_cache() {
  Cache& c = cache();
  std::map<int, std::multimap<size_t, void*>> take;
  std::map<int, std::vector<hipStream_t>> streams;
  std::map<int, std::vector<hipEvent_t>> events[2];
  std::map<int, std::vector<void*>> pinned;
  std::map<int, void*> staging;
  {
    std::lock_guard<std::mutex> sl(c.staging_m);   // (no upload is using the ring)
    std::lock_guard<std::mutex> lk(c.m);
    take.swap(c.free_blocks);
    c.cached_bytes = 0;
    streams.swap(c.streams); events[0].swap(c.events[0]); events[1].swap(c.events[1]); pinned.swap(c.pinned); staging.swap(c.staging);
  }
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  for (auto& d : take) {
    if (d.second.empty()) continue;
    (void)hipSetDevice(d.first);
    for (auto& kv : d.second) (void)hipFree(kv.second);
  }
  for (auto& d : streams) { (void)hipSetDevice(d.first); for (hipStream_t s : d.second) (void)hipStreamDestroy(s); }
  for (auto& ev : events) for (auto& d : ev) { (void)hipSetDevice(d.first); for (hipEvent_t e : d.second) (void)hipEventDestroy(e); }
  for (auto& d : pinned) { (void)hipSetDevice(d.first); for (void* p : d.second) (void)hipHostFree(p); }
  for (auto& d : staging) { (void)hipSetDevice(d.first); if (d.second) (void)hipHostFree(d.second); }
  if (have) (void)hipSetDevice(cur);
}

hipError_t dev_stream_acquire(hipStream_t* s) {
  Cache& c = cache();
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lk(c.m);
    auto& v = c.streams[dev];
    if (!v.empty()) { *s = v.back(); v.pop_back(); return hipSuccess; }
  }
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
void dev_stream_release(hipStream_t s) {
  if (!s) return;
  Cache& c = cache();
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    std::lock_guard<std::mutex> lk(c.m);
    auto& v = c.streams[dev];
    if (v.size() < kMaxStreams) { v.push_back(s); return; }
  }
  (void)hipStreamDestroy(s);
}
hipError_t dev_event_acquire(hipEvent_t* ev, bool timing) {
  Cache& c = cache();
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lk(c.m);
    auto& v = c.events[timing ? 1 : 0][dev];
    if (!v.empty()) { *ev = v.back(); v.pop_back(); return hipSuccess; }
  }
  return timing ? hipEventCreate(ev) : hipEventCreateWithFlags(ev, hipEventDisableTiming);
}
void dev_event_release(hipEvent_t ev, bool timing) {
  if (!ev) return;
  Cache& c = cache();
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    std::lock_guard<std::mutex> lk(c.m);
    auto& v = c.events[timing ? 1 : 0][dev];
    if (v.size() < kMaxEvents) { v.push_back(ev); return; }
  }
  (void)hipEventDestroy(ev);
}
hipError_t dev_pinned_acquire(void** p, size_t bytes) {
  if (bytes > kPinnedBytes) return hipErrorInvalidValue;
  Cache& c = cache();
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lk(c.m);
    auto& v = c.pinned[dev];
    if (!v.empty()) { *p = v.back(); v.pop_back(); return hipSuccess; }
  }
  return hipHostMalloc(p, kPinnedBytes, hipHostMallocDefault);
}
void dev_pinned_release(void* p) {
  if (!p) return;
  Cache& c = cache();
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    std::lock_guard<std::mutex> lk(c.m);
    auto& v = c.pinned[dev];
    if (v.size() < kMaxPinned) { v.push_back(p); return; }
  }
  (void)hipHostFree(p);
}


hipError_t dev_upload_staged(const UploadSeg* segs, int nseg) {
  size_t total = 0;
  for (int i = 0; i < nseg; ++i) total += segs[i].bytes;
  auto plain = [&]() -> hipError_t {
    for (int i = 0; i < nseg; ++i) if (segs[i].bytes) { const hipError_t e = hipMemcpy(segs[i].dst, segs[i].src, segs[i].bytes, hipMemcpyHostToDevice); if (e != hipSuccess) return e; }
    return hipSuccess;
  };
  if (total < kStageChunk) return plain();
  Cache& c = cache();
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::unique_lock<std::mutex> ring(c.staging_m, std::try_to_lock);
  if (!ring.owns_lock()) return plain();
  char* base = nullptr;
  {
    std::lock_guard<std::mutex> lk(c.m);
    auto it = c.staging.find(dev);
    if (it != c.staging.end()) base = static_cast<char*>(it->second);
  }
  if (!base) {
    void* q = nullptr;
    if (hipHostMalloc(&q, kStageBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return plain(); }
    base = static_cast<char*>(q);
    std::lock_guard<std::mutex> lk(c.m);
    c.staging[dev] = q;
  }
  struct Piece { int seg; size_t off, len; };
  std::vector<Piece> pieces;
  for (int i = 0; i < nseg; ++i) for (size_t off = 0; off < segs[i].bytes; off += kStageChunk) pieces.push_back(Piece{i, off, std::min(kStageChunk, segs[i].bytes - off)});
  const int T = (int)std::min<size_t>(kStageThreads, pieces.size());
  std::vector<hipError_t> err((size_t)T, hipSuccess);
  auto work = [&](int t) {
    hipError_t& er = err[(size_t)t];
    if ((er = hipSetDevice(dev)) != hipSuccess) return;
    hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr};
    if ((er = dev_stream_acquire(&st)) != hipSuccess) return;
    if ((er = dev_event_acquire(&ev[0], false)) == hipSuccess) er = dev_event_acquire(&ev[1], false);
    char* slot[2] = {base + ((size_t)t * 2) * kStageChunk, base + ((size_t)t * 2 + 1) * kStageChunk};
    int k = 0;
    for (size_t q = (size_t)t; q < pieces.size() && er == hipSuccess; q += (size_t)T, ++k) {
      const Piece& pc = pieces[q];
      const int sl = k & 1;
      if (k >= 2 && (er = hipEventSynchronize(ev[sl])) != hipSuccess) break;   // (the DMA that last read this slot is done)
      std::memcpy(slot[sl], static_cast<const char*>(segs[pc.seg].src) + pc.off, pc.len);
      if ((er = hipMemcpyAsync(static_cast<char*>(segs[pc.seg].dst) + pc.off, slot[sl], pc.len, hipMemcpyHostToDevice, st)) != hipSuccess) break;
      er = hipEventRecord(ev[sl], st);
    }
    const hipError_t es = hipStreamSynchronize(st);   // (whatever happened: nothing of this thread's is in flight when the ring is handed on)
    if (er == hipSuccess) er = es;
    if (ev[0]) dev_event_release(ev[0], false);
    if (ev[1]) dev_event_release(ev[1], false);
    dev_stream_release(st);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < T; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  for (hipError_t x : err) if (x != hipSuccess) return x;
  return hipSuccess;
}

}  // namespace rsba
