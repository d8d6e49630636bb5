// Device memory of handles and plans through a process-wide cache of freed blocks.
// windowedBA builds a new problem per frame (/root/reference/src/rsba/VideoSfMHandler.cc:185-214 -> CeresHandler per call): a handle
// of 100 cameras makes ~90 allocations and frees them a few milliseconds later — 4 ms of hipFree (each one synchronises the device)
// and ~1 ms of hipMalloc per call, a quarter of the whole BA() of that size.  Freed blocks are kept (per device, up to
// RSBA_DEVICE_CACHE_MB, default 2048) and handed out again to requests of about their size; rsba_release_host_scratch() returns them
// to the driver.  A block goes back to the cache only when nothing on the device can still touch it: the owners synchronise their
// streams first (rsba_destroy, rsba_destroy_solver) — hipFree did that implicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace rsba {

hipError_t dev_malloc(void** p, size_t bytes);
void dev_free(void* p);
void dev_release_cache();   // hipFree every cached block, destroy every pooled stream / event / pinned block (all devices)

// The same for the streams, events and the small pinned block a handle and its plan need (three streams and seven events per handle of
// a window: creating and destroying them was 1.0 of the 1.1 ms of rsba_destroy and ~0.4 ms of rsba_create + plan).  Streams are
// hipStreamNonBlocking; a stream or event goes back only when its owner has synchronised it.  Events: with / without timing.
hipError_t dev_stream_acquire(hipStream_t* s);
void dev_stream_release(hipStream_t s);
hipError_t dev_event_acquire(hipEvent_t* e, bool timing);
void dev_event_release(hipEvent_t e, bool timing);
hipError_t dev_pinned_acquire(void** p, size_t bytes);   // hipHostMalloc'd, device-visible; bytes <= 4096 (one size class)
void dev_pinned_release(void* p);


// Large host -> device uploads (the observations of rsba_create: 24 B each, 250 MB at 4k cameras) from PAGEABLE memory: hipMemcpy stages
// them through the runtime's own pinned buffer on one thread (10 - 12 GB/s measured: 23 ms of a 4k-camera rsba_create).  Here a few
// host threads copy 8 MB pieces into a process-wide pinned ring, each piece's DMA (57 GB/s) enqueued as it lands — the copies of one
// piece run under the DMA of another.  Segments below 8 MB in total go through hipMemcpy.  Returns when every byte is on the device.
struct UploadSeg { void* dst; const void* src; size_t bytes; };
hipError_t dev_upload_staged(const UploadSeg* segs, int nseg);

}  // namespace rsba
