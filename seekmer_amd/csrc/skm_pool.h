// Caching allocator for HBM buffers, streams, events and small pinned host
// blocks.  hipMalloc/hipFree, hipStreamCreate/Destroy and hipHostMalloc/Free
// cost 0.1-5 ms each on this platform and hipFree synchronises the device; a
// quantification handle is created per sample, so what it frees is parked per
// device and handed out again.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace skm {

// size class: next m * 2^e with m in 8..15 (at most 12.5 % slack)
size_t pool_round(size_t bytes);
hipError_t pool_alloc(void **out, size_t bytes);      // current device
void pool_free(void *p);                              // any device; no-op for nullptr
hipError_t pool_stream_acquire(hipStream_t *out);     // current device
void pool_stream_release(hipStream_t stream);
hipError_t pool_event_acquire(hipEvent_t *out, bool timing);   // current device
void pool_event_release(hipEvent_t event, bool timing);
constexpr size_t POOL_PINNED_BYTES = 4096;            // one size: control-block readbacks
hipError_t pool_pinned_acquire(void **out);
void pool_pinned_release(void *p);
void pool_trim();                                     // really free everything parked

}  // namespace skm
