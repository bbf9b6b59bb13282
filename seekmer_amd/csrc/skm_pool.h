// Caching allocator for HBM buffers and streams.  hipMalloc/hipFree and
// hipStreamCreate/Destroy cost 0.1-5 ms each on this platform and hipFree
// synchronises the device; a quantification handle is created per sample, so
// freed buffers and streams are parked per device and handed out again.
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
void pool_trim();                                     // really free everything parked

}  // namespace skm
