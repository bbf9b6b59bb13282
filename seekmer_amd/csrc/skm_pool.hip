#include "skm_pool.h"

#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace skm {

namespace {

struct Block { int device; size_t bytes; };
std::mutex g_mu;
std::unordered_map<void *, Block> g_live;                       // handed out
std::map<std::pair<int, size_t>, std::vector<void *>> g_parked; // by (device, size class)
std::unordered_map<int, std::vector<hipStream_t>> g_streams;
std::unordered_map<hipStream_t, int> g_stream_device;
std::unordered_map<int, std::vector<hipEvent_t>> g_events[2];     // [timing]
std::vector<void *> g_pinned;
size_t g_parked_bytes = 0;
constexpr size_t PARK_LIMIT = 48ULL << 30;                      // beyond this, free for real

}  // namespace

size_t pool_round(size_t bytes)
{
    if (bytes < 4096) return 4096;
    int e = 0;
    size_t v = bytes - 1;
    while ((v >> e) >= 16) ++e;
    return (((v >> e) + 1) << e);
}

hipError_t pool_alloc(void **out, size_t bytes)
{
    int device = 0;
    hipError_t err = hipGetDevice(&device);
    if (err != hipSuccess) return err;
    const size_t cls = pool_round(bytes);
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_parked.find({device, cls});
        if (it != g_parked.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            g_parked_bytes -= cls;
            g_live[p] = Block{device, cls};
            *out = p;
            return hipSuccess;
        }
    }
    void *p = nullptr;
    err = hipMalloc(&p, cls);
    if (err != hipSuccess) {
        pool_trim();                                  // give parked memory back and retry once
        err = hipMalloc(&p, cls);
        if (err != hipSuccess) return err;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    g_live[p] = Block{device, cls};
    *out = p;
    return hipSuccess;
}

void pool_free(void *p)
{
    if (!p) return;
    Block blk;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) return;
        blk = it->second;
        g_live.erase(it);
        if (g_parked_bytes + blk.bytes <= PARK_LIMIT) {
            g_parked[{blk.device, blk.bytes}].push_back(p);
            g_parked_bytes += blk.bytes;
            return;
        }
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(blk.device);
    (void)hipFree(p);
    (void)hipSetDevice(prev);
}

void pool_trim()
{
    std::map<std::pair<int, size_t>, std::vector<void *>> parked;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        parked.swap(g_parked);
        g_parked_bytes = 0;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (auto &kv : parked) {
        (void)hipSetDevice(kv.first.first);
        for (void *p : kv.second) (void)hipFree(p);
    }
    (void)hipSetDevice(prev);
}

hipError_t pool_stream_acquire(hipStream_t *out)
{
    int device = 0;
    hipError_t err = hipGetDevice(&device);
    if (err != hipSuccess) return err;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto &v = g_streams[device];
        if (!v.empty()) {
            *out = v.back();
            v.pop_back();
            return hipSuccess;
        }
    }
    err = hipStreamCreate(out);
    if (err != hipSuccess) return err;
    std::lock_guard<std::mutex> lock(g_mu);
    g_stream_device[*out] = device;
    return hipSuccess;
}

hipError_t pool_event_acquire(hipEvent_t *out, bool timing)
{
    int device = 0;
    hipError_t err = hipGetDevice(&device);
    if (err != hipSuccess) return err;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto &v = g_events[timing ? 1 : 0][device];
        if (!v.empty()) {
            *out = v.back();
            v.pop_back();
            return hipSuccess;
        }
    }
    return timing ? hipEventCreate(out) : hipEventCreateWithFlags(out, hipEventDisableTiming);
}

void pool_event_release(hipEvent_t event, bool timing)
{
    if (!event) return;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipEventDestroy(event); return; }
    std::lock_guard<std::mutex> lock(g_mu);
    g_events[timing ? 1 : 0][device].push_back(event);
}

hipError_t pool_pinned_acquire(void **out)
{
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!g_pinned.empty()) {
            *out = g_pinned.back();
            g_pinned.pop_back();
            return hipSuccess;
        }
    }
    return hipHostMalloc(out, POOL_PINNED_BYTES);
}

void pool_pinned_release(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_mu);
    g_pinned.push_back(p);
}

void pool_stream_release(hipStream_t stream)
{
    if (!stream) return;
    (void)hipStreamSynchronize(stream);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_stream_device.find(stream);
    if (it == g_stream_device.end()) return;
    g_streams[it->second].push_back(stream);
}

}  // namespace skm
