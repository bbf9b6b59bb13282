// C ABI of libseekmer_hip.so (declared in include/seekmer_hip.h): handle
// management, HBM buffers, stream/event plumbing and the launch sequences.
// No torch types, no exceptions across the boundary.
#include "../../include/seekmer_hip.h"
#include "skm_kernels.h"
#include "skm_pool.h"

#include <dlfcn.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <atomic>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <unordered_map>
#include <mutex>
#include <thread>
#include <numeric>
#include <string>
#include <vector>

using namespace skm;

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(SKM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// SKM_TRACE_STALE=1: report (and clear) a HIP error that some earlier call left behind (tuning aid)
#define STALE_CHECK(where)                                                                           \
    do {                                                                                             \
        static const bool trace_stale_ = getenv("SKM_TRACE_STALE") != nullptr;                       \
        if (trace_stale_) {                                                                          \
            const hipError_t s_ = hipGetLastError();                                                 \
            if (s_ != hipSuccess) fprintf(stderr, "[skm] stale HIP error at %s: %s\n", where, hipGetErrorString(s_)); \
        }                                                                                            \
    } while (0)

#define SKM_TRY(call)          \
    do {                       \
        int rc_ = (call);      \
        if (rc_ != SKM_OK) return rc_; \
    } while (0)

// device buffer that only ever grows
template <class T>
struct DBuf {
    T *p = nullptr;
    size_t cap = 0;      // elements
    int ensure(size_t n, bool keep = false, hipStream_t stream = nullptr)
    {
        if (n <= cap) return SKM_OK;
        size_t want = std::max(n, cap + cap / 2);
        T *q = nullptr;
        HIP_TRY(pool_alloc((void **)&q, want * sizeof(T)));
        want = pool_round(want * sizeof(T)) / sizeof(T);
        if (keep && p && cap) {
            HIP_TRY(hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        pool_free(p);
        p = q;
        cap = want;
        return SKM_OK;
    }
    void release()
    {
        pool_free(p);
        p = nullptr;
        cap = 0;
    }
    size_t bytes() const { return cap * sizeof(T); }
};

// runs a clean-up on every exit of the enclosing scope unless dismissed (the HIP_TRY / SKM_TRY
// macros return from the middle of a function)
template <class F>
struct ScopeGuard {
    F f;
    bool armed = true;
    explicit ScopeGuard(F fn) : f(fn) {}
    ~ScopeGuard() { if (armed) f(); }
    void dismiss() { armed = false; }
};
template <class F> ScopeGuard<F> on_exit(F f) { return ScopeGuard<F>(f); }

int set_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return SKM_OK;
}

}  // namespace

struct skm_index {
    int device = 0;
    DevIndex d{};
    void *kmers = nullptr, *contigs = nullptr, *seq2 = nullptr, *targets = nullptr, *buckets = nullptr;
    void *edge_kmers = nullptr, *signatures = nullptr;
    int64_t n_slots = 0, bytes = 0;
    int64_t layout[8] = {0};          // skm_index_layout
    int cu_count = 256;
    // the caller's handle + one per mapper that maps against it: skm_index_destroy only gives up the
    // caller's, the device copy goes with the last one (garbage collectors finalise a mapper and
    // its index in any order; a mapper destroyed after its index used to read freed memory here)
    std::atomic<int> holders{1};
};

struct skm_mapper {
    skm_index *ix = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::mutex mu;
    // class table
    ClassTable t{};
    DBuf<ClassSlot> slots;
    DBuf<int32_t> arena;
    DBuf<int64_t> class_list;
    DBuf<unsigned long long> counters;   // [0]=arena_cursor [1]=n_classes [2]=n_unaligned [3]=n_units [8..2007]=fld
    DBuf<int> error;
    // batch buffers
    DBuf<uint8_t> bases;
    DBuf<int64_t> offsets;
    DBuf<uint32_t> records;
    DBuf<int32_t> workspace;
    DBuf<int32_t> unit_begin, unit_end, rec_unit;
    DBuf<Coord> unit_anchor;
    DBuf<int64_t> unit_slot;
    DBuf<unsigned long long> rec_tuple;
    unsigned long long *pinned = nullptr;   // host-pinned readback words
    DBuf<uint64_t> rec_key;
    bool keep_spans = false, last_spans = false;   // spans wanted / written by the last batch
    DBuf<int32_t> unit_entries;
    DBuf<unsigned long long> batch_ctl;  // [0]=ids_cursor [8..2007]=fld [2048..2063]=stats
    int grid_blocks = 0;
    int64_t expected_units = 0;       // skm_mapper_expect_units: the sample's size, announced before its reads
    bool packed_sized = false;        // the batch buffers hold a run of PACKED_MAX_UNITS already
    int64_t units_done = 0;
    int64_t first_seen_bound = 0;     // every first_seen in the table is below this
    int64_t last_units = 0, last_ids = 0;
    int64_t host_classes = 0, host_arena_used = 0;
    // the table's unit totals as the last mapped batch read them back (valid: nothing else -- a merge
    // of a foreign table -- has changed the device counters since): spares skm_quant_infer a
    // synchronous read of two words
    bool host_totals_valid = false;
    unsigned long long host_units = 0, host_unaligned = 0;
    int want_stats = 0;               // 0 production, 1 counting build, 2 census build
    double t_pack_ns = 0, t_map_ns = 0, t_class_ns = 0, batches = 0;
    double t_em_ns = 0, em_iters = 0;          // skm_quant_infer calls on this mapper
    unsigned long long stats_total[48] = {0};
    // ---- host batches: staging lanes + one worker (skm_mapper_map_batch[_async])
    // A host batch is copied to HBM on a lane's own stream by the thread that submits it (the
    // copy of batch i+1 runs under the kernels of batch i) and then queued; the worker maps
    // the queued batches in submission order on the mapper's stream, holding `mu` only for its
    // batch.  Every call that reads or changes the table first waits for the queue to drain.
    struct Lane {
        DBuf<uint8_t> bases;
        DBuf<int64_t> offsets;
        hipStream_t stream = nullptr;
        bool busy = false;
    };
    static constexpr int N_LANES = 3;
    Lane lanes[N_LANES];
    struct Job { int lane; int64_t n_units; int paired; int64_t offset_base; int64_t first_unit; uint64_t ticket; };
    std::mutex q_mu;
    std::condition_variable q_cv;             // job queued / stop
    std::condition_variable done_cv;          // job finished, lane released
    std::deque<Job> jobs;
    std::thread worker;
    bool worker_started = false, stop = false;
    uint64_t next_ticket = 1, done_ticket = 0;
    int job_error = SKM_OK;                   // first failure of a queued batch (sticky until reported)
    uint64_t job_error_ticket = 0;
    std::string job_error_msg;
    DBuf<unsigned long long> scan_out;        // device-side max read length / monotonicity of a batch
    // ---- packed pieces (skm_mapper_push_packed): reads that arrive as 2-bit code words wait in HBM,
    // one ascending list of pieces per stream, until the worker maps a run of units that every
    // stream covers.  All of it under q_mu.
    struct Piece {
        int64_t first = 0, n = 0;             // units [first, first + n) of the stream (what is left of the piece)
        int64_t origin = 0;                   // first_read of the piece as pushed: the arrays start there
        int cw = 1;
        int64_t uniform_len = -1;
        std::shared_ptr<char> block;          // one HBM allocation: codes | lengths | exception reads | masks
        uint64_t *codes = nullptr;            // [n as pushed][cw]
        uint32_t *lengths = nullptr;          // or NULL (uniform_len)
        uint32_t *exc_reads_dev = nullptr, *exc_masks = nullptr;
        std::vector<uint32_t> exc_reads;      // host copy (indices relative to origin), for cutting runs
        bool in_job = false;
    };
    std::deque<Piece> pending[2];
    hipStream_t packed_stream = nullptr;
    int packed_paired = -1;                   // -1 until the first piece
    int packed_flush = 0;                     // callers waiting for everything mappable to be mapped
    int packed_waiters = 0;                   // pushers held back by the byte limit: whatever run there is gets mapped
    int64_t packed_max_pending = 8LL << 30;   // bytes of HBM that pieces may hold before a pusher waits
    bool packed_busy = false;
    int64_t packed_dropped = 0;               // reads that never got a mate
    // HBM held by pieces that wait to be mapped; a pusher that is ahead of the GPU by more than
    // PACKED_MAX_PENDING bytes waits (only while the worker has something to map: a caller that
    // pushes one stream long before the other is never blocked on itself)
    std::shared_ptr<std::atomic<int64_t>> packed_bytes = std::make_shared<std::atomic<int64_t>>(0);
    int vote[8] = {1, 1, 1, 1, 1, 1, 1, 0};   // quorum per action (start, lookup, merge, left, right, emit, scan)
    // skm_mapper_device_table: the classes in registry order, as class_compact leaves them
    DBuf<int64_t> view_start, view_len;
    DBuf<double> view_count;
    DBuf<unsigned long long> view_first_seen;
};

struct skm_quant {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipEvent_t chunk_ev[2] = {nullptr, nullptr};
    unsigned long long *pinned = nullptr;     // host-pinned readback of the control block
    std::mutex mu;
    int64_t n_tx = 0, n_classes = 0, n_ids = 0, n_rows = 0;
    DBuf<int64_t> cls_offset, row_start, tx_row;
    DBuf<int32_t> ids, tx_cls, row_tx, perm;      // perm[k] = caller's index of internal class k
    DBuf<double> cls_count, cls_count_saved, inner, row_sum;
    DBuf<double> eff_len, x0, x1, acc, part_max;
    DBuf<unsigned int> part_flags, arrivals;
    DBuf<unsigned long long> ctl, cum;
    DBuf<unsigned int> tile_total;            // scratch of the tiled multinomial draw
    DBuf<double> x_start, boot_out;           // bootstrap: the common start vector, the replicates' results
    // working set of the batched EM (skm_em_batch.hip): eight bootstrap replicates side by side
    struct Batch {
        DBuf<double> cls_count, inner, row_sum, x0, x1, part_max;
        DBuf<unsigned int> part_flags;
        DBuf<unsigned long long> ctl, mgr;
        DBuf<double> counts_all;              // [group][C] pre-drawn class counts of a group of replicates
        DBuf<int64_t> iters;                  // [group] their step counts
    } batch;
    double n_total = 0;
    bool n_total_reduced = false;             // n_total already is the sum over all ranks
    // RCCL communicator (borrowed from an skm_comm), loaded lazily
    void *comm = nullptr;
    int rank = 0, world = 1;
    double t_em_ns = 0, iters_total = 0, launches = 0;
};

struct skm_comm {
    int device = 0;
    void *comm = nullptr;
    int rank = 0, world = 1;
};

// ------------------------------------------------------------------- errors
extern "C" const char *skm_last_error(void) { return g_error.c_str(); }

extern "C" int skm_device_count(int *count)
{
    if (!count) return fail(SKM_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        *count = 0;
        return fail(SKM_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    }
    *count = n;
    return SKM_OK;
}

extern "C" int skm_device_malloc(int device, int64_t bytes, void **out)
{
    if (!out || bytes < 0) return fail(SKM_ERR_ARG, "bad argument");
    SKM_TRY(set_device(device));
    HIP_TRY(hipMalloc(out, (size_t)std::max<int64_t>(bytes, 1)));
    return SKM_OK;
}

extern "C" int skm_device_free(int device, void *ptr)
{
    SKM_TRY(set_device(device));
    if (ptr) HIP_TRY(hipFree(ptr));
    return SKM_OK;
}

extern "C" int skm_device_upload(int device, void *dst, const void *src, int64_t bytes)
{
    if (!dst || !src || bytes < 0) return fail(SKM_ERR_ARG, "bad argument");
    SKM_TRY(set_device(device));
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return SKM_OK;
}

extern "C" int skm_device_download(int device, void *dst, const void *src, int64_t bytes)
{
    if (!dst || !src || bytes < 0) return fail(SKM_ERR_ARG, "bad argument");
    SKM_TRY(set_device(device));
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return SKM_OK;
}

extern "C" int skm_device_synchronize(int device)
{
    SKM_TRY(set_device(device));
    HIP_TRY(hipDeviceSynchronize());
    return SKM_OK;
}

// Page-locked host memory: a batch handed over from it crosses the link at the full PCIe rate
// and asynchronously (no staging copy by the runtime).  Plain C allocator signatures so that
// libseekmer_host.so's FASTQ reader can take them as its slab allocator (skm_fastq_set_allocator).
namespace {

std::atomic<int> g_pinned_device{0};

void *pinned_direct(size_t bytes)
{
    void *p = nullptr;
    const int want = g_pinned_device.load();
    int current = 0;
    if (hipGetDevice(&current) != hipSuccess) return nullptr;
    if (current != want && hipSetDevice(want) != hipSuccess) return nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
    if (current != want) (void)hipSetDevice(current);
    return e == hipSuccess ? p : nullptr;
}

// One page-locked arena per process for the FASTQ readers' pieces.  Page-locking is slow
// (hipHostMalloc: about a millisecond per megabyte, one call at a time inside the driver) and a
// reader asks for a few dozen pieces of a megabyte or two in its first milliseconds -- measured: 19
// concurrent calls that end 15 ms later, the parse waiting behind them.  The arena is page-locked
// ONCE, by a helper thread that skm_pinned_set_device starts (infer.run calls it before it loads the
// index, so the reservation runs under the index upload), and handed out first-fit; a request it
// cannot serve (too large, arena full, no arena) is page-locked on its own as before.
struct PinnedArena {
    std::mutex mu;
    std::condition_variable cv;
    bool started = false, ready = false;
    std::thread helper;
    char *base = nullptr;
    size_t bytes = 0;
    std::map<size_t, size_t> free_at;          // offset -> length of every free range
    std::unordered_map<size_t, size_t> live;   // offset -> length handed out
    ~PinnedArena() { if (helper.joinable()) helper.join(); }
} g_arena;

size_t arena_size()
{
    size_t mb = 128;
    if (const char *v = getenv("SKM_PINNED_ARENA_MB")) mb = (size_t)std::max(0L, atol(v));
    return mb << 20;
}

}  // namespace

extern "C" int skm_pinned_set_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(SKM_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= count) return fail(SKM_ERR_ARG, "no GPU %d", device);
    g_pinned_device.store(device);
    std::lock_guard<std::mutex> hold(g_arena.mu);
    if (!g_arena.started && arena_size() > 0) {
        g_arena.started = true;
        g_arena.helper = std::thread([device]() {
            const size_t want = arena_size();
            void *p = nullptr;
            if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) p = nullptr;
            std::lock_guard<std::mutex> hold(g_arena.mu);
            g_arena.base = (char *)p;
            g_arena.bytes = p ? want : 0;
            if (p) g_arena.free_at[0] = want;
            g_arena.ready = true;
            g_arena.cv.notify_all();
        });
    }
    return SKM_OK;
}

// (called from the FASTQ readers' worker threads, which never chose a device: page-lock against the
// process's GPU, not against GPU 0, and portably, so that any device of the process may copy from it)
extern "C" void *skm_pinned_alloc(size_t bytes)
{
    const size_t need = (std::max<size_t>(bytes, 1) + 4095) & ~(size_t)4095;
    {
        std::unique_lock<std::mutex> hold(g_arena.mu);
        if (g_arena.started && need <= arena_size() / 8) {
            // (a reservation in progress is worth waiting for: a call of our own would queue behind it)
            g_arena.cv.wait(hold, [] { return g_arena.ready; });
            for (auto it = g_arena.free_at.begin(); it != g_arena.free_at.end(); ++it) {
                if (it->second < need) continue;
                const size_t at = it->first, rest = it->second - need;
                g_arena.free_at.erase(it);
                if (rest) g_arena.free_at[at + need] = rest;
                g_arena.live[at] = need;
                return g_arena.base + at;
            }
        }
    }
    return pinned_direct(bytes);
}

extern "C" void skm_pinned_free(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> hold(g_arena.mu);
        if (g_arena.base && (char *)p >= g_arena.base && (char *)p < g_arena.base + g_arena.bytes) {
            const size_t at = (size_t)((char *)p - g_arena.base);
            auto it = g_arena.live.find(at);
            if (it == g_arena.live.end()) return;
            size_t len = it->second, from = at;
            g_arena.live.erase(it);
            auto next = g_arena.free_at.lower_bound(from);          // merge with the free neighbours
            if (next != g_arena.free_at.end() && next->first == from + len) { len += next->second; next = g_arena.free_at.erase(next); }
            if (next != g_arena.free_at.begin()) {
                auto prev = std::prev(next);
                if (prev->first + prev->second == from) { from = prev->first; len += prev->second; g_arena.free_at.erase(prev); }
            }
            g_arena.free_at[from] = len;
            return;
        }
    }
    (void)hipHostFree(p);
}

// Diagnostic: random 16-byte gathers over a zero-filled table of `table_bytes`
// (power of two).  chain=0: independent gathers (throughput); chain=1: each
// address depends on the previous slot (latency under load).
extern "C" int skm_device_gather_ceiling(int device, int64_t table_bytes, int blocks, int per_lane,
                                         int chain, double *gathers_per_second)
{
    if (!gathers_per_second || table_bytes < 4096 || (table_bytes & (table_bytes - 1)) || blocks < 1
            || per_lane < 4)
        return fail(SKM_ERR_ARG, "bad argument");
    SKM_TRY(set_device(device));
    void *table = nullptr;
    unsigned long long *sink = nullptr;
    HIP_TRY(hipMalloc(&table, (size_t)table_bytes));
    HIP_TRY(hipMalloc((void **)&sink, 8));
    HIP_TRY(hipMemset(table, 0, (size_t)table_bytes));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    launch_gather_probe(table, (uint64_t)table_bytes / 16, blocks, per_lane, chain, sink, nullptr);   // warm-up
    HIP_TRY(hipEventRecord(e0, nullptr));
    launch_gather_probe(table, (uint64_t)table_bytes / 16, blocks, per_lane, chain, sink, nullptr);
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *gathers_per_second = (double)blocks * 256.0 * per_lane / (ms * 1e-3);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIP_TRY(hipFree(table));
    HIP_TRY(hipFree(sink));
    return SKM_OK;
}

// -------------------------------------------------------------------- index
extern "C" int skm_index_create(const void *kmers, int64_t n_slots, const void *contigs,
                                int64_t n_contigs, const char *sequences, int64_t n_bases,
                                const void *targets, int64_t n_targets, int device,
                                skm_index **out)
{
    if (!kmers || !contigs || !sequences || !targets || !out)
        return fail(SKM_ERR_ARG, "NULL array");
    if (n_slots <= 0 || (n_slots & (n_slots - 1)) || n_slots > (1LL << 31))
        return fail(SKM_ERR_ARG, "k-mer table size %lld is not a power of two <= 2^31", (long long)n_slots);
    if (n_contigs <= 0 || n_bases < ALIGN_LENGTH || n_targets < 0 || n_bases >= (1LL << 31)
            || n_targets + 32 * n_contigs >= (1LL << 30) || n_contigs >= (1LL << 25))
        return fail(SKM_ERR_ARG, "bad index sizes");
    int n_dev = 0;
    SKM_TRY(skm_device_count(&n_dev));
    if (device < 0 || device >= n_dev) return fail(SKM_ERR_ARG, "device %d out of range", device);
    SKM_TRY(set_device(device));
    // Under the validation and the upload below, a helper loads the library's code objects and
    // parks two streams (hipStreamCreate: > 100 ms apiece, measured) for the handles of the first sample:
    // first-use costs of the GPU side that would otherwise fall into the sample's own run.
    std::thread warm([device]() {
        if (hipSetDevice(device) != hipSuccess) return;
        warm_code_map(); warm_code_classes(); warm_code_em(); warm_code_em_batch(); warm_code_quant_setup();
        static std::once_flag streams_once;
        std::call_once(streams_once, []() {
            hipStream_t a = nullptr, b = nullptr;
            if (pool_stream_acquire(&a) == hipSuccess && pool_stream_acquire(&b) == hipSuccess) {
                pool_stream_release(a);
                pool_stream_release(b);
            }
            void *block[2] = {nullptr, nullptr};           // (and the control-block readbacks of two handles)
            for (auto &p : block) if (pool_pinned_acquire(&p) != hipSuccess) p = nullptr;
            for (auto &p : block) pool_pinned_release(p);
        });
    });
    auto join_warm = on_exit([&]() { warm.join(); });

    // host-side validation: every shape the kernels index with must be in range
    const ContigEntry *hc = (const ContigEntry *)contigs;
    int64_t max_tc = 0;
    bool edge_windows = true;       // first_kmer / last_kmer spell the contig's first / last k bases
    auto encode = [&](int64_t at) {                      // _kmer.pxd:46-68 over the pooled bases
        uint64_t k = 0;
        for (int i = 0; i < K; ++i) {
            const unsigned ch = (unsigned char)sequences[at + i] & 0xDFu;
            k = (k << 2) | (ch == 'T' ? 3u : ch == 'G' ? 2u : ch == 'C' ? 1u : 0u);
        }
        return k;
    };
    for (int64_t c = 0; c < n_contigs; ++c) {
        if (hc[c].offset < 0 || hc[c].length < 0 || hc[c].offset + hc[c].length > n_bases
                || hc[c].target_offset < 0 || hc[c].target_length < 0
                || hc[c].target_offset + hc[c].target_length > n_targets)
            return fail(SKM_ERR_ARG, "contig %lld points outside the pooled arrays", (long long)c);
        max_tc = std::max<int64_t>(max_tc, hc[c].target_length);
        if (edge_windows && (hc[c].length < K || encode(hc[c].offset) != hc[c].first_kmer
                             || encode(hc[c].offset + hc[c].length - K) != hc[c].last_kmer))
            edge_windows = false;
    }
    bool sorted_targets = true;      // (the builder sorts (contig, entry, offset), _index_builder.pyx)
    {
        const Coord *ht = (const Coord *)targets;
        for (int64_t c = 0; c < n_contigs && sorted_targets; ++c) {
            const int64_t first = hc[c].target_offset, end = first + hc[c].target_length;
            for (int64_t t = first + 1; t < end; ++t)
                if (ht[t - 1].entry > ht[t].entry) { sorted_targets = false; break; }
        }
    }
    if (max_tc >= (1LL << 22))
        return fail(SKM_ERR_ARG, "a contig lists %lld targets (limit 4194303)", (long long)max_tc);
    const IndexEntry *hk = (const IndexEntry *)kmers;
    // (2 GiB of slots at 190k transcripts: checked by all host cores)
    const int n_workers = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(
                              std::max(1u, std::thread::hardware_concurrency()), 16u), n_slots >> 20));
    std::vector<int64_t> empty_part((size_t)n_workers, 0), bad_part((size_t)n_workers, -1);
    {
        std::vector<std::thread> workers;
        for (int w = 0; w < n_workers; ++w)
            workers.emplace_back([&, w]() {
                const int64_t first = n_slots * w / n_workers, last = n_slots * (w + 1) / n_workers;
                int64_t empty = 0;
                for (int64_t i = first; i < last; ++i) {
                    if (hk[i].kmer == KMER_INVALID) { ++empty; continue; }
                    const int32_t e = hk[i].pos.entry < 0 ? ~hk[i].pos.entry : hk[i].pos.entry;
                    if (hk[i].pos.offset >= 0
                            && (e < 0 || e >= n_contigs || hk[i].pos.offset + K > hc[e].length)) {
                        bad_part[(size_t)w] = i;
                        break;
                    }
                }
                empty_part[(size_t)w] = empty;
            });
        for (auto &t : workers) t.join();
    }
    int64_t empty = 0;
    for (int w = 0; w < n_workers; ++w) {
        if (bad_part[(size_t)w] >= 0)
            return fail(SKM_ERR_ARG, "k-mer slot %lld points outside its contig", (long long)bad_part[(size_t)w]);
        empty += empty_part[(size_t)w];
    }
    if (empty == 0) return fail(SKM_ERR_ARG, "k-mer table has no empty slot");

    skm_index *ix = new skm_index();
    ix->device = device;
    char *d_ascii = nullptr;
    auto undo = on_exit([&]() {                 // any early return below: nothing is left behind
        (void)hipFree(d_ascii);
        skm_index_destroy(ix);
    });
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    ix->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int64_t n_words = (n_bases + 31) / 32 + 1;
    HIP_TRY(hipMalloc(&ix->kmers, (size_t)n_slots * sizeof(IndexEntry)));
    HIP_TRY(hipMalloc(&ix->seq2, (size_t)n_words * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void **)&d_ascii, (size_t)n_bases));
    HIP_TRY(hipMemcpy(ix->kmers, kmers, (size_t)n_slots * sizeof(IndexEntry), hipMemcpyHostToDevice));
    int64_t n_overflow = 0;
    {   // contig records (skm_device.h: DevContig) and, behind them in the same allocation, the target
        // slices that do not fit a side; of every target only the signed entry is kept
        std::vector<DevContig> rows((size_t)n_contigs);
        std::vector<uint64_t> edges((size_t)n_contigs * 2);
        std::vector<int32_t> overflow;
        const Coord *ht = (const Coord *)targets;
        const int64_t row_words = (int64_t)sizeof(DevContig) / 4;      // (int32 words per record)
        for (int64_t c = 0; c < n_contigs; ++c) {
            DevContig &d = rows[(size_t)c];
            memset(&d, 0, sizeof(d));
            edges[(size_t)(2 * c)] = hc[c].first_kmer;
            edges[(size_t)(2 * c + 1)] = hc[c].last_kmer;
            const int64_t first = hc[c].target_offset, count = hc[c].target_length;
            const int64_t place = n_contigs * row_words + (int64_t)overflow.size();
            for (int s = 0; s < 2; ++s) {
                DevSide &side = d.side[s];
                side.offset = (int32_t)hc[c].offset;
                side.length = (int32_t)hc[c].length;
                // [0] the contig's last 8 bases (low bits of last_kmer), [1] its first 8 (top of first_kmer)
                const uint32_t edge8 = s == 0 ? (uint32_t)(hc[c].last_kmer & 0xffffu)
                                              : (uint32_t)(hc[c].first_kmer >> (2 * K - 16)) & 0xffffu;
                side.count_edge = ((uint32_t)std::min<int64_t>(count, 0xffff) << 16) | edge8;
                if (count <= CONTIG_INLINE_TARGETS) {
                    for (int64_t i = 0; i < count; ++i) side.targets[i] = ht[first + i].entry;
                } else {
                    side.targets[0] = (int32_t)place;
                    side.targets[1] = (int32_t)count;
                }
            }
            if (count > CONTIG_INLINE_TARGETS)
                for (int64_t i = 0; i < count; ++i) overflow.push_back(ht[first + i].entry);
        }
        n_overflow = (int64_t)overflow.size();
        const size_t row_bytes = (size_t)n_contigs * sizeof(DevContig);
        HIP_TRY(hipMalloc(&ix->contigs, row_bytes + (size_t)(n_overflow + 16) * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(ix->contigs, rows.data(), row_bytes, hipMemcpyHostToDevice));
        if (n_overflow)
            HIP_TRY(hipMemcpy((char *)ix->contigs + row_bytes, overflow.data(), (size_t)n_overflow * sizeof(int32_t),
                              hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&ix->edge_kmers, (edges.size() + 2) * sizeof(uint64_t)));
        HIP_TRY(hipMemcpy(ix->edge_kmers, edges.data(), edges.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(d_ascii, sequences, (size_t)n_bases, hipMemcpyHostToDevice));
    launch_pack_sequences(d_ascii, n_bases, (uint64_t *)ix->seq2, n_words, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(d_ascii));
    d_ascii = nullptr;
    ix->n_slots = n_slots;
    ix->d.kmers = (const IndexEntry *)ix->kmers;
    ix->d.slot_mask = (uint32_t)(n_slots - 1);
    ix->d.contigs = (const DevContig *)ix->contigs;
    ix->d.n_contigs = n_contigs;
    ix->d.seq2 = (const uint64_t *)ix->seq2;
    ix->d.n_bases = n_bases;
    ix->d.targets = (const int32_t *)ix->contigs;       // (rows and overflow slices as one int32 array)
    ix->d.edge_kmers = (const uint64_t *)ix->edge_kmers;
    ix->d.n_targets = n_targets;
    ix->d.max_target_count = (int32_t)std::max<int64_t>(max_tc, 1);
    ix->d.edge_windows = edge_windows ? 1 : 0;
    ix->d.sorted_targets = sorted_targets ? 1 : 0;
    ix->bytes = n_slots * (int64_t)sizeof(IndexEntry) + n_contigs * (int64_t)sizeof(DevContig)
                + n_overflow * (int64_t)sizeof(int32_t) + n_words * 8 + n_contigs * 16;
    {   // the same set of k-mers by bucket (skm_device.h: DevBucket): about one k-mer per bucket
        const int64_t occupied = n_slots - empty;
        uint64_t n_buckets = 16;
        while ((int64_t)n_buckets < occupied) n_buckets <<= 1;
        int log2_buckets = 0;
        while ((1ULL << log2_buckets) < n_buckets) ++log2_buckets;
        unsigned long long *d_report = nullptr;
        unsigned long long report[4] = {0, 0, 0, 0};
        const char *off = getenv("SKM_NO_BUCKETS");          // tuning aid: probe the reference's layout
        if (!(off && off[0] == '1') && n_buckets <= (1ULL << 31)) {
            auto undo_report = on_exit([&]() { (void)hipFree(d_report); });
            HIP_TRY(hipMalloc(&ix->buckets, (size_t)n_buckets * sizeof(DevBucket)));
            HIP_TRY(hipMalloc((void **)&d_report, sizeof(report)));
            HIP_TRY(hipMemset(d_report, 0, sizeof(report)));
            launch_bucket_build(ix->d, (uint64_t)n_slots, (DevBucket *)ix->buckets, (uint32_t)(n_buckets - 1),
                                (uint32_t)(32 - log2_buckets), d_report, nullptr);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(report, d_report, sizeof(report), hipMemcpyDeviceToHost));
            ix->layout[1] = (int64_t)n_buckets;
            ix->layout[2] = (int64_t)report[0];
            ix->layout[3] = (int64_t)report[1];
            ix->layout[4] = (int64_t)report[2];
            ix->layout[5] = (int64_t)report[3];
            if ((int64_t)report[0] == occupied && report[2] == 0 && report[3] == 0) {
                ix->d.buckets = (const DevBucket *)ix->buckets;
                ix->d.bucket_mask = (uint32_t)(n_buckets - 1);
                ix->d.bucket_shift = (uint32_t)(32 - log2_buckets);
                ix->bytes += (int64_t)n_buckets * (int64_t)sizeof(DevBucket);
                ix->layout[0] = 1;
                const char *no_sig = getenv("SKM_NO_SIGNATURES");        // tuning aid: the roll asks the buckets straight away
                // signatures of the k-mers by minimizer (skm_device.h: kmer_min_hash): a slot per four k-mers
                int sig_bits = 10;
                while (sig_bits < 30 && (4LL << sig_bits) < occupied) ++sig_bits;
                const size_t sig_bytes = 2 * sizeof(uint64_t) << sig_bits;
                if (!(no_sig && no_sig[0] == '1') && hipMalloc(&ix->signatures, sig_bytes) != hipSuccess) {
                    (void)hipGetLastError();                   // (no room: the roll asks the buckets, as without them)
                    ix->signatures = nullptr;
                }
                if (ix->signatures) {
                    HIP_TRY(hipMemset(ix->signatures, 0, sig_bytes));
                    launch_signature_build((const DevBucket *)ix->buckets, n_buckets, (uint64_t *)ix->signatures,
                                           (uint32_t)(32 - sig_bits), nullptr);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipDeviceSynchronize());
                    if (getenv("SKM_TRACE_SIGNATURES")) {       // tuning aid: how full the signatures are
                        std::vector<uint64_t> head(std::min<size_t>((size_t)1 << 20, sig_bytes / 8));
                        HIP_TRY(hipMemcpy(head.data(), ix->signatures, head.size() * 8, hipMemcpyDeviceToHost));
                        int64_t bits = 0, used = 0, full = 0;
                        for (size_t k = 0; k + 1 < head.size(); k += 2) {
                            const int c = __builtin_popcountll(head[k]) + __builtin_popcountll(head[k + 1]);
                            bits += c; used += c != 0; full += c == 128;
                        }
                        fprintf(stderr, "[skm_index_create] signatures: 2^%d slots of 128 bits; of the first %zu: %.2f bits set per slot, "
                                "%.1f %% in use, %lld full\n", sig_bits, head.size() / 2, 2.0 * (double)bits / (double)head.size(),
                                200.0 * (double)used / (double)head.size(), (long long)full);
                    }
                    ix->d.signatures = (const uint64_t *)ix->signatures;
                    ix->d.signature_shift = (uint32_t)(32 - sig_bits);
                    ix->bytes += (int64_t)sig_bytes;
                    ix->layout[7] = (int64_t)1 << sig_bits;
                }
                const char *no_succ = getenv("SKM_NO_SUCCESSORS");     // tuning aid: every junction k-mer looked up
                if (!(no_succ && no_succ[0] == '1')) {
                    // junction successors of every contig record (skm_device.h: DevContig), by the
                    // lookup the kernels would do.  SKM_TEST_SUCC_LOOKUP=1 (test hook) marks them all
                    // "look it up", so that every hop takes the fall-back a real index hardly ever needs.
                    const char *force = getenv("SKM_TEST_SUCC_LOOKUP");
                    launch_successor_build(ix->d, (DevContig *)ix->contigs, n_contigs, force && force[0] == '1',
                                           nullptr);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipDeviceSynchronize());
                    ix->d.successors = 1;
                    ix->layout[6] = 1;
                }
            } else {                 // not a set the reference's probe reaches everywhere: its layout decides
                HIP_TRY(hipFree(ix->buckets));
                ix->buckets = nullptr;
            }
        }
    }
    undo.dismiss();
    *out = ix;
    return SKM_OK;
}

extern "C" int skm_index_destroy(skm_index *ix)
{
    if (!ix) return SKM_OK;
    if (ix->holders.fetch_sub(1) > 1) return SKM_OK;      // a mapper still maps against it
    (void)hipSetDevice(ix->device);
    (void)hipFree(ix->kmers); (void)hipFree(ix->contigs); (void)hipFree(ix->targets); (void)hipFree(ix->seq2);
    (void)hipFree(ix->buckets); (void)hipFree(ix->edge_kmers); (void)hipFree(ix->signatures);
    delete ix;
    return SKM_OK;
}

extern "C" int skm_index_info(const skm_index *ix, int64_t info[8])
{
    if (!ix || !info) return fail(SKM_ERR_ARG, "NULL argument");
    info[0] = ix->n_slots;
    info[1] = ix->d.n_contigs;
    info[2] = ix->d.n_bases;
    info[3] = ix->d.n_targets;
    info[4] = ix->d.max_target_count;
    info[5] = ix->bytes;
    info[6] = ix->d.edge_windows;
    info[7] = ix->d.sorted_targets;
    return SKM_OK;
}

extern "C" int skm_index_layout(const skm_index *ix, int64_t layout[8])
{
    if (!ix || !layout) return fail(SKM_ERR_ARG, "NULL argument");
    for (int i = 0; i < 8; ++i) layout[i] = ix->layout[i];
    return SKM_OK;
}

// ------------------------------------------------------------------- mapper
namespace {

constexpr int CTR_ARENA = 0, CTR_CLASSES = 1, CTR_UNALIGNED = 2, CTR_UNITS = 3, CTR_LISTED = 4,
              CTR_DEFERRED = 5, CTR_COMMITTED = 6, CTR_FLD = 8;
constexpr int CTR_WORDS = 8 + MAX_FRAGMENT_LENGTH;
constexpr int BC_IDS = 0, BC_FLD = 8, BC_STATS = 2048, BC_WORDS = 2096;

void bind_table(skm_mapper *m, uint64_t n_slots)
{
    m->t.slots = m->slots.p;
    m->t.slot_mask = n_slots - 1;
    m->t.arena = m->arena.p;
    m->t.arena_capacity = (int64_t)m->arena.cap;
    m->t.arena_cursor = m->counters.p + CTR_ARENA;
    m->t.n_classes = m->counters.p + CTR_CLASSES;
    m->t.n_unaligned = m->counters.p + CTR_UNALIGNED;
    m->t.n_units = m->counters.p + CTR_UNITS;
    m->t.global_fld = m->counters.p + CTR_FLD;
    m->t.class_list = m->class_list.p;
    m->t.class_list_capacity = (int64_t)m->class_list.cap;
    m->t.n_listed = m->counters.p + CTR_LISTED;
    m->t.n_deferred = m->counters.p + CTR_DEFERRED;
    m->t.arena_committed = m->counters.p + CTR_COMMITTED;
    m->t.error = m->error.p;
}

int table_reset(skm_mapper *m, uint64_t n_slots)
{
    SKM_TRY(m->slots.ensure(n_slots));
    SKM_TRY(m->arena.ensure(1 << 20));
    SKM_TRY(m->class_list.ensure(1 << 16));
    SKM_TRY(m->counters.ensure(CTR_WORDS));
    SKM_TRY(m->error.ensure(1));
    HIP_TRY(hipMemsetAsync(m->counters.p, 0, CTR_WORDS * sizeof(unsigned long long), m->stream));
    HIP_TRY(hipMemsetAsync(m->error.p, 0, sizeof(int), m->stream));
    bind_table(m, n_slots);
    launch_class_init(m->t, m->stream);
    HIP_TRY(hipGetLastError());
    m->host_classes = 0;
    m->host_arena_used = 0;
    m->host_units = m->host_unaligned = 0;
    m->host_totals_valid = true;
    m->units_done = 0;
    m->first_seen_bound = 0;
    return SKM_OK;
}

// make the class table at least `want_slots` big (power of two), moving its
// entries; the registry (and, for a batch in flight, the unit -> slot map) is
// redirected through a forwarding array
int table_grow(skm_mapper *m, uint64_t want_slots, int64_t units_in_flight)
{
    uint64_t n_slots = m->t.slot_mask + 1;
    uint64_t need = n_slots;
    while (need < want_slots) need <<= 1;
    if (need == n_slots) return SKM_OK;
    DBuf<ClassSlot> new_slots;
    DBuf<int64_t> forward;
    auto undo = on_exit([&]() { new_slots.release(); forward.release(); });
    SKM_TRY(new_slots.ensure(need));
    SKM_TRY(forward.ensure(n_slots));
    ClassTable to = m->t;
    to.slots = new_slots.p;
    to.slot_mask = need - 1;
    launch_class_init(to, m->stream);
    launch_class_rehash(m->t, to, forward.p, m->stream);
    unsigned long long listed = 0;
    HIP_TRY(hipMemcpyAsync(&listed, m->counters.p + CTR_LISTED, 8, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    launch_slot_remap(m->class_list.p, (int64_t)listed, forward.p, m->stream);
    launch_slot_remap(m->unit_slot.p, units_in_flight, forward.p, m->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->slots.release();
    forward.release();
    m->slots = new_slots;
    undo.dismiss();
    bind_table(m, need);
    return SKM_OK;
}

// What skm_mapper_expect_units announced (0: nothing): the table and the batch buffers are sized for
// it once instead of growing step by step under the first runs of a sample.
int table_reserve_for_sample(skm_mapper *m, int64_t sample_units)
{
    // (a hint, possibly a wild one: what it reserves is bounded -- 2^26 classes, 2 GB of table -- and
    // a sample that needs more grows the table as before)
    const double expect = (double)m->host_classes + (double)std::min<int64_t>(sample_units, 1LL << 29) / 8.0 + 1024.0;
    uint64_t want = 1 << 16;
    while ((double)want * 0.5 < expect && want < (1ULL << 26)) want <<= 1;
    SKM_TRY(table_grow(m, want, 0));
    SKM_TRY(m->class_list.ensure((size_t)(expect + 1024.0), true, m->stream));
    SKM_TRY(m->arena.ensure((size_t)(m->host_arena_used + (int64_t)(expect * 6.0) + 1024), true, m->stream));
    bind_table(m, m->t.slot_mask + 1);
    return SKM_OK;
}

// Size the table for a batch of `n_units`: classes are far fewer than units in
// practice (0.09 per pair at 10 M pairs), so reserve for one new class per 8
// units at load <= 0.5 and let the bounded probe defer the rest.
int table_reserve(skm_mapper *m, int64_t n_units)
{
    const double expect = (double)m->host_classes + (double)n_units / 8.0 + 1024.0;
    uint64_t want = 1 << 16;
    while ((double)want * 0.5 < expect) want <<= 1;
    SKM_TRY(table_grow(m, want, 0));
    // registry and arena can never need more than one entry per unit / id of the batch
    SKM_TRY(m->class_list.ensure((size_t)(m->host_classes + n_units + 1024), true, m->stream));
    bind_table(m, m->t.slot_mask + 1);
    return SKM_OK;
}

int read_error(skm_mapper *m)
{
    int err = 0;
    HIP_TRY(hipMemcpyAsync(&err, m->error.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (err == SKM_ERR_COLLISION)
        return fail(SKM_ERR_COLLISION, "two different class tuples share a 64-bit key");
    if (err) return fail(err, "class table kernel reported error %d", err);
    return SKM_OK;
}

// first_unit: global index of the batch's first unit (first-seen values count from it), or -1
// to continue after the units mapped so far
// fill_records: what makes the read records of the batch (records, words per read, u32 words per
// record); nullptr = pack_reads_kernel over ASCII bases + offsets
typedef std::function<int(uint32_t *, int, int)> RecordStage;
int map_batch_resident(skm_mapper *m, const uint8_t *d_bases, const int64_t *d_offsets,
                       int64_t n_units, int paired, int max_len, int64_t first_unit = -1,
                       const RecordStage *fill_records = nullptr)
{
    const int64_t unit_base = first_unit >= 0 ? first_unit : m->units_done;
    skm_index *ix = m->ix;
    const int64_t n_reads = paired ? 2 * n_units : n_units;
    const int words = (max_len + 31) / 32 + 1;
    m->last_units = n_units;
    m->last_ids = 0;
    if (n_units == 0) return SKM_OK;

    const int record_words = ((3 * words + 1 + 15) / 16) * 16;      // 64-byte records
    SKM_TRY(m->records.ensure((size_t)n_reads * record_words + 16));
    if (n_units >= (1LL << 31)) return fail(SKM_ERR_ARG, "more than 2^31 - 1 units in one batch");
    if (m->keep_spans) {
        SKM_TRY(m->unit_begin.ensure(n_units));
        SKM_TRY(m->unit_end.ensure(n_units));
        SKM_TRY(m->unit_anchor.ensure(n_units));
    }
    m->last_spans = m->keep_spans;
    SKM_TRY(m->rec_unit.ensure(n_units));
    SKM_TRY(m->rec_tuple.ensure(n_units));
    SKM_TRY(m->unit_slot.ensure(n_units));
    SKM_TRY(m->rec_key.ensure(n_units));

    SKM_TRY(m->batch_ctl.ensure(BC_WORDS));

    // launch geometry: persistent blocks of 256 lanes, each with MAP_CONTEXTS unit contexts in LDS
    // (just under 40 KB -> 4 blocks per CU); fewer blocks when the per-context list workspace
    // (2 lists of max_target_count entries) would not fit the budget
    constexpr int64_t CONTEXTS = MAP_CONTEXTS;
    int64_t blocks = std::min<int64_t>((n_units + CONTEXTS - 1) / CONTEXTS, (int64_t)ix->cu_count * MAP_BLOCKS_PER_CU);
    // per context: mask extension words (live + staging, two mates) for slices > 64 targets
    const int64_t ext_words = std::max<int64_t>(0, (ix->d.max_target_count + 63) / 64 - 1);
    SKM_TRY(m->workspace.ensure((size_t)(blocks * CONTEXTS * 4 * ext_words * 2 + 16)));
    m->grid_blocks = (int)blocks;
    {   // the kernel addresses a block's records with 32-bit byte offsets
        const int64_t per_block = (n_units + blocks - 1) / blocks;
        if (per_block * (paired ? 2 : 1) * (int64_t)record_words * 4 >= (1LL << 32))
            return fail(SKM_ERR_ARG, "batch of %lld units with %d-base reads is too large for one launch",
                        (long long)n_units, max_len);
    }
    // entry arena: ~8 ids per unit plus one 2048-id slice of slack per wave
    SKM_TRY(m->unit_entries.ensure((size_t)n_units * 8 + (size_t)blocks * (MAP_THREADS / 64) * 2048 + 4096));

    MapBatch b{};
    b.records = m->records.p;
    b.n_units = n_units;
    b.words_per_read = words;
    b.record_words = record_words;
    b.paired = paired;
    b.workspace = m->workspace.p;
    b.unit_begin = m->unit_begin.p;
    b.unit_end = m->unit_end.p;
    b.unit_anchor = m->unit_anchor.p;
    b.keep_spans = m->keep_spans ? 1 : 0;
    b.rec_unit = m->rec_unit.p;
    b.rec_tuple = m->rec_tuple.p;
    b.rec_key = m->rec_key.p;
    b.ids_cursor = m->batch_ctl.p + BC_IDS;
    b.fld = m->batch_ctl.p + BC_FLD;
    b.stats = m->batch_ctl.p + BC_STATS;
    for (int i = 0; i < 8; ++i) b.vote[i] = m->vote[i];

    HIP_TRY(hipEventRecord(m->ev[0], m->stream));
    if (fill_records) SKM_TRY((*fill_records)(m->records.p, words, record_words));
    else launch_pack_reads(d_bases, d_offsets, n_reads, words, record_words, m->records.p, m->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m->ev[1], m->stream));
    unsigned long long ids = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        b.unit_entries = m->unit_entries.p;
        b.ids_capacity = (int64_t)m->unit_entries.cap;
        HIP_TRY(hipMemsetAsync(m->batch_ctl.p, 0, BC_WORDS * sizeof(unsigned long long), m->stream));
        launch_map_units(ix->d, b, m->grid_blocks, m->want_stats, m->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(m->ev[2], m->stream));
        unsigned long long *cursor_and_flag = m->pinned + 32;
        HIP_TRY(hipMemcpyAsync(cursor_and_flag, b.ids_cursor, 16, hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
        ids = cursor_and_flag[0];
        if (cursor_and_flag[1]) return fail(SKM_ERR_STATE, "map kernel: the in-kernel scheduler stalled");
        if ((int64_t)ids <= b.ids_capacity) break;
        if (attempt == 2) return fail(SKM_ERR_STATE, "entry arena overflow");
        SKM_TRY(m->unit_entries.ensure((size_t)ids + 1024));
    }
    m->last_ids = (int64_t)ids;
    if (m->want_stats) {
        unsigned long long st[48];
        HIP_TRY(hipMemcpy(st, b.stats, sizeof(st), hipMemcpyDeviceToHost));
        for (int i = 0; i < 48; ++i) m->stats_total[i] += st[i];
    }

    // class counting
    SKM_TRY(table_reserve(m, n_units));
    SKM_TRY(m->arena.ensure((size_t)(m->host_arena_used + (int64_t)ids + 1024), true, m->stream));
    bind_table(m, m->t.slot_mask + 1);
    // A large batch on an empty table goes in two waves of records: once the classes of the first
    // quarter are committed, most records of the rest land on a committed class and are verified
    // inside class_insert (the slot's tuple word came with the probe); class_verify's second random
    // pass over the table is left with the first wave and the records of classes new in the second.
    const bool two_waves = m->host_classes == 0 && n_units >= (1 << 21);
    for (int pass = 0;; ++pass) {
        // insert (with the commit of the new classes) -> totals -> verify: one pipeline, one
        // synchronisation; the optimistic case needs a single pass
        HIP_TRY(hipMemsetAsync(m->counters.p + CTR_DEFERRED, 0, 8, m->stream));
        // (wave borders in sixteenths of the batch; SKM_CLASS_WAVES="2,8" etc. is a tuning aid)
        int cuts[8] = {4, 16, 16, 16, 16, 16, 16, 16};
        int n_waves = two_waves && pass == 0 ? 2 : 1;
#ifdef SKM_TUNING                                    // (tuning builds only: scripts/build_variant.sh)
        if (n_waves > 1)
            if (const char *e = getenv("SKM_CLASS_WAVES")) {
                int parsed[8], n = 0;
                bool ok = true;
                for (const char *c = e; *c && n < 7;) {
                    parsed[n] = atoi(c);
                    ok = ok && parsed[n] >= 1 && parsed[n] <= 16 && (n == 0 || parsed[n] > parsed[n - 1]);
                    ++n;
                    while (*c && *c != ',') ++c;
                    if (*c == ',') ++c;
                }
                if (ok && n > 0) {                 // strictly ascending sixteenths, else ignored
                    if (parsed[n - 1] < 16) parsed[n++] = 16;
                    for (int i = 0; i < n; ++i) cuts[i] = parsed[i];
                    n_waves = n;
                }
            }
#endif
        for (int wave = 0; wave < n_waves; ++wave) {
            const int64_t w0 = n_waves == 1 || wave == 0 ? 0 : n_units * cuts[wave - 1] / 16;
            const int64_t w1 = n_waves == 1 ? n_units : n_units * cuts[wave] / 16;
            MapBatch part = b;                   // records [w0, w1) of the batch
            part.rec_unit += w0; part.rec_key += w0; part.rec_tuple += w0;
            part.n_units = w1 - w0;
            launch_class_insert(m->t, part, unit_base, m->unit_slot.p + w0, pass > 0, pass == 0 && wave == 0, m->stream);
            launch_class_verify(m->t, part, m->unit_slot.p + w0, m->stream);
        }
        HIP_TRY(hipGetLastError());
        if (pass == 0) HIP_TRY(hipEventRecord(m->ev[3], m->stream));
        HIP_TRY(hipMemcpyAsync(m->pinned, m->counters.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipMemcpyAsync(m->pinned + 16, m->error.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
        const int err = *reinterpret_cast<int *>(m->pinned + 16);
        if (err == SKM_ERR_COLLISION)
            return fail(SKM_ERR_COLLISION, "two different class tuples share a 64-bit key");
        if (err == SKM_ERR_ARG) return fail(SKM_ERR_ARG, "a packed read is longer than its code words hold");
        if (err) return fail(err, "class table kernel reported error %d", err);
        m->host_arena_used = (int64_t)m->pinned[CTR_ARENA];
        m->host_classes = (int64_t)m->pinned[CTR_CLASSES];
        m->host_units = m->pinned[CTR_UNITS];
        m->host_unaligned = m->pinned[CTR_UNALIGNED];
        m->host_totals_valid = true;
        if (m->pinned[CTR_LISTED] != m->pinned[CTR_CLASSES])
            return fail(SKM_ERR_STATE, "class registry out of step (%llu listed, %llu classes)",
                        m->pinned[CTR_LISTED], m->pinned[CTR_CLASSES]);
        if (m->pinned[CTR_DEFERRED] == 0) break;
        if (pass > 40) return fail(SKM_ERR_STATE, "class table cannot absorb the batch");
        // too full for bounded probing: grow 4x (its classes move along), retry the deferred units
        SKM_TRY(table_grow(m, (m->t.slot_mask + 1) * 4, n_units));
    }
    // keep the load below 0.5 for the next batch
    if ((uint64_t)m->host_classes * 2 > m->t.slot_mask + 1)
        SKM_TRY(table_grow(m, (m->t.slot_mask + 1) * 2, 0));
    m->units_done += n_units;
    m->first_seen_bound = std::max(m->first_seen_bound, std::max(m->units_done, unit_base + n_units));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, m->ev[0], m->ev[1])); m->t_pack_ns += ms * 1e6;
    HIP_TRY(hipEventElapsedTime(&ms, m->ev[1], m->ev[2])); m->t_map_ns += ms * 1e6;
    HIP_TRY(hipEventElapsedTime(&ms, m->ev[2], m->ev[3])); m->t_class_ns += ms * 1e6;
    m->batches += 1;
    return SKM_OK;
}

}  // namespace

extern "C" int skm_mapper_create(skm_index *ix, skm_mapper **out)
{
    if (!ix || !out) return fail(SKM_ERR_ARG, "NULL argument");
    SKM_TRY(set_device(ix->device));
    skm_mapper *m = new skm_mapper();
    m->ix = ix;
    ix->holders.fetch_add(1);
    HIP_TRY(pool_stream_acquire(&m->stream));
    for (auto &e : m->ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipHostMalloc((void **)&m->pinned, 64 * sizeof(unsigned long long)));
    if (const char *v = getenv("SKM_MAP_STATS")) m->want_stats = v[0] == '2' ? 2 : 1;
    if (const char *v = getenv("SKM_TEST_PACKED_MAX_PENDING"))      // test hook: the pushers' byte limit
        if (atoll(v) > 0) m->packed_max_pending = atoll(v);
    if (const char *v = getenv("SKM_MAP_VOTE"))          // tuning aid: "start,lookup,merge,left,right,emit,scan"
        sscanf(v, "%d,%d,%d,%d,%d,%d,%d", &m->vote[0], &m->vote[1], &m->vote[2], &m->vote[3], &m->vote[4],
               &m->vote[5], &m->vote[6]);
    HIP_TRY(hipStreamCreateWithFlags(&m->packed_stream, hipStreamNonBlocking));   // (10 ms: not at the first piece's push)
    int rc = table_reset(m, 1 << 16);
    if (rc != SKM_OK) { delete m; ix->holders.fetch_sub(1); return rc; }
    HIP_TRY(hipStreamSynchronize(m->stream));
    *out = m;
    return SKM_OK;
}

extern "C" int skm_mapper_destroy(skm_mapper *m)
{
    if (!m) return SKM_OK;
    (void)hipSetDevice(m->ix->device);
    if (m->worker_started) {
        { std::lock_guard<std::mutex> hold(m->q_mu); m->stop = true; }
        m->q_cv.notify_all();
        m->worker.join();                     // (finishes what is queued)
    }
    (void)hipStreamSynchronize(m->stream);
    for (auto &lane : m->lanes) {
        if (lane.stream) { (void)hipStreamSynchronize(lane.stream); (void)hipStreamDestroy(lane.stream); }
        lane.bases.release();
        lane.offsets.release();
    }
    for (auto &list : m->pending) list.clear();
    if (m->packed_stream) { (void)hipStreamSynchronize(m->packed_stream); (void)hipStreamDestroy(m->packed_stream); }
    m->scan_out.release();
    m->view_start.release(); m->view_len.release(); m->view_count.release(); m->view_first_seen.release();
    m->slots.release(); m->arena.release(); m->class_list.release();
    m->counters.release();
    m->error.release(); m->bases.release(); m->offsets.release(); m->records.release();
    m->workspace.release(); m->unit_begin.release(); m->unit_end.release();
    m->rec_unit.release(); m->unit_anchor.release(); m->rec_tuple.release();
    m->unit_slot.release(); m->rec_key.release(); m->unit_entries.release(); m->batch_ctl.release();
    for (auto &e : m->ev) (void)hipEventDestroy(e);
    if (m->pinned) (void)hipHostFree(m->pinned);
    pool_stream_release(m->stream);
    skm_index *const ix = m->ix;
    delete m;
    return skm_index_destroy(ix);                          // (the mapper's hold on the index)
}

namespace {

// ---- host batches: staging lanes and the mapping worker --------------------------------
int run_job(skm_mapper *m, const skm_mapper::Job &job)
{
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    skm_mapper::Lane &lane = m->lanes[job.lane];
    const int64_t n_reads = job.paired ? 2 * job.n_units : job.n_units;
    // longest read and monotonicity from the offsets in HBM (the lane's copy is complete)
    SKM_TRY(m->scan_out.ensure(2));
    HIP_TRY(hipMemsetAsync(m->scan_out.p, 0, 16, m->stream));
    launch_offsets_scan(lane.offsets.p, n_reads, job.offset_base, m->scan_out.p, m->stream);   // (and rebases them to 0)
    HIP_TRY(hipGetLastError());
    unsigned long long *scan = m->pinned + 40;
    HIP_TRY(hipMemcpyAsync(scan, m->scan_out.p, 16, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (scan[1]) return fail(SKM_ERR_ARG, "offsets are not monotone (%llu descents)", scan[1]);
    if (scan[0] > (1ULL << 20)) return fail(SKM_ERR_ARG, "read longer than 2^20 bases");
    return map_batch_resident(m, lane.bases.p, lane.offsets.p, job.n_units, job.paired, (int)scan[0],
                              job.first_unit);
}

// ---- packed pieces ----------------------------------------------------------------------
constexpr int64_t PACKED_MIN_UNITS = 1 << 15;      // smaller runs wait for more (or for a flush)
constexpr int64_t PACKED_MAX_UNITS = 1 << 21;      // one launch

// (q_mu held) the first run of units that every stream covers: [*lo, *hi), at most PACKED_MAX_UNITS
bool packed_find_run(skm_mapper *m, int64_t *lo, int64_t *hi)
{
    const auto &a = m->pending[0];
    if (a.empty()) return false;
    if (m->packed_paired != 1) {
        *lo = a.front().first;
        *hi = *lo;
        for (const auto &piece : a) {
            if (piece.first != *hi) break;
            *hi += piece.n;
            if (*hi - *lo >= PACKED_MAX_UNITS) { *hi = *lo + PACKED_MAX_UNITS; break; }
        }
        return *hi > *lo;
    }
    const auto &b = m->pending[1];
    size_t i = 0, j = 0;
    bool open = false;
    while (i < a.size() && j < b.size()) {
        const int64_t from = std::max(a[i].first, b[j].first);
        const int64_t to = std::min(a[i].first + a[i].n, b[j].first + b[j].n);
        if (from < to) {
            if (!open) { *lo = from; *hi = to; open = true; }
            else if (from == *hi) *hi = to;
            else break;
            if (*hi - *lo >= PACKED_MAX_UNITS) { *hi = *lo + PACKED_MAX_UNITS; break; }
        }
        if (a[i].first + a[i].n <= b[j].first + b[j].n) ++i; else ++j;
    }
    return open;
}

// (q_mu held) forget units [lo, hi) of a stream: they have been mapped.  What a piece holds
// before `lo` (reads whose mates have not arrived yet) and after `hi` stays, sharing the block.
void packed_consume(skm_mapper *m, int stream, int64_t lo, int64_t hi)
{
    auto &list = m->pending[stream];
    std::deque<skm_mapper::Piece> kept;
    for (auto &piece : list) {
        piece.in_job = false;
        const int64_t end = piece.first + piece.n;
        if (end <= lo || piece.first >= hi) { kept.push_back(std::move(piece)); continue; }
        if (piece.first < lo) {
            skm_mapper::Piece head = piece;
            head.n = lo - piece.first;
            kept.push_back(std::move(head));
        }
        if (end > hi) {
            skm_mapper::Piece tail = piece;
            tail.first = hi;
            tail.n = end - hi;
            kept.push_back(std::move(tail));
        }
    }
    list.swap(kept);
}

// (q_mu held) nothing more can be mapped: reads without a mate go
void packed_drop_all(skm_mapper *m)
{
    for (auto &list : m->pending) {
        for (auto &piece : list) m->packed_dropped += piece.n;
        list.clear();
    }
}

struct PackedSegment {               // part of one piece inside a run
    int mate;
    int64_t first, n;                // units
    const uint64_t *codes;
    const uint32_t *lengths;
    int cw;
    uint32_t uniform_len;
    const uint32_t *exc_reads, *exc_masks;   // device, already offset to the segment's first exception
    int64_t n_exc;
    int64_t exc_base;                // index (relative to the piece as pushed) of the segment's first read
};

int run_packed_job(skm_mapper *m, int64_t lo, int64_t hi, int paired, const std::vector<PackedSegment> &segments,
                   int max_cw)
{
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    const int mates = paired ? 2 : 1;
    if (!m->packed_sized && m->expected_units > 0) {
        // Runs grow with what has arrived (a few ten thousand units, then hundreds of thousands, then
        // PACKED_MAX_UNITS): sized run by run, every batch buffer is allocated three or four times in
        // a sample's first milliseconds.  Size them for the largest run of the announced sample once.
        const int64_t most = std::max<int64_t>(hi - lo, std::min(m->expected_units, PACKED_MAX_UNITS));
        const int words = max_cw + 1, record_words = ((3 * words + 1 + 15) / 16) * 16;
        SKM_TRY(m->records.ensure((size_t)most * mates * record_words + 16));
        SKM_TRY(m->rec_unit.ensure(most)); SKM_TRY(m->rec_tuple.ensure(most));
        SKM_TRY(m->unit_slot.ensure(most)); SKM_TRY(m->rec_key.ensure(most));
        SKM_TRY(m->unit_entries.ensure((size_t)most * 8 + (size_t)m->ix->cu_count * MAP_BLOCKS_PER_CU * (MAP_THREADS / 64) * 2048 + 4096));
        if (m->host_classes == 0) SKM_TRY(table_reserve_for_sample(m, m->expected_units));
        m->packed_sized = true;
    }
    const RecordStage fill = [&](uint32_t *records, int words, int record_words) -> int {
        for (const PackedSegment &seg : segments) {
            uint32_t *dst = records + ((seg.first - lo) * mates + seg.mate) * (int64_t)record_words;
            launch_unpack_reads(seg.codes, seg.cw, seg.cw, seg.lengths, seg.uniform_len, seg.n, words, dst,
                                (int64_t)mates * record_words, m->error.p, m->stream);
            if (seg.n_exc)
                launch_unpack_exceptions(seg.exc_reads, seg.exc_masks, seg.n_exc, seg.cw, seg.exc_base, words, dst,
                                         (int64_t)mates * record_words, m->stream);
        }
        HIP_TRY(hipGetLastError());
        return SKM_OK;
    };
    return map_batch_resident(m, nullptr, nullptr, hi - lo, paired, max_cw * 32, lo, &fill);
}

void worker_main(skm_mapper *m)
{
    for (;;) {
        skm_mapper::Job job;
        bool packed = false;
        int64_t lo = 0, hi = 0;
        int paired = 0, max_cw = 1;
        std::vector<PackedSegment> segments;
        {
            std::unique_lock<std::mutex> hold(m->q_mu);
            for (;;) {
                if (!m->jobs.empty()) break;
                if (m->job_error != SKM_OK && (!m->pending[0].empty() || !m->pending[1].empty())) {
                    packed_drop_all(m);         // after a failure the table is not to be trusted
                    m->done_cv.notify_all();
                }
                // (a pusher that waits at the byte limit may be the only one who could make the run
                // longer: its presence maps whatever run there is, as a flush does)
                if (packed_find_run(m, &lo, &hi)
                        && (hi - lo >= PACKED_MIN_UNITS || m->packed_flush || m->packed_waiters || m->stop)) {
                    packed = true;
                    break;
                }
                if (m->stop) {                  // stop requested and nothing left to map
                    packed_drop_all(m);
                    return;
                }
                m->q_cv.wait(hold);
            }
            if (!packed) {
                job = m->jobs.front();
                m->jobs.pop_front();
            } else {
                paired = m->packed_paired == 1;
                for (int s = 0; s < (paired ? 2 : 1); ++s)
                    for (auto &piece : m->pending[s]) {
                        const int64_t from = std::max(lo, piece.first), to = std::min(hi, piece.first + piece.n);
                        if (from >= to) continue;
                        piece.in_job = true;
                        PackedSegment seg{};
                        seg.mate = s;
                        seg.first = from;
                        seg.n = to - from;
                        const int64_t skip = from - piece.origin;
                        seg.codes = piece.codes + skip * piece.cw;
                        seg.lengths = piece.lengths ? piece.lengths + skip : nullptr;
                        seg.cw = piece.cw;
                        seg.uniform_len = (uint32_t)std::max<int64_t>(piece.uniform_len, 0);
                        const auto e0 = std::lower_bound(piece.exc_reads.begin(), piece.exc_reads.end(), (uint32_t)skip);
                        const auto e1 = std::lower_bound(piece.exc_reads.begin(), piece.exc_reads.end(), (uint32_t)(skip + seg.n));
                        seg.n_exc = e1 - e0;
                        const int64_t at = e0 - piece.exc_reads.begin();
                        seg.exc_reads = piece.exc_reads_dev + at;
                        seg.exc_masks = piece.exc_masks + at * piece.cw;
                        seg.exc_base = skip;
                        max_cw = std::max(max_cw, piece.cw);
                        segments.push_back(seg);
                    }
                m->packed_busy = true;
            }
        }
        int rc = SKM_OK;
        std::string message;
        if (packed) {
            rc = run_packed_job(m, lo, hi, paired, segments, max_cw);
            if (rc != SKM_OK) {
                message = g_error;
                // kernels that read the pieces' blocks may still be queued: nothing is handed back
                // to the pool (and from there to the next pusher's copy) before the stream has drained
                (void)hipStreamSynchronize(m->stream);
            }
            {
                std::lock_guard<std::mutex> hold(m->q_mu);
                if (rc != SKM_OK && m->job_error == SKM_OK) {
                    m->job_error = rc;
                    m->job_error_ticket = 0;              // (reported by every wait)
                    m->job_error_msg = message;
                }
                for (int s = 0; s < (paired ? 2 : 1); ++s) packed_consume(m, s, lo, hi);
                m->packed_busy = false;
            }
            m->done_cv.notify_all();
            continue;
        }
        bool skip;
        {
            std::lock_guard<std::mutex> hold(m->q_mu);
            skip = m->job_error != SKM_OK;        // after a failure the table is not to be trusted
        }
        if (!skip) {
            rc = run_job(m, job);
            if (rc != SKM_OK) message = g_error;  // (this thread's message)
        }
        {
            std::lock_guard<std::mutex> hold(m->q_mu);
            if (rc != SKM_OK && m->job_error == SKM_OK) {
                m->job_error = rc;
                m->job_error_ticket = job.ticket;
                m->job_error_msg = message;
            }
            m->lanes[job.lane].busy = false;
            m->done_ticket = job.ticket;
        }
        m->done_cv.notify_all();
    }
}

// wait until every queued batch up to `ticket` has been mapped (0 = all of them AND every packed
// read whose mate has arrived; reads still without one keep waiting in HBM -- another thread may be
// about to push their mates -- until skm_mapper_reset / _clear / _destroy); reports (and, with
// `consume`, forgets) the first failure among them
int wait_jobs(skm_mapper *m, uint64_t ticket, bool consume)
{
    std::unique_lock<std::mutex> hold(m->q_mu);
    const uint64_t upto = ticket ? ticket : m->next_ticket - 1;
    m->done_cv.wait(hold, [&] { return m->done_ticket >= upto; });
    if (ticket == 0 && (m->packed_busy || !m->pending[0].empty() || !m->pending[1].empty())) {
        m->packed_flush++;
        m->q_cv.notify_all();
        m->done_cv.wait(hold, [&] {
            int64_t lo, hi;
            return !m->packed_busy && (m->job_error != SKM_OK || !packed_find_run(m, &lo, &hi));
        });
        m->packed_flush--;
    }
    if (m->job_error != SKM_OK && m->job_error_ticket <= upto) {
        const int rc = m->job_error;
        const std::string message = m->job_error_msg;
        if (consume) { m->job_error = SKM_OK; m->job_error_ticket = 0; m->job_error_msg.clear(); }
        g_error = message;
        return rc;
    }
    return SKM_OK;
}

// offsets == nullptr: every read is `uniform_len` bases long (the offsets are then made on the device)
int submit_batch(skm_mapper *m, const char *bases, const int64_t *offsets, int64_t uniform_len, int64_t n_units,
                 int paired, int64_t first_unit, uint64_t *ticket_out)
{
    if (!m || n_units < 0 || (!offsets && uniform_len < 0)) return fail(SKM_ERR_ARG, "bad argument");
    if (n_units > 0 && !bases) return fail(SKM_ERR_ARG, "bases is NULL");
    if (n_units >= (1LL << 31)) return fail(SKM_ERR_ARG, "more than 2^31 - 1 units in one batch");
    if (!offsets && uniform_len > (1 << 20)) return fail(SKM_ERR_ARG, "read longer than 2^20 bases");
    SKM_TRY(set_device(m->ix->device));
    const int64_t n_reads = paired ? 2 * n_units : n_units;
    const int64_t first_byte = offsets ? offsets[0] : 0;
    const int64_t n_bytes = offsets ? offsets[n_reads] - offsets[0] : n_reads * uniform_len;
    if (n_bytes < 0) return fail(SKM_ERR_ARG, "offsets are not monotone");
    // a free lane (at most N_LANES batches are in HBM at a time: one being mapped, others copied)
    int li = -1;
    {
        std::unique_lock<std::mutex> hold(m->q_mu);
        if (!m->worker_started) {
            m->worker = std::thread(worker_main, m);
            m->worker_started = true;
        }
        m->done_cv.wait(hold, [&] {
            for (int i = 0; i < skm_mapper::N_LANES; ++i) if (!m->lanes[i].busy) { li = i; return true; }
            return false;
        });
        m->lanes[li].busy = true;
    }
    skm_mapper::Lane &lane = m->lanes[li];
    auto release = on_exit([&]() {
        { std::lock_guard<std::mutex> hold(m->q_mu); lane.busy = false; }
        m->done_cv.notify_all();
    });
    if (!lane.stream) HIP_TRY(hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking));
    SKM_TRY(lane.bases.ensure((size_t)n_bytes + 64));
    SKM_TRY(lane.offsets.ensure((size_t)n_reads + 1));
    // pinned sources go over the link at full rate and asynchronously; pageable ones are staged
    // by the runtime.  Either way the source is free again when this call returns.
    if (n_bytes)
        HIP_TRY(hipMemcpyAsync(lane.bases.p, bases + first_byte, (size_t)n_bytes, hipMemcpyHostToDevice, lane.stream));
    if (offsets)
        HIP_TRY(hipMemcpyAsync(lane.offsets.p, offsets, (size_t)(n_reads + 1) * sizeof(int64_t), hipMemcpyHostToDevice,
                               lane.stream));
    else
        launch_offsets_uniform(lane.offsets.p, n_reads, uniform_len, lane.stream);
    HIP_TRY(hipStreamSynchronize(lane.stream));
    uint64_t ticket;
    {
        std::lock_guard<std::mutex> hold(m->q_mu);
        ticket = m->next_ticket++;
        m->jobs.push_back(skm_mapper::Job{li, n_units, paired, first_byte, first_unit, ticket});
    }
    release.dismiss();
    m->q_cv.notify_one();
    if (ticket_out) *ticket_out = ticket;
    return SKM_OK;
}

}  // namespace

extern "C" int skm_mapper_map_batch_async(skm_mapper *m, const char *bases, const int64_t *offsets,
                                          int64_t n_units, int paired, int64_t first_unit)
{
    if (!offsets) return fail(SKM_ERR_ARG, "offsets is NULL");
    return submit_batch(m, bases, offsets, -1, n_units, paired, first_unit, nullptr);
}

extern "C" int skm_mapper_map_batch_uniform_async(skm_mapper *m, const char *bases, int32_t read_len,
                                                  int64_t n_units, int paired, int64_t first_unit)
{
    if (read_len < 0) return fail(SKM_ERR_ARG, "negative read length");
    return submit_batch(m, bases, nullptr, read_len, n_units, paired, first_unit, nullptr);
}

extern "C" int skm_mapper_expect_units(skm_mapper *m, int64_t n_units)
{
    if (!m || n_units < 0) return fail(SKM_ERR_ARG, "bad argument");
    std::lock_guard<std::mutex> hold(m->q_mu);
    m->expected_units = n_units;
    m->packed_sized = false;
    return SKM_OK;
}

extern "C" int skm_mapper_sync(skm_mapper *m)
{
    if (!m) return fail(SKM_ERR_ARG, "NULL mapper");
    return wait_jobs(m, 0, true);
}

namespace {

// A piece on its way into the mapper: `stage` checks it, waits at the byte limit, takes a block of HBM
// and QUEUES the copies on the mapper's copy stream (*staged = false: a cut or an empty piece, dealt
// with on the spot); `commit` -- once the copies have completed -- makes the piece visible to the worker.
int packed_stage(skm_mapper *m, const skm_packed_reads *piece, int paired, skm_mapper::Piece *out, bool *staged);
int packed_commit(skm_mapper *m, skm_mapper::Piece &&held, int stream_index);

}  // namespace

extern "C" int skm_mapper_push_packed(skm_mapper *m, const skm_packed_reads *piece, int paired)
{
    if (!m || !piece) return fail(SKM_ERR_ARG, "NULL argument");
    skm_mapper::Piece held;
    bool staged = false;
    SKM_TRY(packed_stage(m, piece, paired, &held, &staged));
    if (!staged) return SKM_OK;
    HIP_TRY(hipStreamSynchronize(m->packed_stream));        // (the caller's arrays are free again)
    return packed_commit(m, std::move(held), piece->stream);
}

namespace {

int packed_stage(skm_mapper *m, const skm_packed_reads *piece, int paired, skm_mapper::Piece *out, bool *staged)
{
    *staged = false;
    const int64_t n = piece->n_reads;
    const int cw = piece->code_words;
    if (n < 0 || n >= (1LL << 31) || piece->first_read < 0) return fail(SKM_ERR_ARG, "bad piece: %lld reads from %lld", (long long)n, (long long)piece->first_read);
    if (piece->stream < 0 || piece->stream > (paired ? 1 : 0)) return fail(SKM_ERR_ARG, "stream %d of a %s sample", piece->stream, paired ? "paired" : "single-ended");
    if (n == 0 && cw == SKM_PACKED_CUT) {
        // a cut: the stream's reads from first_read on are dropped (the longer file of a pair of
        // files, seekmer/common.py:180-197: zip() ends at the shorter one).  The run the worker may
        // be mapping from the same piece ends where both streams had reads, i.e. at or below the cut.
        std::unique_lock<std::mutex> hold(m->q_mu);
        auto &list = m->pending[piece->stream];
        const int64_t from = piece->first_read;
        m->done_cv.wait(hold, [&] {
            if (!m->packed_busy) return true;
            for (const auto &other : list)
                if (other.in_job && other.first + other.n > from) return false;
            return true;
        });
        while (!list.empty() && list.back().first >= from) { m->packed_dropped += list.back().n; list.pop_back(); }
        if (!list.empty() && list.back().first + list.back().n > from) {
            m->packed_dropped += list.back().first + list.back().n - from;
            list.back().n = from - list.back().first;
        }
        return SKM_OK;
    }
    if (n == 0) return SKM_OK;
    if (cw < 1 || cw > (1 << 15) || piece->read_stride < cw || !piece->codes) return fail(SKM_ERR_ARG, "bad code words");
    if (piece->uniform_len < 0 && !piece->lengths) return fail(SKM_ERR_ARG, "lengths is NULL");
    if (piece->uniform_len > 32LL * cw) return fail(SKM_ERR_ARG, "reads of %lld bases in %d code words", (long long)piece->uniform_len, cw);
    const int64_t n_exc = piece->n_exceptions;
    if (n_exc < 0 || n_exc > n || (n_exc > 0 && (!piece->exception_reads || !piece->exception_masks)))
        return fail(SKM_ERR_ARG, "bad exception list");
    for (int64_t e = 0; e < n_exc; ++e)
        if (piece->exception_reads[e] >= (uint64_t)n || (e && piece->exception_reads[e] <= piece->exception_reads[e - 1]))
            return fail(SKM_ERR_ARG, "exception reads must ascend and lie inside the piece");
    SKM_TRY(set_device(m->ix->device));
    {
        std::lock_guard<std::mutex> hold(m->q_mu);
        if (m->packed_paired >= 0 && m->packed_paired != (paired ? 1 : 0) && (!m->pending[0].empty() || !m->pending[1].empty() || m->packed_busy))
            return fail(SKM_ERR_STATE, "paired and single-ended pieces in one run");
        m->packed_paired = paired ? 1 : 0;
        if (!m->packed_stream) HIP_TRY(hipStreamCreateWithFlags(&m->packed_stream, hipStreamNonBlocking));
    }
    // one HBM block: codes | lengths | exception reads | exception bit planes
    const bool uniform = piece->uniform_len >= 0;
    auto round = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t codes_bytes = round((size_t)n * cw * 8);
    const size_t len_bytes = uniform ? 0 : round((size_t)n * 4);
    const size_t exc_bytes = round((size_t)n_exc * 4), mask_bytes = round((size_t)n_exc * cw * 4);
    const int64_t block_bytes = (int64_t)(codes_bytes + len_bytes + exc_bytes + mask_bytes + 256);
    {
        std::unique_lock<std::mutex> hold(m->q_mu);
        auto may_go = [&] {
            if (m->packed_bytes->load() + block_bytes <= m->packed_max_pending || m->job_error != SKM_OK) return true;
            int64_t lo, hi;
            return !(m->packed_busy || packed_find_run(m, &lo, &hi));     // nothing the worker could free
        };
        if (!may_go()) {
            m->packed_waiters++;                  // the worker now maps runs below PACKED_MIN_UNITS too
            m->q_cv.notify_all();
            m->done_cv.wait(hold, may_go);
            m->packed_waiters--;
        }
    }
    char *raw = nullptr;
    HIP_TRY(pool_alloc((void **)&raw, (size_t)block_bytes));
    skm_mapper::Piece &held = *out;
    {
        std::shared_ptr<std::atomic<int64_t>> counter = m->packed_bytes;   // (outlives the mapper if a block does)
        counter->fetch_add(block_bytes);
        held.block = std::shared_ptr<char>(raw, [counter, block_bytes](char *q) { pool_free(q); counter->fetch_sub(block_bytes); });
    }
    held.first = held.origin = piece->first_read;
    held.n = n;
    held.cw = cw;
    held.uniform_len = uniform ? piece->uniform_len : -1;
    held.codes = (uint64_t *)raw;
    held.lengths = uniform ? nullptr : (uint32_t *)(raw + codes_bytes);
    held.exc_reads_dev = (uint32_t *)(raw + codes_bytes + len_bytes);
    held.exc_masks = (uint32_t *)(raw + codes_bytes + len_bytes + exc_bytes);
    if (n_exc) held.exc_reads.assign(piece->exception_reads, piece->exception_reads + n_exc);
    hipStream_t stream = m->packed_stream;
    if (piece->read_stride == cw)
        HIP_TRY(hipMemcpyAsync(held.codes, piece->codes, (size_t)n * cw * 8, hipMemcpyHostToDevice, stream));
    else
        HIP_TRY(hipMemcpy2DAsync(held.codes, (size_t)cw * 8, piece->codes, (size_t)piece->read_stride * 8, (size_t)cw * 8,
                                 (size_t)n, hipMemcpyHostToDevice, stream));
    if (!uniform) HIP_TRY(hipMemcpyAsync(held.lengths, piece->lengths, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    if (n_exc) {
        HIP_TRY(hipMemcpyAsync(held.exc_reads_dev, piece->exception_reads, (size_t)n_exc * 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(held.exc_masks, piece->exception_masks, (size_t)n_exc * cw * 4, hipMemcpyHostToDevice, stream));
    }
    *staged = true;
    return SKM_OK;
}

int packed_commit(skm_mapper *m, skm_mapper::Piece &&held, int stream_index)
{
    {
        std::unique_lock<std::mutex> hold(m->q_mu);
        if (!m->worker_started) {
            m->worker = std::thread(worker_main, m);
            m->worker_started = true;
        }
        auto &list = m->pending[stream_index];
        // a piece that overlaps what its stream holds replaces the reads from its first one on
        auto overlapping = [&] {
            for (const auto &other : list)
                if (other.first < held.first + held.n && held.first < other.first + other.n) return true;
            return false;
        };
        bool overlaps = overlapping();
        if (overlaps) {
            // a piece the worker is mapping from right now (the longer file of a pair of unequal
            // length: its leftover reads are what this piece replaces): the run in flight ends where
            // both streams had reads, so waiting for it leaves only the leftover to cut off -- the
            // outcome does not depend on when the worker took the run
            auto in_flight = [&] {
                for (const auto &other : list)
                    if (other.in_job && other.first + other.n > held.first) return true;
                return false;
            };
            m->done_cv.wait(hold, [&] { return !m->packed_busy || !in_flight(); });
            for (const auto &other : list)
                if (other.first < held.first && other.first + other.n > held.first && other.in_job)
                    return fail(SKM_ERR_STATE, "a piece replaces reads that are being mapped");
            overlaps = overlapping();
        }
        if (overlaps) {
            while (!list.empty() && list.back().first >= held.first) list.pop_back();
            if (!list.empty() && list.back().first + list.back().n > held.first) list.back().n = held.first - list.back().first;
            list.push_back(std::move(held));
        } else {
            size_t at = list.size();
            while (at > 0 && list[at - 1].first > held.first) --at;
            list.insert(list.begin() + (long)at, std::move(held));
        }
    }
    m->q_cv.notify_all();
    return SKM_OK;
}

}  // namespace

extern "C" int skm_mapper_map_packed_source(skm_mapper *m, skm_packed_source next, void *context, int paired,
                                            int64_t *n_pieces)
{
    if (!m || !next) return fail(SKM_ERR_ARG, "NULL argument");
    if (n_pieces) *n_pieces = 0;
    // The copy of piece k runs while the source is asked for piece k + 1 (a source keeps a piece's
    // arrays valid through ONE more call, include/seekmer_hip.h): the drain loop -- one thread --
    // paid every piece's copy time in full before (524 pieces x 25 us of a 42 ms pass).
    SKM_TRY(set_device(m->ix->device));
    hipEvent_t copied[2] = {nullptr, nullptr};
    skm_mapper::Piece waiting;
    bool have_waiting = false;
    auto undo = on_exit([&]() {
        // (an early return with a copy still queued: its block goes back to the pool with `waiting`)
        if (have_waiting) (void)hipStreamSynchronize(m->packed_stream);
        for (auto &e : copied) pool_event_release(e, false);
    });
    for (auto &e : copied) HIP_TRY(pool_event_acquire(&e, false));
    int waiting_stream = 0, turn = 0;
    auto land = [&]() -> int {                   // the piece whose copy was queued a call ago joins its stream
        if (!have_waiting) return SKM_OK;
        have_waiting = false;
        HIP_TRY(hipEventSynchronize(copied[turn ^ 1]));
        return packed_commit(m, std::move(waiting), waiting_stream);
    };
    for (;;) {
        skm_packed_reads piece;
        memset(&piece, 0, sizeof(piece));
        const int rc = next(context, &piece);
        if (rc != SKM_OK) { (void)land(); return fail(rc, "the source of packed reads failed (%d)", rc); }
        if (piece.n_reads == 0 && piece.code_words != SKM_PACKED_CUT) return land();
        if (piece.n_reads == 0) SKM_TRY(land());         // a cut applies behind everything that came before it
        skm_mapper::Piece held;
        bool staged = false;
        const int staged_rc = packed_stage(m, &piece, paired, &held, &staged);
        if (staged_rc != SKM_OK) { (void)land(); return staged_rc; }
        if (staged) HIP_TRY(hipEventRecord(copied[turn], m->packed_stream));
        SKM_TRY(land());
        if (staged) {
            waiting = std::move(held);
            waiting_stream = piece.stream;
            have_waiting = true;
            turn ^= 1;
        }
        if (n_pieces) ++*n_pieces;
    }
}

extern "C" int skm_mapper_map_batch(skm_mapper *m, const char *bases, const int64_t *offsets,
                                    int64_t n_units, int paired)
{
    uint64_t ticket = 0;
    if (!offsets) return fail(SKM_ERR_ARG, "offsets is NULL");
    SKM_TRY(submit_batch(m, bases, offsets, -1, n_units, paired, -1, &ticket));
    return wait_jobs(m, ticket, true);
}

extern "C" int skm_mapper_map_batch_device(skm_mapper *m, const void *d_bases, const void *d_offsets,
                                           int64_t n_units, int paired, int32_t max_read_len)
{
    if (!m || !d_offsets || n_units < 0 || max_read_len < 0)
        return fail(SKM_ERR_ARG, "bad argument");
    SKM_TRY(wait_jobs(m, 0, false));           // queued host batches first
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    return map_batch_resident(m, (const uint8_t *)d_bases, (const int64_t *)d_offsets, n_units,
                              paired, max_read_len);
}

extern "C" int skm_mapper_last_batch(skm_mapper *m, int32_t *begin, int32_t *end,
                                     int32_t *anchor_entry, int32_t *anchor_offset, int32_t *counts,
                                     int32_t *entries, int64_t cap_entries, int64_t *n_entries)
{
    if (!m) return fail(SKM_ERR_ARG, "NULL mapper");
    SKM_TRY(wait_jobs(m, 0, false));           // queued host batches first
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    const int64_t n = m->last_units;
    if (n_entries) *n_entries = 0;
    if (n == 0) return SKM_OK;
    if ((begin || end || anchor_entry || anchor_offset) && !m->last_spans)
        return fail(SKM_ERR_STATE, "the spans of the last batch were not kept: call skm_mapper_keep_spans(mapper, 1) before mapping");
    if (begin) HIP_TRY(hipMemcpy(begin, m->unit_begin.p, n * 4, hipMemcpyDeviceToHost));
    if (end) HIP_TRY(hipMemcpy(end, m->unit_end.p, n * 4, hipMemcpyDeviceToHost));
    if (anchor_entry || anchor_offset) {
        std::vector<Coord> a(n);
        HIP_TRY(hipMemcpy(a.data(), m->unit_anchor.p, n * sizeof(Coord), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) {
            if (anchor_entry) anchor_entry[i] = a[i].entry;
            if (anchor_offset) anchor_offset[i] = a[i].offset;
        }
    }
    if (!counts && !entries && !n_entries) return SKM_OK;
    // the records of the batch (emission order) -> per unit
    std::vector<int32_t> unit(n);
    std::vector<unsigned long long> tuple(n);
    HIP_TRY(hipMemcpy(unit.data(), m->rec_unit.p, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tuple.data(), m->rec_tuple.p, n * 8, hipMemcpyDeviceToHost));
    std::vector<int32_t> cnt(n, 0);
    std::vector<int64_t> off(n, 0);
    std::vector<char> seen(n, 0);
    for (int64_t r = 0; r < n; ++r) {
        const int64_t u = unit[r];
        if (u < 0 || u >= n || seen[u]) return fail(SKM_ERR_STATE, "record %lld names unit %lld", (long long)r, (long long)u);
        seen[u] = 1;
        cnt[u] = (int32_t)(tuple[r] >> 40);
        off[u] = (int64_t)(tuple[r] & ((1ULL << 40) - 1));
    }
    if (counts) memcpy(counts, cnt.data(), n * 4);
    if (n_entries) {
        int64_t total = 0;
        for (int64_t u = 0; u < n; ++u) total += cnt[u];
        *n_entries = total;           // the arena itself has per-wave slack
    }
    if (entries) {
        std::vector<int32_t> raw((size_t)std::max<int64_t>(m->last_ids, 1));
        if (m->last_ids)
            HIP_TRY(hipMemcpy(raw.data(), m->unit_entries.p, (size_t)m->last_ids * 4, hipMemcpyDeviceToHost));
        int64_t pos = 0;
        for (int64_t u = 0; u < n; ++u) {
            for (int i = 0; i < cnt[u]; ++i)
                if (pos + i < cap_entries) entries[pos + i] = raw[off[u] + i];
            pos += cnt[u];
        }
    }
    return SKM_OK;
}

extern "C" int skm_mapper_keep_spans(skm_mapper *m, int enable)
{
    if (!m) return fail(SKM_ERR_ARG, "NULL mapper");
    (void)wait_jobs(m, 0, false);
    std::lock_guard<std::mutex> lock(m->mu);
    m->keep_spans = enable != 0;
    return SKM_OK;
}

extern "C" int skm_mapper_summary(skm_mapper *m, int64_t summary[4])
{
    if (!m || !summary) return fail(SKM_ERR_ARG, "NULL argument");
    SKM_TRY(wait_jobs(m, 0, false));           // queued host batches first
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    unsigned long long ctr[4];
    HIP_TRY(hipMemcpy(ctr, m->counters.p, sizeof(ctr), hipMemcpyDeviceToHost));
    summary[0] = (int64_t)ctr[CTR_CLASSES];
    summary[1] = (int64_t)ctr[CTR_ARENA];
    summary[2] = (int64_t)ctr[CTR_UNALIGNED];
    summary[3] = (int64_t)ctr[CTR_UNITS];
    return SKM_OK;
}

extern "C" int skm_mapper_export(skm_mapper *m, int64_t *class_offsets, int32_t *class_targets,
                                 int64_t *class_counts, int64_t *first_seen, int64_t *fld)
{
    if (!m) return fail(SKM_ERR_ARG, "NULL mapper");
    SKM_TRY(wait_jobs(m, 0, false));           // queued host batches first
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    if (fld)
        HIP_TRY(hipMemcpy(fld, m->counters.p + CTR_FLD, MAX_FRAGMENT_LENGTH * 8, hipMemcpyDeviceToHost));
    if (!class_offsets && !class_targets && !class_counts && !first_seen) return SKM_OK;
    const int64_t C = m->host_classes, M = m->host_arena_used;
    if (class_offsets) class_offsets[0] = 0;
    if (C == 0) return SKM_OK;
    DBuf<int64_t> d_off, d_len; DBuf<double> d_cnt; DBuf<unsigned long long> d_fs;
    auto undo = on_exit([&]() { d_off.release(); d_len.release(); d_cnt.release(); d_fs.release(); });
    SKM_TRY(d_off.ensure(C)); SKM_TRY(d_len.ensure(C)); SKM_TRY(d_cnt.ensure(C)); SKM_TRY(d_fs.ensure(C));
    launch_class_compact(m->t, C, d_off.p, d_len.p, d_cnt.p, d_fs.p, m->stream);
    HIP_TRY(hipGetLastError());
    std::vector<int64_t> off(C), len(C);
    std::vector<double> cnt(C);
    std::vector<unsigned long long> fs(C);
    std::vector<int32_t> arena((size_t)std::max<int64_t>(M, 1));
    HIP_TRY(hipMemcpyAsync(off.data(), d_off.p, C * 8, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(len.data(), d_len.p, C * 8, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(cnt.data(), d_cnt.p, C * 8, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(fs.data(), d_fs.p, C * 8, hipMemcpyDeviceToHost, m->stream));
    if (M) HIP_TRY(hipMemcpyAsync(arena.data(), m->arena.p, (size_t)M * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    // Counter insertion order under -j1 = ascending first-seen unit (mapper.py:88)
    std::vector<int64_t> order(C);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return fs[a] < fs[b]; });
    int64_t pos = 0;
    for (int64_t k = 0; k < C; ++k) {
        const int64_t c = order[k];
        if (class_targets)
            memcpy(class_targets + pos, arena.data() + off[c], (size_t)len[c] * 4);
        pos += len[c];
        if (class_offsets) class_offsets[k + 1] = pos;
        if (class_counts) class_counts[k] = (int64_t)cnt[c];
        if (first_seen) first_seen[k] = (int64_t)fs[c];
    }
    return SKM_OK;
}

extern "C" int skm_mapper_merge(skm_mapper *m, int64_t n_classes, const int64_t *class_offsets,
                                const int32_t *class_targets, const int64_t *class_counts,
                                const int64_t *first_seen, int64_t unaligned, const int64_t *fld)
{
    if (!m || n_classes < 0 || unaligned < 0) return fail(SKM_ERR_ARG, "bad argument");
    if (n_classes && (!class_offsets || !class_targets || !class_counts || !first_seen))
        return fail(SKM_ERR_ARG, "NULL class arrays");
    SKM_TRY(wait_jobs(m, 0, false));           // queued host batches first
    std::lock_guard<std::mutex> lock(m->mu);
    m->host_totals_valid = false;             // (the totals change on the device below)
    SKM_TRY(set_device(m->ix->device));
    std::vector<unsigned long long> add(CTR_WORDS, 0);
    int64_t units = unaligned;
    for (int64_t c = 0; c < n_classes; ++c) units += class_counts[c];
    add[CTR_UNALIGNED] = (unsigned long long)unaligned;
    add[CTR_UNITS] = (unsigned long long)units;
    if (fld) for (int i = 0; i < MAX_FRAGMENT_LENGTH; ++i) add[CTR_FLD + i] = (unsigned long long)fld[i];
    if (n_classes) {
        const int64_t M = class_offsets[n_classes];
        {   // foreign classes may all be new: size for them at load <= 0.5, unbounded probes
            uint64_t want = 1 << 16;
            while ((double)want * 0.5 < (double)(m->host_classes + n_classes + 1024)) want <<= 1;
            SKM_TRY(table_grow(m, want, 0));
            SKM_TRY(m->class_list.ensure((size_t)(m->host_classes + n_classes + 1024), true, m->stream));
        }
        SKM_TRY(m->arena.ensure((size_t)(m->host_arena_used + M + 1024), true, m->stream));
        bind_table(m, m->t.slot_mask + 1);
        DBuf<int64_t> d_off, d_cnt, d_fs; DBuf<int32_t> d_ids;
        auto undo = on_exit([&]() { d_off.release(); d_cnt.release(); d_fs.release(); d_ids.release(); });
        SKM_TRY(d_off.ensure(n_classes + 1)); SKM_TRY(d_cnt.ensure(n_classes));
        SKM_TRY(d_fs.ensure(n_classes)); SKM_TRY(d_ids.ensure(std::max<int64_t>(M, 1)));
        HIP_TRY(hipMemcpy(d_off.p, class_offsets, (n_classes + 1) * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_cnt.p, class_counts, n_classes * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_fs.p, first_seen, n_classes * 8, hipMemcpyHostToDevice));
        if (M) HIP_TRY(hipMemcpy(d_ids.p, class_targets, M * 4, hipMemcpyHostToDevice));
        launch_class_merge(m->t, n_classes, d_off.p, d_ids.p, d_cnt.p, d_fs.p, m->stream);
        HIP_TRY(hipGetLastError());
        unsigned long long ctr[8];
        HIP_TRY(hipMemcpyAsync(ctr, m->counters.p, sizeof(ctr), hipMemcpyDeviceToHost, m->stream));
        SKM_TRY(read_error(m));
        m->host_arena_used = (int64_t)ctr[CTR_ARENA];
        m->host_classes = (int64_t)ctr[CTR_CLASSES];
    }
    // totals and histogram only once the classes are in: a failed merge leaves them untouched
    std::vector<unsigned long long> cur(CTR_WORDS);
    HIP_TRY(hipMemcpy(cur.data(), m->counters.p, CTR_WORDS * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < CTR_WORDS; ++i) cur[i] += add[i];
    HIP_TRY(hipMemcpy(m->counters.p, cur.data(), CTR_WORDS * 8, hipMemcpyHostToDevice));
    m->units_done += units;
    for (int64_t c = 0; c < n_classes; ++c)
        m->first_seen_bound = std::max(m->first_seen_bound, first_seen[c] + 1);
    m->first_seen_bound = std::max(m->first_seen_bound, m->units_done);
    return SKM_OK;
}

// The table where it lies (SURVEY 8(e).1 without the host: the hand-over between GPUs is then a
// copy of these arrays over xGMI, ncclSend / ncclRecv or a peer copy, and a merge by key).
extern "C" int skm_mapper_device_table(skm_mapper *m, skm_device_table *out)
{
    if (!m || !out) return fail(SKM_ERR_ARG, "NULL argument");
    SKM_TRY(wait_jobs(m, 0, false));           // queued host batches first
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    memset(out, 0, sizeof(*out));
    unsigned long long ctr[4];
    HIP_TRY(hipMemcpyAsync(ctr, m->counters.p, sizeof(ctr), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    const int64_t C = (int64_t)ctr[CTR_CLASSES];
    out->device = m->ix->device;
    out->n_classes = C;
    out->n_ids = (int64_t)ctr[CTR_ARENA];
    out->unaligned = (int64_t)ctr[CTR_UNALIGNED];
    out->units = (int64_t)ctr[CTR_UNITS];
    out->first_seen_bound = m->first_seen_bound;
    out->fld = (const uint64_t *)(m->counters.p + CTR_FLD);
    out->ids = m->arena.p;
    if (C == 0) return SKM_OK;
    SKM_TRY(m->view_start.ensure(C)); SKM_TRY(m->view_len.ensure(C));
    SKM_TRY(m->view_count.ensure(C)); SKM_TRY(m->view_first_seen.ensure(C));
    launch_class_compact(m->t, C, m->view_start.p, m->view_len.p, m->view_count.p, m->view_first_seen.p, m->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(m->stream));
    out->class_start = m->view_start.p;
    out->class_len = m->view_len.p;
    out->class_count = m->view_count.p;
    out->first_seen = (const uint64_t *)m->view_first_seen.p;
    return SKM_OK;
}

extern "C" int skm_mapper_merge_device(skm_mapper *m, const skm_device_table *table)
{
    if (!m || !table || table->n_classes < 0 || table->n_ids < 0 || table->unaligned < 0 || table->units < 0)
        return fail(SKM_ERR_ARG, "bad argument");
    if (table->n_classes && (!table->class_start || !table->class_len || !table->class_count || !table->first_seen || !table->ids))
        return fail(SKM_ERR_ARG, "NULL class arrays");
    if (table->device != m->ix->device)
        return fail(SKM_ERR_ARG, "the table lies on GPU %d, the mapper on GPU %d: copy it over first", table->device, m->ix->device);
    SKM_TRY(wait_jobs(m, 0, false));
    std::lock_guard<std::mutex> lock(m->mu);
    m->host_totals_valid = false;
    SKM_TRY(set_device(m->ix->device));
    const int64_t n_classes = table->n_classes;
    {   // foreign classes may all be new: size for them at load <= 0.5, unbounded probes
        uint64_t want = 1 << 16;
        while ((double)want * 0.5 < (double)(m->host_classes + n_classes + 1024)) want <<= 1;
        SKM_TRY(table_grow(m, want, 0));
        SKM_TRY(m->class_list.ensure((size_t)(m->host_classes + n_classes + 1024), true, m->stream));
    }
    SKM_TRY(m->arena.ensure((size_t)(m->host_arena_used + table->n_ids + 1024), true, m->stream));
    bind_table(m, m->t.slot_mask + 1);
    launch_class_merge_device(m->t, n_classes, table->class_start, table->class_len, table->ids, table->class_count,
                              (const unsigned long long *)table->first_seen, (unsigned long long)table->unaligned,
                              (unsigned long long)table->units, (const unsigned long long *)table->fld, m->stream);
    HIP_TRY(hipGetLastError());
    unsigned long long ctr[8];
    HIP_TRY(hipMemcpyAsync(ctr, m->counters.p, sizeof(ctr), hipMemcpyDeviceToHost, m->stream));
    SKM_TRY(read_error(m));
    m->host_arena_used = (int64_t)ctr[CTR_ARENA];
    m->host_classes = (int64_t)ctr[CTR_CLASSES];
    m->host_units = ctr[CTR_UNITS];
    m->host_unaligned = ctr[CTR_UNALIGNED];
    m->host_totals_valid = true;
    m->units_done += table->units;
    m->first_seen_bound = std::max(m->first_seen_bound, std::max(table->first_seen_bound, m->units_done));
    return SKM_OK;
}

extern "C" int skm_mapper_clear(skm_mapper *m)
{
    if (!m) return fail(SKM_ERR_ARG, "NULL mapper");
    (void)wait_jobs(m, 0, true);            // (a failed queued batch is forgotten with the table)
    {
        std::lock_guard<std::mutex> hold(m->q_mu);      // (and packed reads that never got a mate)
        packed_drop_all(m);
    }
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    // MapResult.clear only clears the counter (mapper.py:143-145): the FLD stays
    std::vector<unsigned long long> fld(MAX_FRAGMENT_LENGTH);
    HIP_TRY(hipMemcpy(fld.data(), m->counters.p + CTR_FLD, MAX_FRAGMENT_LENGTH * 8, hipMemcpyDeviceToHost));
    SKM_TRY(table_reset(m, m->t.slot_mask + 1));
    HIP_TRY(hipMemcpyAsync(m->counters.p + CTR_FLD, fld.data(), MAX_FRAGMENT_LENGTH * 8, hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return SKM_OK;
}

extern "C" int skm_mapper_reset(skm_mapper *m)
{
    if (!m) return fail(SKM_ERR_ARG, "NULL mapper");
    (void)wait_jobs(m, 0, true);            // (a failed queued batch is forgotten with the table)
    {
        std::lock_guard<std::mutex> hold(m->q_mu);      // (and packed reads that never got a mate)
        packed_drop_all(m);
    }
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    SKM_TRY(table_reset(m, m->t.slot_mask + 1));
    HIP_TRY(hipStreamSynchronize(m->stream));         // (readers of the table use streams of their own)
    m->last_units = 0;
    m->last_ids = 0;
    return SKM_OK;
}

extern "C" int skm_mapper_timing(skm_mapper *m, double stats[8])
{
    if (!m || !stats) return fail(SKM_ERR_ARG, "NULL argument");
    (void)wait_jobs(m, 0, false);
    std::lock_guard<std::mutex> lock(m->mu);
    stats[0] = m->t_pack_ns; stats[1] = m->t_map_ns; stats[2] = m->t_class_ns;
    stats[3] = m->batches; stats[4] = (double)m->units_done;
    stats[5] = m->t_em_ns; stats[6] = m->em_iters; stats[7] = 0;
    return SKM_OK;
}

// access counters of the STATS build of the map kernel (SKM_MAP_STATS=1):
// [0]=reads [1]=read bases [2]=lookups [3]=slots [4]=contig reads [5]=targets
// copied [6]=targets merged [7]=8-base fetches [8]=merges [9]=tuple ids
extern "C" int skm_mapper_access_stats(skm_mapper *m, int64_t out[48])
{
    if (!m || !out) return fail(SKM_ERR_ARG, "NULL argument");
    (void)wait_jobs(m, 0, false);
    std::lock_guard<std::mutex> lock(m->mu);
    for (int i = 0; i < 48; ++i) out[i] = (int64_t)m->stats_total[i];
    return SKM_OK;
}

extern "C" int skm_mapper_set_stats(skm_mapper *m, int enable)
{
    if (!m) return fail(SKM_ERR_ARG, "NULL mapper");
    (void)wait_jobs(m, 0, false);
    std::lock_guard<std::mutex> lock(m->mu);
    m->want_stats = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    for (auto &v : m->stats_total) v = 0;
    return SKM_OK;
}

// ------------------------------------------------------------ quantification
extern "C" int skm_effective_lengths(int device, const int64_t *fld, const double *lengths,
                                     int64_t n_tx, double *out)
{
    if (!fld || !lengths || !out || n_tx < 0) return fail(SKM_ERR_ARG, "bad argument");
    int n_dev = 0;
    SKM_TRY(skm_device_count(&n_dev));
    if (device < 0 || device >= n_dev) return fail(SKM_ERR_ARG, "device %d out of range", device);
    SKM_TRY(set_device(device));
    if (n_tx == 0) return SKM_OK;
    DBuf<unsigned long long> d_fld; DBuf<double> d_len, d_out;
    SKM_TRY(d_fld.ensure(MAX_FRAGMENT_LENGTH)); SKM_TRY(d_len.ensure(n_tx)); SKM_TRY(d_out.ensure(n_tx));
    HIP_TRY(hipMemcpy(d_fld.p, fld, MAX_FRAGMENT_LENGTH * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_len.p, lengths, n_tx * 8, hipMemcpyHostToDevice));
    launch_effective_lengths(d_fld.p, d_len.p, n_tx, d_out.p, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, n_tx * 8, hipMemcpyDeviceToHost));
    d_fld.release(); d_len.release(); d_out.release();
    return SKM_OK;
}

namespace {

int quant_alloc(skm_quant *q, int device, int64_t n_tx, int64_t n_classes, int64_t n_ids)
{
    STALE_CHECK("quant_alloc entry");
    q->device = device;
    q->n_tx = n_tx;
    q->n_classes = n_classes;
    q->n_ids = n_ids;
    HIP_TRY(pool_stream_acquire(&q->stream));
    for (auto &e : q->ev) HIP_TRY(pool_event_acquire(&e, true));
    for (auto &e : q->chunk_ev) HIP_TRY(pool_event_acquire(&e, false));
    HIP_TRY(pool_pinned_acquire((void **)&q->pinned));
    const size_t C = (size_t)std::max<int64_t>(n_classes, 1), M = (size_t)std::max<int64_t>(n_ids, 1);
    const size_t T = (size_t)std::max<int64_t>(n_tx, 1);
    const size_t R = (size_t)quant_rows_upper_bound(n_tx, n_ids);
    SKM_TRY(q->cls_offset.ensure(C + 1));
    SKM_TRY(q->cls_count.ensure(C));
    SKM_TRY(q->inner.ensure(C));
    SKM_TRY(q->ids.ensure(M));
    SKM_TRY(q->perm.ensure(C));
    SKM_TRY(q->tx_cls.ensure(M));
    SKM_TRY(q->tx_row.ensure(T + 1));
    SKM_TRY(q->row_start.ensure(R + 1));
    SKM_TRY(q->row_tx.ensure(R));
    SKM_TRY(q->row_sum.ensure(R));
    SKM_TRY(q->eff_len.ensure(T));
    SKM_TRY(q->x0.ensure(T));
    SKM_TRY(q->x1.ensure(T));
    SKM_TRY(q->acc.ensure(T));
    SKM_TRY(q->ctl.ensure(16));
    SKM_TRY(q->part_max.ensure(EM_FINAL_BLOCKS));
    SKM_TRY(q->part_flags.ensure(EM_FINAL_BLOCKS));
    SKM_TRY(q->arrivals.ensure(T));
    HIP_TRY(hipMemsetAsync(q->arrivals.p, 0, T * sizeof(unsigned int), q->stream));
    STALE_CHECK("quant_alloc exit");
    return SKM_OK;
}

QuantBuild quant_build_view(skm_quant *q)
{
    QuantBuild b{};
    b.n_tx = q->n_tx;
    b.n_classes = q->n_classes;
    b.n_ids = q->n_ids;
    b.cls_offset = q->cls_offset.p;
    b.ids = q->ids.p;
    b.cls_count = q->cls_count.p;
    b.tx_cls = q->tx_cls.p;
    b.tx_row = q->tx_row.p;
    b.row_start = q->row_start.p;
    b.row_tx = q->row_tx.p;
    b.n_rows_cap = (int64_t)q->row_tx.cap;
    return b;
}

int quant_finish_setup(skm_quant *q, const ClassTable *table, int64_t units_seen = -1)
{
    STALE_CHECK("quant_finish_setup");
    QuantBuild b = quant_build_view(q);
    b.first_seen_bound = units_seen > 0 ? units_seen : 0;      // first-seen values are unit indices below this
    const int64_t rows = quant_setup(table, b, q->perm.p, q->stream);
    if (rows < 0) return fail(SKM_ERR_HIP, "building the class views failed (%lld): %s", (long long)rows, quant_setup_failure());
    q->n_rows = rows;
    return SKM_OK;
}

// ---- RCCL through dlopen: the library is only needed for N > 1
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;      // (table hand-over; optional)
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
struct UniqueId { char internal[128]; };
typedef int (*comm_init_fn)(void **, int, UniqueId, int);
Rccl g_rccl;
comm_init_fn g_comm_init = nullptr;

int load_rccl()
{
    if (g_rccl.lib) return SKM_OK;
    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(SKM_ERR_COMM, "cannot load librccl.so: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
    g_comm_init = (comm_init_fn)dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
    g_rccl.CommCount = (int (*)(void *, int *))dlsym(lib, "ncclCommCount");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
    g_rccl.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclSend");
    g_rccl.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclRecv");
    g_rccl.GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
    if (!g_rccl.GetUniqueId || !g_comm_init || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(SKM_ERR_COMM, "librccl.so lacks a required symbol");
    g_rccl.lib = lib;
    return SKM_OK;
}

constexpr int NCCL_FLOAT64 = 8, NCCL_UINT64 = 5, NCCL_INT32 = 2, NCCL_SUM = 0;

#define NCCL_TRY(call)                                                                   \
    do {                                                                                 \
        int r_ = (call);                                                                 \
        if (r_ != 0)                                                                     \
            return fail(SKM_ERR_COMM, "%s failed: %s", #call,                            \
                        g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?");        \
    } while (0)

EmProblem em_problem(skm_quant *q, double rel_tol, double x_floor, int64_t max_iters,
                     int64_t fixed_iters)
{
    EmProblem p{};
    p.n_tx = q->n_tx;
    p.n_classes = q->n_classes;
    p.n_rows = q->n_rows;
    p.cls_offset = q->cls_offset.p;
    p.ids = q->ids.p;
    p.cls_count = q->cls_count.p;
    p.inner = q->inner.p;
    p.row_start = q->row_start.p;
    p.row_tx = q->row_tx.p;
    p.tx_cls = q->tx_cls.p;
    p.tx_row = q->tx_row.p;
    p.row_sum = q->row_sum.p;
    p.eff_len = q->eff_len.p;
    p.x[0] = q->x0.p;
    p.x[1] = q->x1.p;
    p.acc = q->acc.p;
    p.n_total = q->n_total;
    p.rel_tol = rel_tol;
    p.x_floor = x_floor;
    p.ctl = q->ctl.p;
    p.part_max = q->part_max.p;
    p.part_flags = q->part_flags.p;
    p.max_iters = max_iters;
    p.fixed_iters = fixed_iters;
    static const bool unfused = getenv("SKM_EM_UNFUSED") != nullptr;     // tuning aid: rows and finalize as two launches
    p.fused = q->comm || unfused ? 0 : 1;      // (several ranks: the all-reduce sits between rows and finalize)
    p.arrivals = q->arrivals.p;
    return p;
}

// runs the EM from the abundance already in q->x0; result left in x[iters & 1]
int em_run(skm_quant *q, double rel_tol, double x_floor, int64_t max_iters, int64_t fixed_iters,
           int64_t *iters_out, int64_t chunk_steps = 16)
{
    // n = class_count.sum() over ALL ranks (infer.py:152)
    double n_total = q->n_total;
    if (q->comm && !q->n_total_reduced) {
        HIP_TRY(hipMemcpyAsync(q->acc.p, &n_total, 8, hipMemcpyHostToDevice, q->stream));
        NCCL_TRY(g_rccl.AllReduce(q->acc.p, q->acc.p, 1, NCCL_FLOAT64, NCCL_SUM, q->comm, q->stream));
        HIP_TRY(hipMemcpyAsync(&n_total, q->acc.p, 8, hipMemcpyDeviceToHost, q->stream));
        HIP_TRY(hipStreamSynchronize(q->stream));
    }
    EmProblem p = em_problem(q, rel_tol, x_floor, max_iters, fixed_iters);
    p.n_total = n_total;
    HIP_TRY(hipMemsetAsync(q->ctl.p, 0, 16 * 8, q->stream));
    int64_t k = 0;
    const int64_t chunk = fixed_iters > 0 ? std::min<int64_t>(fixed_iters, chunk_steps) : chunk_steps;
    HIP_TRY(hipEventRecord(q->ev[0], q->stream));
    // Steps are enqueued in chunks; after each chunk the control block is copied to pinned
    // memory and an event recorded.  The host stays one chunk ahead: chunk i+1 is already
    // queued when it waits for chunk i's verdict, so the GPU never idles at a check-point
    // (a converged EM turns at most one chunk of launches into no-ops).
    static const bool unfused_rows = getenv("SKM_EM_UNFUSED") != nullptr;
    auto enqueue_chunk = [&](int slot) -> int {
        for (int64_t i = 0; i < chunk; ++i, ++k) {
            // (the first step of a chunk follows the chunk-end em_decide: already judged)
            launch_em_inner(p, (int)(k & 1), i > 0, k, q->stream);
            if (q->comm) {
                // rows -> this rank's numerators (one launch, or two with SKM_EM_UNFUSED), summed over
                // the ranks, then the finalize on every rank alike
                if (unfused_rows) {
                    launch_em_rows(p, (int)(k & 1), q->stream);
                    launch_em_rows_to_acc(p, q->stream);
                } else {
                    launch_em_rows_acc(p, (int)(k & 1), q->stream);
                }
                NCCL_TRY(g_rccl.AllReduce(q->acc.p, q->acc.p, (size_t)q->n_tx, NCCL_FLOAT64, NCCL_SUM,
                                          q->comm, q->stream));
                launch_em_finalize(p, (int)(k & 1), true, q->stream);
            } else if (p.fused) {
                launch_em_rows_finalize(p, (int)(k & 1), q->stream);     // (rows + finalize: one launch)
            } else {
                launch_em_rows(p, (int)(k & 1), q->stream);
                launch_em_finalize(p, (int)(k & 1), false, q->stream);
            }
            q->launches += q->comm ? (unfused_rows ? 4 : 3) : (p.fused ? 2 : 3);
        }
        launch_em_decide(p, k, q->stream);
        q->launches += 1;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(q->pinned + 8 * slot, q->ctl.p, 8 * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, q->stream));
        HIP_TRY(hipEventRecord(q->chunk_ev[slot], q->stream));
        return SKM_OK;
    };
    unsigned long long ctl[8] = {0};
    SKM_TRY(enqueue_chunk(0));
    for (int slot = 0;; slot ^= 1) {
        SKM_TRY(enqueue_chunk(slot ^ 1));                 // stay one chunk ahead
        HIP_TRY(hipEventSynchronize(q->chunk_ev[slot]));
        memcpy(ctl, q->pinned + 8 * slot, sizeof(ctl));
        if (ctl[0]) break;
    }
    HIP_TRY(hipStreamSynchronize(q->stream));             // drain the look-ahead chunk (no-ops)
    HIP_TRY(hipEventRecord(q->ev[1], q->stream));
    HIP_TRY(hipEventSynchronize(q->ev[1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, q->ev[0], q->ev[1]));
    q->t_em_ns += ms * 1e6;
    q->iters_total += (double)ctl[1];
    if (iters_out) *iters_out = (int64_t)ctl[1];
    if (ctl[3]) return fail(SKM_ERR_UNDEFINED, "no abundance above x_floor: numpy raises on max() of an empty selection");
    return SKM_OK;
}

}  // namespace

namespace {

// numpy.sum of a contiguous f8 array on the host, bit for bit (blocks of 8192, pairwise inside;
// see np_sum_blocks_kernel): n = class_count.sum() of infer.py:152 for counts that are not
// integers (blended single-cell tables, impute.py:248-252)
double np_pairwise_host(const double *a, int64_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_host(a, n2) + np_pairwise_host(a + n2, n - n2);
}

double np_sum_host(const double *a, int64_t n)
{
    double acc = 0.0;
    for (int64_t i = 0; i < n; i += 8192) acc += np_pairwise_host(a + i, std::min<int64_t>(8192, n - i));
    return acc;
}

}  // namespace

extern "C" int skm_quant_create(int device, int64_t n_tx, int64_t n_classes,
                                const int64_t *class_offsets, const int32_t *class_targets,
                                const double *class_counts, skm_quant **out)
{
    if (!out || n_tx <= 0 || n_classes < 0) return fail(SKM_ERR_ARG, "bad argument");
    if (n_classes && (!class_offsets || !class_targets || !class_counts))
        return fail(SKM_ERR_ARG, "NULL class arrays");
    int n_dev = 0;
    SKM_TRY(skm_device_count(&n_dev));
    if (device < 0 || device >= n_dev) return fail(SKM_ERR_ARG, "device %d out of range", device);
    const int64_t M = n_classes ? class_offsets[n_classes] - class_offsets[0] : 0;
    for (int64_t c = 0; c < n_classes; ++c)
        if (class_offsets[c + 1] < class_offsets[c]) return fail(SKM_ERR_ARG, "class offsets are not monotone");
    const double total = np_sum_host(class_counts, n_classes);
    for (int64_t j = 0; j < M; ++j) {
        const int32_t t = class_targets[class_offsets[0] + j];
        if (t < 0 || t >= n_tx) return fail(SKM_ERR_ARG, "class target %d outside [0, n_tx)", t);
    }
    SKM_TRY(set_device(device));
    skm_quant *q = new skm_quant();
    int rc = quant_alloc(q, device, n_tx, n_classes, M);
    if (rc != SKM_OK) { delete q; return rc; }
    std::vector<int64_t> rebased(n_classes + 1, 0);
    for (int64_t c = 0; c <= n_classes && n_classes; ++c) rebased[c] = class_offsets[c] - class_offsets[0];
    HIP_TRY(hipMemcpy(q->cls_offset.p, rebased.data(), (n_classes + 1) * 8, hipMemcpyHostToDevice));
    if (n_classes) {
        HIP_TRY(hipMemcpy(q->cls_count.p, class_counts, n_classes * 8, hipMemcpyHostToDevice));
        if (M) HIP_TRY(hipMemcpy(q->ids.p, class_targets + class_offsets[0], M * 4, hipMemcpyHostToDevice));
    }
    q->n_total = total;
    rc = quant_finish_setup(q, nullptr);
    if (rc != SKM_OK) { skm_quant_destroy(q); return rc; }
    *out = q;
    return SKM_OK;
}

extern "C" int skm_quant_create_from_mapper(skm_mapper *m, int64_t n_tx, skm_quant **out)
{
    if (!m || !out || n_tx <= 0) return fail(SKM_ERR_ARG, "bad argument");
    SKM_TRY(wait_jobs(m, 0, false));           // queued host batches first
    std::lock_guard<std::mutex> lock(m->mu);
    SKM_TRY(set_device(m->ix->device));
    const int64_t C = m->host_classes, M = m->host_arena_used;
    skm_quant *q = new skm_quant();
    int rc = quant_alloc(q, m->ix->device, n_tx, C, M);
    if (rc != SKM_OK) { delete q; return rc; }
    unsigned long long ctr[4];
    HIP_TRY(hipMemcpy(ctr, m->counters.p, sizeof(ctr), hipMemcpyDeviceToHost));
    q->n_total = (double)(ctr[CTR_UNITS] - ctr[CTR_UNALIGNED]);
    rc = quant_finish_setup(q, &m->t, m->first_seen_bound);
    if (rc != SKM_OK) { skm_quant_destroy(q); return rc; }
    *out = q;
    return SKM_OK;
}

// One sample from the resident class table to TPM without leaving the device:
// fragment-length histogram (all-reduced over the ranks of `comm`) -> effective
// lengths (mapper.py:134-141) -> start vector 1/l normalised with numpy's sum
// (infer.py:116-119) -> EM to the stop rule (:133-168) -> TPM scaling (:127-129).
extern "C" int skm_quant_infer(skm_mapper *m, skm_comm *comm, const double *lengths, int64_t n_tx,
                               double rel_tol, double x_floor, int64_t max_iters,
                               double *tpm, double *effective_lengths, int64_t *iters)
{
    if (!m || !lengths || n_tx <= 0) return fail(SKM_ERR_ARG, "bad argument");
    SKM_TRY(wait_jobs(m, 0, false));           // queued host batches first
    std::lock_guard<std::mutex> lock(m->mu);
    if (comm && comm->device != m->ix->device)
        return fail(SKM_ERR_ARG, "communicator and mapper live on different GPUs");
    SKM_TRY(set_device(m->ix->device));
    const int64_t C = m->host_classes, M = m->host_arena_used;
    // SKM_TRACE_INFER=1: host wall time between the phases below, on stderr (tuning aid)
    static const bool trace = getenv("SKM_TRACE_INFER") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[skm_quant_infer] %-12s %8.1f us\n", what,
                std::chrono::duration<double, std::micro>(now - t_last).count());
        t_last = now;
    };
    skm_quant *q = new skm_quant();
    int rc = quant_alloc(q, m->ix->device, n_tx, C, M);
    if (rc != SKM_OK) { delete q; return rc; }
    if (comm) { q->comm = comm->comm; q->rank = comm->rank; q->world = comm->world; }
    lap("alloc");
    DBuf<unsigned long long> fld;
    DBuf<double> sums;
    const int64_t n_blocks = (n_tx + 8191) / 8192;
    auto body = [&]() -> int {
        SKM_TRY(fld.ensure(MAX_FRAGMENT_LENGTH + 1));
        SKM_TRY(sums.ensure(n_blocks + 2));
        double *const total = sums.p + n_blocks;          // [0] sum, [1] sum / divisor
        unsigned long long ctr[4] = {0, 0, m->host_unaligned, m->host_units};
        if (!m->host_totals_valid)
            HIP_TRY(hipMemcpy(ctr, m->counters.p, sizeof(ctr), hipMemcpyDeviceToHost));   // (mapper stream is idle)
        unsigned long long aligned = ctr[CTR_UNITS] - ctr[CTR_UNALIGNED];
        HIP_TRY(hipMemcpyAsync(fld.p, m->counters.p + CTR_FLD, MAX_FRAGMENT_LENGTH * 8,
                               hipMemcpyDeviceToDevice, q->stream));
        if (q->comm) {
            // merge_fragment_lengths over the ranks, and with it (word 2000 of the same
            // collective) n = class_count.sum() of infer.py:152 over ALL ranks.  Whether there
            // is anything to quantify (infer.py:106-107 tests the merged table) must be decided
            // on the global sum: a rank whose shard produced no class still has to take part
            // in every collective of the EM below, with empty class views.
            q->pinned[32] = aligned;
            HIP_TRY(hipMemcpyAsync(fld.p + MAX_FRAGMENT_LENGTH, q->pinned + 32, 8, hipMemcpyHostToDevice, q->stream));
            NCCL_TRY(g_rccl.AllReduce(fld.p, fld.p, MAX_FRAGMENT_LENGTH + 1, NCCL_UINT64, NCCL_SUM, q->comm,
                                      q->stream));
            HIP_TRY(hipMemcpyAsync(q->pinned + 32, fld.p + MAX_FRAGMENT_LENGTH, 8, hipMemcpyDeviceToHost, q->stream));
            HIP_TRY(hipStreamSynchronize(q->stream));
            aligned = q->pinned[32];
            q->n_total_reduced = true;
            // test hook: "the other ranks aligned this many units" -- a rank whose own shard produced no
            // class (C == 0) then goes through the set-up and every collective of the EM with empty
            // class views, a state one rank cannot reach by itself (tests/test_gpu_parity.py)
            if (const char *v = getenv("SKM_TEST_ALIGNED_GLOBAL")) aligned += strtoull(v, nullptr, 10);
        }
        q->n_total = (double)aligned;
        HIP_TRY(hipMemcpyAsync(q->x1.p, lengths, n_tx * 8, hipMemcpyHostToDevice, q->stream));
        launch_effective_lengths(fld.p, q->x1.p, n_tx, q->eff_len.p, q->stream);
        // (the effective lengths go home at the end, with the TPM: a copy to pageable memory holds the
        // host up, and the kernels that follow are not launched meanwhile)
        // quantify(): no class -> zeros (infer.py:106-107); over several ranks "no class anywhere"
        // is "no aligned unit anywhere" (every aligned unit belongs to a class)
        if (q->comm ? aligned == 0 : C == 0) {
            if (effective_lengths)
                HIP_TRY(hipMemcpyAsync(effective_lengths, q->eff_len.p, n_tx * 8, hipMemcpyDeviceToHost, q->stream));
            HIP_TRY(hipStreamSynchronize(q->stream));
            if (tpm) memset(tpm, 0, (size_t)n_tx * 8);
            if (iters) *iters = 0;
            return SKM_OK;
        }
        launch_reciprocal(q->eff_len.p, n_tx, q->x0.p, q->stream);
        launch_np_sum(q->x0.p, n_tx, 1.0, sums.p, total, q->stream);
        launch_divide(q->x0.p, n_tx, total, false, 0.0, q->stream);
        lap("start vector");
        SKM_TRY(quant_finish_setup(q, &m->t, m->first_seen_bound));
        lap("setup");
        if (trace) fprintf(stderr, "[skm_quant_infer] %lld classes, %lld transcripts, %lld rows\n", (long long)C, (long long)n_tx,
                           (long long)q->n_rows);
        int64_t it = 0;
        SKM_TRY(em_run(q, rel_tol, x_floor, max_iters, 0, &it));
        lap("em");
        m->t_em_ns += q->t_em_ns;
        m->em_iters += (double)it;
        double *const x = (it & 1) ? q->x1.p : q->x0.p;
        launch_np_sum(x, n_tx, 1000000.0, sums.p, total, q->stream);
        launch_divide(x, n_tx, total + 1, true, 0.001, q->stream);
        launch_np_sum(x, n_tx, 1000000.0, sums.p, total, q->stream);
        launch_divide(x, n_tx, total + 1, false, 0.0, q->stream);
        HIP_TRY(hipGetLastError());
        if (tpm) HIP_TRY(hipMemcpyAsync(tpm, x, n_tx * 8, hipMemcpyDeviceToHost, q->stream));
        if (effective_lengths)
            HIP_TRY(hipMemcpyAsync(effective_lengths, q->eff_len.p, n_tx * 8, hipMemcpyDeviceToHost, q->stream));
        HIP_TRY(hipStreamSynchronize(q->stream));
        lap("tpm");
        if (iters) *iters = it;
        return SKM_OK;
    };
    rc = body();
    fld.release();
    sums.release();
    skm_quant_destroy(q);
    lap("destroy");
    return rc;
}

extern "C" int skm_quant_destroy(skm_quant *q)
{
    if (!q) return SKM_OK;
    STALE_CHECK("quant_destroy entry");
    (void)hipSetDevice(q->device);
    (void)hipStreamSynchronize(q->stream);
    q->cls_offset.release(); q->row_start.release(); q->tx_row.release(); q->ids.release();
    q->tx_cls.release(); q->row_tx.release(); q->perm.release(); q->cls_count.release(); q->cls_count_saved.release();
    q->inner.release(); q->row_sum.release(); q->eff_len.release(); q->x0.release(); q->x1.release();
    q->acc.release(); q->part_max.release(); q->part_flags.release(); q->arrivals.release(); q->ctl.release();
    q->cum.release(); q->tile_total.release(); q->x_start.release(); q->boot_out.release();
    q->batch.cls_count.release(); q->batch.inner.release(); q->batch.row_sum.release(); q->batch.x0.release();
    q->batch.x1.release(); q->batch.part_max.release(); q->batch.part_flags.release(); q->batch.ctl.release();
    q->batch.mgr.release(); q->batch.counts_all.release(); q->batch.iters.release();
    for (auto &e : q->ev) pool_event_release(e, true);
    for (auto &e : q->chunk_ev) pool_event_release(e, false);
    pool_pinned_release(q->pinned);
    pool_stream_release(q->stream);
    delete q;
    STALE_CHECK("quant_destroy exit");
    return SKM_OK;
}

extern "C" int skm_quant_em(skm_quant *q, double *x, const double *l, double rel_tol, double x_floor,
                            int64_t max_iters, int64_t fixed_iters, int64_t *iters)
{
    if (!q || !x || !l) return fail(SKM_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> lock(q->mu);
    SKM_TRY(set_device(q->device));
    HIP_TRY(hipMemcpyAsync(q->x0.p, x, q->n_tx * 8, hipMemcpyHostToDevice, q->stream));
    HIP_TRY(hipMemcpyAsync(q->eff_len.p, l, q->n_tx * 8, hipMemcpyHostToDevice, q->stream));
    int64_t it = 0;
    SKM_TRY(em_run(q, rel_tol, x_floor, max_iters, fixed_iters, &it));
    HIP_TRY(hipMemcpy(x, (it & 1) ? q->x1.p : q->x0.p, q->n_tx * 8, hipMemcpyDeviceToHost));
    if (iters) *iters = it;
    return SKM_OK;
}

extern "C" int skm_quant_set_counts(skm_quant *q, const double *class_counts)
{
    if (!q || (!class_counts && q->n_classes)) return fail(SKM_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> lock(q->mu);
    SKM_TRY(set_device(q->device));
    const double total = np_sum_host(class_counts, q->n_classes);
    if (q->n_classes) {
        // caller's class order -> internal (locality) order
        SKM_TRY(q->cls_count_saved.ensure(q->n_classes));
        HIP_TRY(hipMemcpyAsync(q->cls_count_saved.p, class_counts, q->n_classes * 8, hipMemcpyHostToDevice, q->stream));
        launch_permute_f64(q->cls_count_saved.p, q->perm.p, q->n_classes, q->cls_count.p, false, q->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(q->stream));
    }
    q->n_total = total;
    return SKM_OK;
}

namespace {
// Replicate b of the call (b = 0 .. n_boot - 1) is replicate number rep_first + b * rep_step of the
// `-b N` run: its draw depends on (seed, that number) alone, so a rank's share of the replicates
// gives the same results as the one-GPU loop, replicate by replicate.
int bootstrap_impl(skm_quant *q, int64_t n_boot, uint64_t seed, const double *x0,
                   const double *l, double rel_tol, double x_floor, int64_t max_iters,
                   double *out, int64_t *counts_out, int64_t *iters_out, bool tpm,
                   int64_t rep_first = 0, int64_t rep_step = 1)
{
    if (!q || !x0 || !l || !out || n_boot < 0 || rep_first < 0 || rep_step < 1) return fail(SKM_ERR_ARG, "bad argument");
    std::lock_guard<std::mutex> lock(q->mu);
    SKM_TRY(set_device(q->device));
    // (seekmer/infer.py:108-111 resamples the table of the WHOLE sample: a handle that holds one
    // rank's share of the classes -- a communicator of several ranks attached -- cannot)
    if (q->comm && q->world > 1)
        return fail(SKM_ERR_STATE, "bootstraps resample the merged class table: detach the communicator "
                                   "(skm_quant_set_comm(quant, NULL)) and give every rank the merged table");
    auto number = [&](int64_t b) { return (uint64_t)(rep_first + b * rep_step); };
    const int64_t C = q->n_classes;
    if (C == 0) return fail(SKM_ERR_STATE, "no classes to resample");
    // integer cumulative counts of the observed table
    SKM_TRY(q->cls_count_saved.ensure(C));
    SKM_TRY(q->cum.ensure(C));
    SKM_TRY(q->tile_total.ensure(4096));
    HIP_TRY(hipMemcpyAsync(q->cls_count_saved.p, q->cls_count.p, C * 8, hipMemcpyDeviceToDevice, q->stream));
    std::vector<double> cnt(C);
    HIP_TRY(hipMemcpyAsync(cnt.data(), q->cls_count.p, C * 8, hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipStreamSynchronize(q->stream));
    std::vector<unsigned long long> cum(C);
    unsigned long long run = 0;
    for (int64_t c = 0; c < C; ++c) { run += (unsigned long long)cnt[c]; cum[c] = run; }
    if (run >= (1ULL << 32)) return fail(SKM_ERR_STATE, "more than 2^32 - 1 units to resample");
    const int64_t T = q->n_tx;
    // the replicates' results stay in HBM and come back in groups (one copy per group, not one per replicate)
    const int64_t group = std::max<int64_t>(1, std::min<int64_t>(n_boot, (int64_t)(1LL << 28) / T));
    SKM_TRY(q->x_start.ensure(T));
    SKM_TRY(q->boot_out.ensure((size_t)(group * T)));
    HIP_TRY(hipMemcpyAsync(q->cum.p, cum.data(), C * 8, hipMemcpyHostToDevice, q->stream));
    HIP_TRY(hipMemcpyAsync(q->eff_len.p, l, T * 8, hipMemcpyHostToDevice, q->stream));
    HIP_TRY(hipMemcpyAsync(q->x_start.p, x0, T * 8, hipMemcpyHostToDevice, q->stream));
    const double saved_total = q->n_total;
    const int64_t n_draws = (int64_t)run;            // n = class_count.sum(), infer.py:109
    int rc = SKM_OK;
    auto restore = on_exit([&]() {                   // every exit: the handle holds the observed counts again
        (void)hipMemcpyAsync(q->cls_count.p, q->cls_count_saved.p, C * 8, hipMemcpyDeviceToDevice, q->stream);
        (void)hipStreamSynchronize(q->stream);
        q->n_total = saved_total;
    });
    // One replicate the careful way (host-checked chunks of the single-problem EM): draw, EM from
    // x_start, result to `dst` (HBM).
    auto replicate_checked = [&](int64_t b, int64_t *it_out, double *dst) -> int {
        if (!launch_multinomial(q->cum.p, C, n_draws, seed, number(b), q->tile_total.p, q->cls_count.p, 1, q->stream))
            return fail(SKM_ERR_STATE, "class table too large to resample (%lld classes)", (long long)C);
        HIP_TRY(hipGetLastError());
        if (counts_out) {
            // internal (locality) class order -> caller's order; the counts fit a double exactly
            SKM_TRY(q->inner.ensure(C));
            launch_permute_f64(q->cls_count.p, q->perm.p, C, q->inner.p, true, q->stream);
            std::vector<double> as_double(C);
            HIP_TRY(hipMemcpyAsync(as_double.data(), q->inner.p, C * 8, hipMemcpyDeviceToHost, q->stream));
            HIP_TRY(hipStreamSynchronize(q->stream));
            for (int64_t c = 0; c < C; ++c) counts_out[b * C + c] = (int64_t)as_double[c];
        }
        HIP_TRY(hipMemcpyAsync(q->x0.p, q->x_start.p, T * 8, hipMemcpyDeviceToDevice, q->stream));
        q->n_total = (double)n_draws;
        SKM_TRY(em_run(q, rel_tol, x_floor, max_iters, 0, it_out, 8));
        HIP_TRY(hipMemcpyAsync(dst, (*it_out & 1) ? q->x1.p : q->x0.p, T * 8, hipMemcpyDeviceToDevice, q->stream));
        if (iters_out) iters_out[b] = *it_out;
        return SKM_OK;
    };
    DBuf<double> sums;
    auto drop_sums = on_exit([&]() { sums.release(); });
    // infer.py:127-129 on `count` results in HBM (TPM scaling with numpy's sums), when TPM is asked for
    auto scale = [&](double *results, int64_t count) -> int {
        if (!tpm || count <= 0) return SKM_OK;
        const int64_t n_blocks = (T + 8191) / 8192;
        SKM_TRY(sums.ensure((size_t)(count * (n_blocks + 2))));
        double *const totals = sums.p + count * n_blocks;            // (sum, sum / 1e6) per replicate
        launch_np_sum_many(results, T, count, T, 1000000.0, sums.p, totals, q->stream);
        launch_divide_many(results, T, count, T, totals + 1, true, 0.001, q->stream);
        launch_np_sum_many(results, T, count, T, 1000000.0, sums.p, totals, q->stream);
        launch_divide_many(results, T, count, T, totals + 1, false, 0.0, q->stream);
        HIP_TRY(hipGetLastError());
        return SKM_OK;
    };
    auto send_home = [&](int64_t first, int64_t count) -> int {     // boot_out[0 .. count) = replicates first ..
        if (count <= 0) return SKM_OK;
        SKM_TRY(scale(q->boot_out.p, count));
        HIP_TRY(hipMemcpyAsync(out + first * T, q->boot_out.p, (size_t)count * T * 8, hipMemcpyDeviceToHost, q->stream));
        HIP_TRY(hipStreamSynchronize(q->stream));
        return SKM_OK;
    };
    // With the resampled counts wanted, a step cap set or a communicator attached (its collectives
    // must stay matched) every replicate goes the careful way.  Otherwise EM_BATCH replicates sit side
    // by side in the batched EM (skm_em_batch.hip) and the working set is kept full: steps are queued
    // in short chunks; after each chunk the host reads which replicates have latched their stopping
    // rule, takes their results, and puts the next replicates (fresh draw, the common start vector)
    // in their places -- the others run on undisturbed.  Step counts have a long tail (most
    // replicates of the 20 M-pair table stop near 30 steps, one in six needs 60-90): replicates
    // that wait for the slowest of a fixed group of eight waste half the working set's steps.
    // Every replicate still runs the single-problem EM's steps bit for bit, whoever its neighbours are.
    const bool batched = !counts_out && !q->comm && max_iters <= 0;
    if (!batched) {
        for (int64_t b = 0; b < n_boot; ++b) {
            int64_t it = 0;
            SKM_TRY(replicate_checked(b, &it, q->boot_out.p));
            SKM_TRY(send_home(b, 1));
        }
        return rc;
    }
    // EM_BATCH replicates sit side by side in the batched EM (skm_em_batch.hip) and THE DEVICE keeps the
    // working set full: the class counts of a whole group of replicates are drawn first (HBM has the
    // room: 8 MB per replicate at a million classes), and after every step two small launches take
    // the results of the replicates that have latched their stopping rule and put the next ones in
    // their places (launch_em_batch_manage) -- no host look between steps; the host queues steps
    // and reads now and then how many replicates have finished.  (Through round 3's first half the
    // host looked every four steps and refilled: ~100 looks and a fifth of the phase's wall time in
    // gaps.)  Step counts have a long tail (most replicates of the 20 M-pair table stop near 30
    // steps, one in six needs 60-90); every replicate still runs the single-problem EM's steps bit
    // for bit, whoever its neighbours are and whenever it starts.
    const int64_t by_counts = std::max<int64_t>(1, (int64_t)((1LL << 31) / (8 * std::max<int64_t>(C, 1))));
    const int64_t slots = std::max<int64_t>(1, std::min(group, by_counts));
    SKM_TRY(q->boot_out.ensure((size_t)(slots * T)));
    skm_quant::Batch &w = q->batch;
    SKM_TRY(w.cls_count.ensure((size_t)C * EM_BATCH)); SKM_TRY(w.inner.ensure((size_t)C * EM_BATCH));
    SKM_TRY(w.row_sum.ensure((size_t)std::max<int64_t>(q->n_rows, 1) * EM_BATCH));
    SKM_TRY(w.x0.ensure((size_t)T * EM_BATCH)); SKM_TRY(w.x1.ensure((size_t)T * EM_BATCH));
    SKM_TRY(w.part_max.ensure((size_t)EM_FINAL_BLOCKS * EM_BATCH));
    SKM_TRY(w.part_flags.ensure((size_t)EM_FINAL_BLOCKS * EM_BATCH));
    SKM_TRY(w.ctl.ensure(32));
    SKM_TRY(w.mgr.ensure(64));
    SKM_TRY(w.counts_all.ensure((size_t)slots * C));
    SKM_TRY(w.iters.ensure((size_t)slots));
    EmBatchProblem p{};
    p.n_tx = T; p.n_classes = C; p.n_rows = q->n_rows;
    p.cls_offset = q->cls_offset.p; p.ids = q->ids.p; p.row_start = q->row_start.p; p.row_tx = q->row_tx.p;
    p.tx_cls = q->tx_cls.p; p.tx_row = q->tx_row.p; p.eff_len = q->eff_len.p;
    p.cls_count = w.cls_count.p; p.inner = w.inner.p; p.row_sum = w.row_sum.p;
    p.x[0] = w.x0.p; p.x[1] = w.x1.p;
    p.n_total = (double)n_draws; p.rel_tol = rel_tol; p.x_floor = x_floor;
    p.ctl = w.ctl.p; p.part_max = w.part_max.p; p.part_flags = w.part_flags.p;
    p.managed = 1;
    p.mgr = w.mgr.p;
    p.iters_out = w.iters.p;
    p.fused = getenv("SKM_EM_UNFUSED") ? 0 : 1;
    p.arrivals = q->arrivals.p;
    int64_t chunk = 16;                                          // steps queued between two looks at the progress
    if (const char *e = getenv("SKM_BOOTSTRAP_CHUNK")) chunk = std::max<int64_t>(1, atoll(e));   // (tests)
    unsigned long long *const look = q->pinned + 64;             // 64 + 8 words of the pinned block
    std::vector<int64_t> iters_host((size_t)slots);
    for (int64_t w0 = 0; w0 < n_boot; w0 += slots) {
        const int64_t n = std::min(n_boot, w0 + slots) - w0;
        for (int64_t i = 0; i < n; ++i)
            if (!launch_multinomial(q->cum.p, C, n_draws, seed, number(w0 + i), q->tile_total.p,
                                    w.counts_all.p + (size_t)i * C, 1, q->stream))
                return fail(SKM_ERR_STATE, "class table too large to resample (%lld classes)", (long long)C);
        launch_em_batch_manage_init(p, w.mgr.p, look, n, w.counts_all.p, q->x_start.p, q->boot_out.p, q->stream);
        HIP_TRY(hipGetLastError());
        for (int64_t k = 0;;) {
            for (int64_t i = 0; i < chunk; ++i, ++k) {
                launch_em_batch_step(p, k, q->stream);        // (its first kernel also plans for what has stopped)
                launch_em_batch_manage(p, w.mgr.p, w.counts_all.p, q->x_start.p, q->boot_out.p, w.iters.p, k, true, q->stream);
            }
            // before the host looks, the last pass is judged too (otherwise only the next step's first
            // kernel would) and what it stops is taken: a place that is still occupied then is running
            launch_em_batch_decide(p, k, q->stream);
            launch_em_batch_manage(p, w.mgr.p, w.counts_all.p, q->x_start.p, q->boot_out.p, w.iters.p, k - 1, false, q->stream);
            q->launches += 4 * chunk + 3;
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(look, w.mgr.p, 24 * 8, hipMemcpyDeviceToHost, q->stream));
            HIP_TRY(hipMemcpyAsync(look + 24, w.ctl.p, 8 * 8, hipMemcpyDeviceToHost, q->stream));
            HIP_TRY(hipStreamSynchronize(q->stream));
            if (look[3])
                return fail(SKM_ERR_UNDEFINED, "no abundance above x_floor: numpy raises on max() of an empty selection");
            if (look[24]) break;                                 // every replicate of the group has finished
            if (k > (1LL << 24)) return fail(SKM_ERR_STATE, "the bootstrap EM does not stop");
            // The tail: nothing left to put in, a few replicates still running.  A step of the
            // working set costs the same however many places are live (~5 single-problem steps), so
            // the last three or fewer go on one by one in the single-problem EM, from where they are.
            int live = 0;
            for (int r = 0; r < EM_BATCH; ++r) live += look[8 + r] != 0;
            if (look[0] >= look[1] && live <= 3) {
                for (int r = 0; r < EM_BATCH; ++r) {
                    if (look[8 + r] == 0) continue;
                    const int64_t rep = (int64_t)look[8 + r] - 1, since = (int64_t)look[16 + r];
                    launch_em_batch_take(p.x[k & 1], T, r, q->x0.p, q->stream);
                    launch_em_batch_take(w.cls_count.p, C, r, q->cls_count.p, q->stream);
                    HIP_TRY(hipGetLastError());
                    q->n_total = (double)n_draws;
                    int64_t more = 0;
                    SKM_TRY(em_run(q, rel_tol, x_floor, max_iters, 0, &more, 8));
                    HIP_TRY(hipMemcpyAsync(q->boot_out.p + rep * T, (more & 1) ? q->x1.p : q->x0.p, (size_t)T * 8,
                                           hipMemcpyDeviceToDevice, q->stream));
                    const int64_t steps = k - since + more;
                    HIP_TRY(hipMemcpyAsync(w.iters.p + rep, &steps, 8, hipMemcpyHostToDevice, q->stream));
                    HIP_TRY(hipStreamSynchronize(q->stream));      // (`steps` lives on this frame)
                    q->iters_total -= (double)more;                // (em_run has counted its own; the sum below counts all)
                }
                break;
            }
        }
        HIP_TRY(hipMemcpyAsync(iters_host.data(), w.iters.p, (size_t)n * 8, hipMemcpyDeviceToHost, q->stream));
        HIP_TRY(hipStreamSynchronize(q->stream));
        for (int64_t i = 0; i < n; ++i) {
            if (iters_out) iters_out[w0 + i] = iters_host[(size_t)i];
            q->iters_total += (double)iters_host[(size_t)i];
        }
        SKM_TRY(send_home(w0, n));
    }
    return rc;
}

}  // namespace

extern "C" int skm_quant_bootstrap(skm_quant *q, int64_t n_boot, uint64_t seed, const double *x0,
                                   const double *l, double rel_tol, double x_floor, int64_t max_iters,
                                   double *out, int64_t *counts_out, int64_t *iters_out)
{
    return bootstrap_impl(q, n_boot, seed, x0, l, rel_tol, x_floor, max_iters, out, counts_out, iters_out, false);
}

extern "C" int skm_quant_bootstrap_tpm(skm_quant *q, int64_t n_boot, uint64_t seed, const double *x0,
                                       const double *l, double rel_tol, double x_floor, int64_t max_iters,
                                       double *out, int64_t *iters_out)
{
    return bootstrap_impl(q, n_boot, seed, x0, l, rel_tol, x_floor, max_iters, out, nullptr, iters_out, true);
}

extern "C" int skm_quant_bootstrap_share_tpm(skm_quant *q, int64_t n_boot, int64_t first, int64_t step, uint64_t seed,
                                             const double *x0, const double *l, double rel_tol, double x_floor,
                                             int64_t max_iters, double *out, int64_t *iters_out)
{
    return bootstrap_impl(q, n_boot, seed, x0, l, rel_tol, x_floor, max_iters, out, nullptr, iters_out, true, first, step);
}

extern "C" int skm_quant_timing(skm_quant *q, double timing[4])
{
    if (!q || !timing) return fail(SKM_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> lock(q->mu);
    timing[0] = q->t_em_ns; timing[1] = q->iters_total; timing[2] = q->launches; timing[3] = 0;
    return SKM_OK;
}

// ---------------------------------------------------------------- multi-GPU
extern "C" int skm_comm_unique_id(void *id128)
{
    if (!id128) return fail(SKM_ERR_ARG, "NULL argument");
    SKM_TRY(load_rccl());
    NCCL_TRY(g_rccl.GetUniqueId(id128));
    return SKM_OK;
}

extern "C" int skm_comm_create(int device, const void *id128, int rank, int world, skm_comm **out)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(SKM_ERR_ARG, "bad argument");
    int n_dev = 0;
    SKM_TRY(skm_device_count(&n_dev));
    if (device < 0 || device >= n_dev) return fail(SKM_ERR_ARG, "device %d out of range", device);
    SKM_TRY(set_device(device));
    SKM_TRY(load_rccl());
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    skm_comm *c = new skm_comm();
    c->device = device;
    c->rank = rank;
    c->world = world;
    int r = g_comm_init(&c->comm, world, id, rank);
    if (r != 0) {
        delete c;
        return fail(SKM_ERR_COMM, "ncclCommInitRank failed: %s",
                    g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    }
    *out = c;
    return SKM_OK;
}

extern "C" int skm_comm_count(skm_comm *c, int *count)
{
    if (!c || !count) return fail(SKM_ERR_ARG, "NULL argument");
    if (!g_rccl.CommCount) return fail(SKM_ERR_COMM, "librccl.so lacks ncclCommCount");
    NCCL_TRY(g_rccl.CommCount(c->comm, count));
    return SKM_OK;
}

// SURVEY 8(e).1 over xGMI: a mapper's table goes from GPU to GPU as it lies in HBM
// (skm_mapper_device_table's arrays, ncclSend / ncclRecv) and is merged by key on the receiving GPU
// (class_merge_kernel over the received arrays): no host copy, no host sort.  One call does both
// directions so that a rank may pair a send with a receive (two ranks swapping, or -- the one-GPU
// test -- a rank sending to itself): first the sizes (a header of eight words), then the arrays.
extern "C" int skm_mapper_exchange_tables(skm_mapper *send, int send_to, skm_mapper *recv, int recv_from, skm_comm *comm)
{
    if (!comm || (!send && !recv)) return fail(SKM_ERR_ARG, "NULL argument");
    if ((send && send_to < 0) || (recv && recv_from < 0) || send_to >= comm->world || recv_from >= comm->world)
        return fail(SKM_ERR_ARG, "peer rank outside the communicator");
    if ((send && send->ix->device != comm->device) || (recv && recv->ix->device != comm->device))
        return fail(SKM_ERR_ARG, "communicator and mapper live on different GPUs");
    SKM_TRY(load_rccl());
    if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd)
        return fail(SKM_ERR_COMM, "librccl.so lacks ncclSend / ncclRecv");
    SKM_TRY(set_device(comm->device));
    skm_device_table out{};
    if (send) SKM_TRY(skm_mapper_device_table(send, &out));            // (waits for what the mapper has queued)
    hipStream_t stream = nullptr;
    HIP_TRY(pool_stream_acquire(&stream));
    DBuf<unsigned long long> header, fld, first_seen;
    DBuf<int64_t> start, len;
    DBuf<double> count;
    DBuf<int32_t> ids;
    auto undo = on_exit([&]() {
        (void)hipStreamSynchronize(stream);
        header.release(); fld.release(); first_seen.release(); start.release(); len.release(); count.release(); ids.release();
        pool_stream_release(stream);
    });
    SKM_TRY(header.ensure(16));
    unsigned long long words[16] = {(unsigned long long)out.n_classes, (unsigned long long)out.n_ids,
                                    (unsigned long long)out.unaligned, (unsigned long long)out.units,
                                    (unsigned long long)out.first_seen_bound, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(header.p, words, 8 * 8, hipMemcpyHostToDevice, stream));
    // (a group that has been started is always ended, whatever a call inside it returned)
    auto grouped = [&](const std::function<int()> &calls) -> int {
        NCCL_TRY(g_rccl.GroupStart());
        const int rc = calls();
        const int end = g_rccl.GroupEnd();
        if (rc != SKM_OK) return rc;
        NCCL_TRY(end);
        return SKM_OK;
    };
    SKM_TRY(grouped([&]() -> int {
        if (send) NCCL_TRY(g_rccl.Send(header.p, 8, NCCL_UINT64, send_to, comm->comm, stream));
        if (recv) NCCL_TRY(g_rccl.Recv(header.p + 8, 8, NCCL_UINT64, recv_from, comm->comm, stream));
        return SKM_OK;
    }));
    HIP_TRY(hipMemcpyAsync(words + 8, header.p + 8, 8 * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    skm_device_table in{};
    in.device = comm->device;
    if (recv) {
        in.n_classes = (int64_t)words[8]; in.n_ids = (int64_t)words[9];
        in.unaligned = (int64_t)words[10]; in.units = (int64_t)words[11]; in.first_seen_bound = (int64_t)words[12];
        if (in.n_classes < 0 || in.n_ids < 0 || in.n_classes >= (1LL << 40) || in.n_ids >= (1LL << 40))
            return fail(SKM_ERR_COMM, "received a table header that makes no sense");
        SKM_TRY(start.ensure(std::max<int64_t>(in.n_classes, 1))); SKM_TRY(len.ensure(std::max<int64_t>(in.n_classes, 1)));
        SKM_TRY(count.ensure(std::max<int64_t>(in.n_classes, 1))); SKM_TRY(first_seen.ensure(std::max<int64_t>(in.n_classes, 1)));
        SKM_TRY(ids.ensure(std::max<int64_t>(in.n_ids, 1))); SKM_TRY(fld.ensure(MAX_FRAGMENT_LENGTH));
    }
    SKM_TRY(grouped([&]() -> int {
    if (send) {
        if (out.n_classes) {
            NCCL_TRY(g_rccl.Send(out.class_start, (size_t)out.n_classes, NCCL_UINT64, send_to, comm->comm, stream));
            NCCL_TRY(g_rccl.Send(out.class_len, (size_t)out.n_classes, NCCL_UINT64, send_to, comm->comm, stream));
            NCCL_TRY(g_rccl.Send(out.class_count, (size_t)out.n_classes, NCCL_FLOAT64, send_to, comm->comm, stream));
            NCCL_TRY(g_rccl.Send(out.first_seen, (size_t)out.n_classes, NCCL_UINT64, send_to, comm->comm, stream));
        }
        if (out.n_ids) NCCL_TRY(g_rccl.Send(out.ids, (size_t)out.n_ids, NCCL_INT32, send_to, comm->comm, stream));
        NCCL_TRY(g_rccl.Send(out.fld, MAX_FRAGMENT_LENGTH, NCCL_UINT64, send_to, comm->comm, stream));
    }
    if (recv) {
        if (in.n_classes) {
            NCCL_TRY(g_rccl.Recv(start.p, (size_t)in.n_classes, NCCL_UINT64, recv_from, comm->comm, stream));
            NCCL_TRY(g_rccl.Recv(len.p, (size_t)in.n_classes, NCCL_UINT64, recv_from, comm->comm, stream));
            NCCL_TRY(g_rccl.Recv(count.p, (size_t)in.n_classes, NCCL_FLOAT64, recv_from, comm->comm, stream));
            NCCL_TRY(g_rccl.Recv(first_seen.p, (size_t)in.n_classes, NCCL_UINT64, recv_from, comm->comm, stream));
        }
        if (in.n_ids) NCCL_TRY(g_rccl.Recv(ids.p, (size_t)in.n_ids, NCCL_INT32, recv_from, comm->comm, stream));
        NCCL_TRY(g_rccl.Recv(fld.p, MAX_FRAGMENT_LENGTH, NCCL_UINT64, recv_from, comm->comm, stream));
    }
    return SKM_OK;
    }));
    HIP_TRY(hipStreamSynchronize(stream));
    if (!recv) return SKM_OK;
    in.class_start = start.p; in.class_len = len.p; in.class_count = count.p;
    in.first_seen = (const uint64_t *)first_seen.p; in.ids = ids.p; in.fld = (const uint64_t *)fld.p;
    return skm_mapper_merge_device(recv, &in);
}

extern "C" int skm_comm_destroy(skm_comm *c)
{
    if (!c) return SKM_OK;
    (void)hipSetDevice(c->device);
    if (c->comm && g_rccl.CommDestroy) NCCL_TRY(g_rccl.CommDestroy(c->comm));
    delete c;
    return SKM_OK;
}

extern "C" int skm_quant_set_comm(skm_quant *q, skm_comm *c)
{
    if (!q) return fail(SKM_ERR_ARG, "NULL quant");
    std::lock_guard<std::mutex> lock(q->mu);
    if (c && c->device != q->device) return fail(SKM_ERR_ARG, "communicator and quant live on different GPUs");
    q->comm = c ? c->comm : nullptr;
    q->rank = c ? c->rank : 0;
    q->world = c ? c->world : 1;
    return SKM_OK;
}
