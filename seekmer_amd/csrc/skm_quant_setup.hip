// Device-side construction of the EM problem: the class-major CSR and its
// transcript-major transpose cut into rows.  The caller's class order is the
// reference's (ascending first-seen unit = collections.Counter insertion order
// under -j1, /root/reference/seekmer/mapper.py:88); internally the classes are
// kept in (smallest transcript id, first-seen) order for gather locality, with
// perm[] leading back to the caller's order.
//
// From a mapper's table the first-seen RANK of every class comes from a bitmap
// over the unit indices (first-seen values are distinct: a unit belongs to one
// class) and a prefix popcount -- no sort; one stable 18-bit radix sort of the
// classes by smallest id and one of the pairs by transcript remain (rocPRIM's
// onesweep), everything else is small index-moving kernels.  One-time setup per
// quantification, not part of the per-step hot loop.
#include "skm_kernels.h"
#include "skm_pool.h"

#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>
#include <vector>

namespace skm {

namespace {

// Temporaries of one setup call: handed out from the caching pool, given back
// together after ONE stream synchronisation at the end (the setup is a single
// asynchronous pipeline; nothing in between needs the host).
struct Scratch {
    hipStream_t stream;
    std::vector<void *> held;
    explicit Scratch(hipStream_t s) : stream(s) {}
    template <class T>
    T *alloc(size_t n)
    {
        void *p = nullptr;
        if (pool_alloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        held.push_back(p);
        return static_cast<T *>(p);
    }
    ~Scratch()
    {
        (void)hipStreamSynchronize(stream);
        for (void *p : held) pool_free(p);
    }
};
// what failed last, for the caller's message (hipGetLastError is consumed by the check itself)
thread_local char g_setup_failure[160] = "";
#define QB_ALLOC(var, type, n) type *var = scratch.alloc<type>(n); \
    if (!var) { snprintf(g_setup_failure, sizeof(g_setup_failure), "no memory for %s", #var); return -1; }

#define QB_TRY(call) do { const hipError_t qb_e = (call); if (qb_e != hipSuccess) { \
    snprintf(g_setup_failure, sizeof(g_setup_failure), "%s: %s", #call, hipGetErrorString(qb_e)); return -1; } } while (0)

__global__ void __launch_bounds__(256)
iota_kernel(int32_t *out, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) out[i] = (int32_t)i;
}

// dump of the table in registry order: arena offset, length, count, first-seen per class
// (+ the class's smallest transcript id, the locality key, when asked for)
__global__ void __launch_bounds__(256)
table_dump_kernel(ClassTable t, int64_t n_classes, int64_t *arena_off, int64_t *len, double *count,
                  unsigned long long *first_seen, uint32_t *min_id)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_classes;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t.class_list[k];
        const ClassSlot s = t.slots[i];
        const int64_t off = s.tuple < 0 ? -1 : tuple_offset(s.tuple);
        const int64_t n = s.tuple < 0 ? 0 : tuple_len(s.tuple);
        arena_off[k] = off;
        len[k] = n;
        count[k] = (double)s.count;
        first_seen[k] = s.first_seen;
        if (min_id) {
            uint32_t m = 0xffffffffu;
            for (int64_t j = 0; j < n; ++j) m = min(m, (uint32_t)t.arena[off + j]);
            min_id[k] = m;
        }
    }
}

// ---- first-seen ranks from a bitmap over the unit indices
// one bit per unit index; a bit found set already means two classes share a first-seen value
// (possible only for tables merged from hand-made input): *duplicate is raised and the caller
// takes the sorting path instead
__global__ void __launch_bounds__(256)
mark_first_seen_kernel(const unsigned long long *first_seen, int64_t n_classes, uint64_t bound,
                       uint32_t *bits, unsigned int *duplicate)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_classes;
         k += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long f = first_seen[k];
        if (f >= bound) { atomicOr(duplicate, 2u); continue; }
        const uint32_t bit = 1u << (f & 31);
        if (atomicOr(&bits[f >> 5], bit) & bit) atomicOr(duplicate, 1u);
    }
}

__global__ void __launch_bounds__(256)
word_popcount_kernel(const uint32_t *bits, int64_t n_words, int64_t *count)
{
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n_words;
         w += (int64_t)gridDim.x * blockDim.x) count[w] = __popc(bits[w]);
}

// rank r of class k = set bits below its own; by_rank[r] = k, key_by_rank[r] = its locality key
__global__ void __launch_bounds__(256)
first_seen_rank_kernel(const unsigned long long *first_seen, const uint32_t *min_id, int64_t n_classes,
                       uint64_t bound, const uint32_t *bits, const int64_t *word_base, int32_t *by_rank,
                       uint32_t *key_by_rank)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_classes;
         k += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long f = first_seen[k];
        if (f >= bound) continue;
        const int64_t r = word_base[f >> 5] + __popc(bits[f >> 5] & ((1u << (f & 31)) - 1u));
        if (r >= n_classes) continue;             // (only after a duplicate; the result is discarded)
        by_rank[r] = (int32_t)k;
        key_by_rank[r] = min_id[k];
    }
}

// internal class j = the class of first-seen rank perm[j] = registry entry by_rank[perm[j]]
__global__ void __launch_bounds__(256)
gather_by_rank_kernel(const int32_t *perm, const int32_t *by_rank, int64_t n, const int64_t *len_in,
                      const double *count_in, const int64_t *arena_off_in, int64_t *len_out,
                      double *count_out, int64_t *arena_off_out)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        const int32_t k = by_rank[perm[j]];
        len_out[j] = len_in[k];
        count_out[j] = count_in[k];
        arena_off_out[j] = arena_off_in[k];
    }
}

// ids of internal class j <- arena[src_off[j] ...]
__global__ void __launch_bounds__(256)
copy_tuples_direct_kernel(int64_t n, const int64_t *src_off, const int32_t *arena, const int64_t *cls_offset,
                          int32_t *ids)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = src_off[j];
        const int64_t dst = cls_offset[j];
        const int64_t len = cls_offset[j + 1] - dst;
        for (int64_t i = 0; i < len; ++i) ids[dst + i] = arena[src + i];
    }
}

__global__ void __launch_bounds__(256)
gather_classes_kernel(const int32_t *perm, int64_t n, const int64_t *len_in, const double *count_in,
                      int64_t *len_out, double *count_out)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t src = perm[k];
        len_out[k] = len_in[src];
        count_out[k] = count_in[src];
    }
}

// copy every class's tuple from the arena to its place in class order
__global__ void __launch_bounds__(256)
copy_tuples_kernel(const int32_t *perm, int64_t n, const int64_t *arena_off, const int32_t *arena,
                   const int64_t *cls_offset, int32_t *ids)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = arena_off[perm[k]];
        const int64_t dst = cls_offset[k];
        const int64_t len = cls_offset[k + 1] - dst;
        for (int64_t j = 0; j < len; ++j) ids[dst + j] = arena[src + j];
    }
}

__global__ void __launch_bounds__(256)
set_last_offset_kernel(int64_t *offsets, const int64_t *lens, int64_t n)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n] = n ? offsets[n - 1] + lens[n - 1] : 0;
}

// class index of every pair, class-major
__global__ void __launch_bounds__(256)
pair_class_kernel(const int64_t *cls_offset, int64_t n_classes, int32_t *pair_cls)
{
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n_classes;
         c += (int64_t)gridDim.x * blockDim.x)
        for (int64_t j = cls_offset[c]; j < cls_offset[c + 1]; ++j) pair_cls[j] = (int32_t)c;
}

// tx_offset[t] = first position of transcript t in the key-sorted pair list
__global__ void __launch_bounds__(256)
segment_starts_kernel(const int32_t *sorted_tx, int64_t n_pairs, int64_t n_tx, int64_t *tx_offset)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j <= n_pairs;
         j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lo = j == 0 ? -1 : sorted_tx[j - 1];
        const int64_t hi = j == n_pairs ? n_tx : sorted_tx[j];
        for (int64_t t = lo + 1; t <= hi; ++t) tx_offset[t] = j;
    }
}

__global__ void __launch_bounds__(256)
row_count_kernel(const int64_t *tx_offset, int64_t n_tx, int64_t *rows)
{
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_tx;
         t += (int64_t)gridDim.x * blockDim.x) {
        // (at least one row, an empty one for a transcript no class names: the launch that sums the
        // rows is also the one that finalizes the transcripts, skm_em.hip: em_rows_finalize_kernel)
        const int64_t degree = tx_offset[t + 1] - tx_offset[t];
        rows[t] = degree > 0 ? (degree + EM_ROW_CAP - 1) / EM_ROW_CAP : 1;
    }
}

__global__ void __launch_bounds__(256)
row_fill_kernel(const int64_t *tx_offset, const int64_t *tx_row, int64_t n_tx, int64_t n_pairs,
                int64_t *row_start, int32_t *row_tx)
{
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_tx;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t pos = tx_offset[t];
        for (int64_t r = tx_row[t]; r < tx_row[t + 1]; ++r) {
            row_start[r] = pos;
            row_tx[r] = (int32_t)t;
            pos += EM_ROW_CAP;
        }
        if (t == n_tx - 1) row_start[tx_row[n_tx]] = n_pairs;
    }
}

// locality key of a class: its smallest transcript id
__global__ void __launch_bounds__(256)
class_min_id_kernel(const int64_t *cls_offset, const int32_t *ids, int64_t n_classes, uint32_t *key,
                    int64_t *len)
{
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n_classes;
         c += (int64_t)gridDim.x * blockDim.x) {
        uint32_t m = 0xffffffffu;
        for (int64_t j = cls_offset[c]; j < cls_offset[c + 1]; ++j) m = min(m, (uint32_t)ids[j]);
        key[c] = m;
        len[c] = cls_offset[c + 1] - cls_offset[c];
    }
}

__global__ void __launch_bounds__(256)
copy_tuples_by_offset_kernel(const int32_t *perm, int64_t n, const int64_t *src_offset, const int32_t *src,
                             const int64_t *dst_offset, int32_t *dst)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t from = src_offset[perm[k]];
        const int64_t to = dst_offset[k];
        const int64_t len = dst_offset[k + 1] - to;
        for (int64_t j = 0; j < len; ++j) dst[to + j] = src[from + j];
    }
}

inline unsigned blocks_for(int64_t n)
{
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)b;
}

// exclusive scan of n int64 values into out[0..n) and the total into out[n] (all on the stream)
int exclusive_scan_with_total(Scratch &scratch, const int64_t *in, int64_t *out, int64_t n)
{
    hipStream_t stream = scratch.stream;
    if (n == 0) {
        QB_TRY(hipMemsetAsync(out, 0, sizeof(int64_t), stream));
        return 0;
    }
    size_t bytes = 0;
    QB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, stream));
    QB_ALLOC(tmp, char, bytes);
    QB_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, in, out, (int)n, stream));
    hipLaunchKernelGGL(set_last_offset_kernel, dim3(1), dim3(64), 0, stream, out, in, n);
    return 0;
}

// Stable radix sort of (key, value) pairs on bits [0, end_bit).  Onesweep at every size: below
// 1 M items rocPRIM's default picks its merge sort, which at the 0.85 M classes of configs[1]
// takes ten merge rounds (155 us) where three onesweep passes take a third of that.
using OnesweepAlways = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                  rocprim::default_config, 0>;
template <class K>
int sort_pairs(Scratch &scratch, const K *keys_in, K *keys_out, const int32_t *vals_in, int32_t *vals_out,
               int64_t n, int end_bit)
{
    size_t bytes = 0;
    QB_TRY(rocprim::radix_sort_pairs<OnesweepAlways>(nullptr, bytes, keys_in, keys_out, vals_in, vals_out,
                                                     (size_t)n, 0u, (unsigned)end_bit, scratch.stream));
    QB_ALLOC(tmp, char, bytes);
    QB_TRY(rocprim::radix_sort_pairs<OnesweepAlways>(tmp, bytes, keys_in, keys_out, vals_in, vals_out,
                                                     (size_t)n, 0u, (unsigned)end_bit, scratch.stream));
    return 0;
}

}  // namespace

int64_t quant_rows_upper_bound(int64_t n_tx, int64_t n_ids)
{
    return n_tx + n_ids / EM_ROW_CAP + 1;
}

namespace {

int build_from_table(Scratch &scratch, const ClassTable &t, QuantBuild &q)
{
    hipStream_t stream = scratch.stream;
    const int64_t n_classes = q.n_classes;
    if (n_classes == 0) {
        QB_TRY(hipMemsetAsync(q.cls_offset, 0, sizeof(int64_t), stream));
        return 0;
    }
    if (n_classes >= (1LL << 31) || q.n_ids >= (1LL << 31)) return -2;
    QB_ALLOC(arena_off, int64_t, n_classes); QB_ALLOC(len, int64_t, n_classes);
    QB_ALLOC(len_sorted, int64_t, n_classes); QB_ALLOC(count, double, n_classes);
    QB_ALLOC(first, unsigned long long, n_classes); QB_ALLOC(first_sorted, unsigned long long, n_classes);
    QB_ALLOC(iota, int32_t, n_classes); QB_ALLOC(perm, int32_t, n_classes);
    hipLaunchKernelGGL(table_dump_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream,
                       t, n_classes, arena_off, len, count, first, (uint32_t *)nullptr);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream, iota, n_classes);
    int first_bits = 64;
    if (q.first_seen_bound > 0) {
        first_bits = 1;
        while (first_bits < 63 && (1LL << first_bits) <= q.first_seen_bound) ++first_bits;
    }
    if (sort_pairs(scratch, first, first_sorted, iota, perm, n_classes, first_bits)) return -1;
    hipLaunchKernelGGL(gather_classes_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream, perm,
                       n_classes, len, count, len_sorted, q.cls_count);
    if (exclusive_scan_with_total(scratch, len_sorted, q.cls_offset, n_classes)) return -1;
    hipLaunchKernelGGL(copy_tuples_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream, perm,
                       n_classes, arena_off, t.arena, q.cls_offset, q.ids);
    return 0;
}

// The same result as build_from_table + localize (classes by smallest transcript id, then by
// first-seen; perm[j] = first-seen rank of internal class j) with one sort instead of two and one
// copy of the tuples instead of two: the first-seen rank is a prefix popcount over a bitmap of
// the unit indices.  Returns 1 (nothing usable written) when two classes share a first-seen
// value or one lies outside the bound: the caller falls back to the sorting path.
constexpr int64_t RANK_BITMAP_MAX_UNITS = 1LL << 31;      // 256 MiB of bits + 4 GiB of word sums at most

int build_from_table_ranked(Scratch &scratch, const ClassTable &t, QuantBuild &q, int32_t *perm)
{
    hipStream_t stream = scratch.stream;
    const int64_t C = q.n_classes;
    if (C == 0) {
        QB_TRY(hipMemsetAsync(q.cls_offset, 0, sizeof(int64_t), stream));
        return 0;
    }
    if (C >= (1LL << 31) || q.n_ids >= (1LL << 31)) return -2;
    const uint64_t bound = (uint64_t)q.first_seen_bound;
    const int64_t n_words = (int64_t)((bound + 31) / 32);
    QB_ALLOC(arena_off, int64_t, C); QB_ALLOC(len, int64_t, C); QB_ALLOC(count, double, C);
    QB_ALLOC(first, unsigned long long, C); QB_ALLOC(min_id, uint32_t, C);
    QB_ALLOC(bits, uint32_t, n_words); QB_ALLOC(duplicate, unsigned int, 1);
    QB_TRY(hipMemsetAsync(bits, 0, n_words * sizeof(uint32_t), stream));
    QB_TRY(hipMemsetAsync(duplicate, 0, sizeof(unsigned int), stream));
    hipLaunchKernelGGL(table_dump_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, t, C, arena_off, len, count,
                       first, min_id);
    hipLaunchKernelGGL(mark_first_seen_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, first, C, bound, bits,
                       duplicate);
    unsigned int shared_first_seen = 0;
    QB_TRY(hipMemcpyAsync(&shared_first_seen, duplicate, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    QB_TRY(hipStreamSynchronize(stream));
    if (shared_first_seen) return 1;

    QB_ALLOC(word_count, int64_t, n_words); QB_ALLOC(word_base, int64_t, n_words + 1);
    QB_ALLOC(by_rank, int32_t, C); QB_ALLOC(key_by_rank, uint32_t, C); QB_ALLOC(key_sorted, uint32_t, C);
    QB_ALLOC(iota, int32_t, C); QB_ALLOC(len_sorted, int64_t, C); QB_ALLOC(src_off, int64_t, C);
    hipLaunchKernelGGL(word_popcount_kernel, dim3(blocks_for(n_words)), dim3(256), 0, stream, bits, n_words,
                       word_count);
    if (exclusive_scan_with_total(scratch, word_count, word_base, n_words)) return -1;
    hipLaunchKernelGGL(first_seen_rank_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, first, min_id, C, bound,
                       bits, word_base, by_rank, key_by_rank);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, iota, C);
    int end_bit = 1;
    while ((1LL << end_bit) < q.n_tx && end_bit < 32) ++end_bit;
    if (sort_pairs(scratch, key_by_rank, key_sorted, iota, perm, C, end_bit)) return -1;
    hipLaunchKernelGGL(gather_by_rank_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, perm, by_rank, C, len,
                       count, arena_off, len_sorted, q.cls_count, src_off);
    if (exclusive_scan_with_total(scratch, len_sorted, q.cls_offset, C)) return -1;
    hipLaunchKernelGGL(copy_tuples_direct_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, C, src_off, t.arena,
                       q.cls_offset, q.ids);
    return 0;
}

// Reorder the classes (stably) by their smallest transcript id.  The EM result
// does not depend on class order beyond floating-point association, but the
// transcript-major pass gathers inner[c] over a transcript's classes, and
// classes that share transcripts then sit in neighbouring cache sectors
// instead of being spread over the whole first-seen order.  perm[k] = index of
// internal class k in the caller's order.
int localize(Scratch &scratch, QuantBuild &q, int32_t *perm)
{
    hipStream_t stream = scratch.stream;
    const int64_t C = q.n_classes, M = q.n_ids;
    if (C == 0) return 0;
    if (C >= (1LL << 31) || M >= (1LL << 31)) return -2;
    QB_ALLOC(key, uint32_t, C); QB_ALLOC(key_sorted, uint32_t, C); QB_ALLOC(iota, int32_t, C);
    QB_ALLOC(ids_copy, int32_t, M); QB_ALLOC(len, int64_t, C); QB_ALLOC(len_sorted, int64_t, C);
    QB_ALLOC(old_offset, int64_t, C + 1); QB_ALLOC(count_copy, double, C);
    hipLaunchKernelGGL(class_min_id_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, q.cls_offset, q.ids, C,
                       key, len);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, iota, C);
    int end_bit = 1;
    while ((1LL << end_bit) < q.n_tx && end_bit < 32) ++end_bit;
    if (sort_pairs(scratch, key, key_sorted, iota, perm, C, end_bit)) return -1;
    QB_TRY(hipMemcpyAsync(old_offset, q.cls_offset, (C + 1) * 8, hipMemcpyDeviceToDevice, stream));
    QB_TRY(hipMemcpyAsync(ids_copy, q.ids, M * 4, hipMemcpyDeviceToDevice, stream));
    QB_TRY(hipMemcpyAsync(count_copy, q.cls_count, C * 8, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(gather_classes_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, perm, C, len,
                       count_copy, len_sorted, q.cls_count);
    if (exclusive_scan_with_total(scratch, len_sorted, q.cls_offset, C)) return -1;
    hipLaunchKernelGGL(copy_tuples_by_offset_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, perm, C,
                       old_offset, ids_copy, q.cls_offset, q.ids);
    return 0;
}

int transpose(Scratch &scratch, QuantBuild &q)
{
    hipStream_t stream = scratch.stream;
    const int64_t C = q.n_classes, M = q.n_ids, T = q.n_tx;
    if (M >= (1LL << 31) || C >= (1LL << 31)) return -2;
    QB_ALLOC(pair_cls, int32_t, M); QB_ALLOC(sorted_tx, int32_t, M);
    QB_ALLOC(tx_offset, int64_t, T + 1); QB_ALLOC(rows, int64_t, T);
    if (M > 0) {
        hipLaunchKernelGGL(pair_class_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, q.cls_offset, C,
                           pair_cls);
        int end_bit = 1;
        while ((1LL << end_bit) < T && end_bit < 31) ++end_bit;
        if (sort_pairs(scratch, q.ids, sorted_tx, pair_cls, q.tx_cls, M, end_bit)) return -1;
    }
    hipLaunchKernelGGL(segment_starts_kernel, dim3(blocks_for(M + 1)), dim3(256), 0, stream, sorted_tx, M,
                       T, tx_offset);
    hipLaunchKernelGGL(row_count_kernel, dim3(blocks_for(T)), dim3(256), 0, stream, tx_offset, T, rows);
    if (exclusive_scan_with_total(scratch, rows, q.tx_row, T)) return -1;
    // rows <= T + M / EM_ROW_CAP, which is what row_start / row_tx were sized for
    hipLaunchKernelGGL(row_fill_kernel, dim3(blocks_for(T)), dim3(256), 0, stream, tx_offset, q.tx_row, T,
                       M, q.row_start, q.row_tx);
    return 0;
}

}  // namespace

// The whole setup as one asynchronous pipeline on `stream`: classes from the mapper's table (when
// `table` is given) in locality order with perm[] back to first-seen order -- ranked by bitmap
// when the first-seen values allow it, by two sorts otherwise -- or the caller's classes
// reordered; then the transcript-major rows.  Returns the number of rows or a negative code.
int64_t quant_setup(const ClassTable *table, QuantBuild &q, int32_t *perm, hipStream_t stream)
{
    const bool can_rank = table && q.first_seen_bound > 0 && q.first_seen_bound <= RANK_BITMAP_MAX_UNITS;
    for (int attempt = can_rank ? 0 : 1; attempt < 2; ++attempt) {
        int64_t n_rows = -1;
        {
            Scratch scratch(stream);
            if (attempt == 0) {
                const int rc = build_from_table_ranked(scratch, *table, q, perm);
                if (rc == 1) continue;
                if (rc) return -1;
            } else {
                if (table && build_from_table(scratch, *table, q)) return -1;
                if (localize(scratch, q, perm)) return -1;
            }
            if (transpose(scratch, q)) return -1;
            QB_TRY(hipGetLastError());
            QB_TRY(hipMemcpyAsync(&n_rows, q.tx_row + q.n_tx, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        }                              // ~Scratch: the synchronisation
        if (n_rows > q.n_rows_cap) return -3;
        return n_rows;
    }
    return -1;
}

const char *quant_setup_failure() { return g_setup_failure; }

void warm_code_quant_setup()
{
    hipFuncAttributes attributes;
    (void)hipFuncGetAttributes(&attributes, reinterpret_cast<const void *>(&iota_kernel));
    (void)hipFuncGetAttributes(&attributes, reinterpret_cast<const void *>(&table_dump_kernel));
}

}  // namespace skm
