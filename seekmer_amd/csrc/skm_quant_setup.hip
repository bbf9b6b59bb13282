// Device-side construction of the EM problem: the class-major CSR in the
// reference's class order (ascending first-seen unit = collections.Counter
// insertion order under -j1, /root/reference/seekmer/mapper.py:88) and its
// transcript-major transpose cut into rows.  Library primitives (hipCUB radix
// sort and scan) do the ordering; the small kernels here only move indices.
// One-time setup per quantification, not part of the per-step hot loop.
#include "skm_kernels.h"
#include "skm_pool.h"

#include <hipcub/hipcub.hpp>
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
#define QB_ALLOC(var, type, n) type *var = scratch.alloc<type>(n); if (!var) return -1

#define QB_TRY(call) do { if ((call) != hipSuccess) return -1; } while (0)

__global__ void __launch_bounds__(256)
iota_kernel(int32_t *out, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) out[i] = (int32_t)i;
}

// dump of the table in registry order: arena offset, length, count, first-seen per class
__global__ void __launch_bounds__(256)
table_dump_kernel(ClassTable t, int64_t n_classes, int64_t *arena_off, int64_t *len, double *count,
                  unsigned long long *first_seen)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_classes;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t.class_list[k];
        const ClassSlot s = t.slots[i];
        arena_off[k] = s.tuple < 0 ? -1 : tuple_offset(s.tuple);
        len[k] = s.tuple < 0 ? 0 : tuple_len(s.tuple);
        count[k] = (double)s.count;
        first_seen[k] = s.first_seen;
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

__global__ void __launch_bounds__(64)
sum_last_kernel(const unsigned long long *in, const unsigned long long *scan, int64_t n,
                unsigned long long *total)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *total = in[n - 1] + scan[n - 1];
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
        const int64_t degree = tx_offset[t + 1] - tx_offset[t];
        rows[t] = (degree + EM_ROW_CAP - 1) / EM_ROW_CAP;
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

template <class K>
int sort_pairs(Scratch &scratch, const K *keys_in, K *keys_out, const int32_t *vals_in, int32_t *vals_out,
               int64_t n, int end_bit)
{
    size_t bytes = 0;
    QB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0,
                                             end_bit, scratch.stream));
    QB_ALLOC(tmp, char, bytes);
    QB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0,
                                             end_bit, scratch.stream));
    return 0;
}

}  // namespace

size_t device_scan_u64_temp_bytes(int64_t n)
{
    size_t bytes = 0;
    unsigned long long *p = nullptr;
    if (n <= 0 || n >= (1LL << 31)) return 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, p, p, (int)n, nullptr) != hipSuccess) return 0;
    return bytes;
}

// out = exclusive prefix sum of in[0..n); *total_device (HBM) = sum of all; asynchronous
int device_exclusive_scan_u64(const unsigned long long *in, unsigned long long *out, int64_t n,
                              unsigned long long *total_device, void *temp, size_t temp_bytes,
                              hipStream_t stream)
{
    if (n == 0) { QB_TRY(hipMemsetAsync(total_device, 0, 8, stream)); return 0; }
    if (n >= (1LL << 31)) return -2;
    QB_TRY(hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, (int)n, stream));
    hipLaunchKernelGGL(sum_last_kernel, dim3(1), dim3(64), 0, stream, in, out, n, total_device);
    return 0;
}

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
                       t, n_classes, arena_off, len, count, first);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream, iota, n_classes);
    const int first_bits = q.first_seen_bits > 0 && q.first_seen_bits < 64 ? q.first_seen_bits : 64;
    if (sort_pairs(scratch, first, first_sorted, iota, perm, n_classes, first_bits)) return -1;
    hipLaunchKernelGGL(gather_classes_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream, perm,
                       n_classes, len, count, len_sorted, q.cls_count);
    if (exclusive_scan_with_total(scratch, len_sorted, q.cls_offset, n_classes)) return -1;
    hipLaunchKernelGGL(copy_tuples_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream, perm,
                       n_classes, arena_off, t.arena, q.cls_offset, q.ids);
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

// The whole setup as one asynchronous pipeline on `stream`: (classes from the
// mapper's table in first-seen order, when `table` is given) -> locality order
// -> transcript-major rows.  One synchronisation at the end; returns the number
// of rows or a negative code.
int64_t quant_setup(const ClassTable *table, QuantBuild &q, int32_t *perm, hipStream_t stream)
{
    int64_t n_rows = -1;
    {
        Scratch scratch(stream);
        if (table && build_from_table(scratch, *table, q)) return -1;
        if (localize(scratch, q, perm)) return -1;
        if (transpose(scratch, q)) return -1;
        QB_TRY(hipGetLastError());
        QB_TRY(hipMemcpyAsync(&n_rows, q.tx_row + q.n_tx, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    }                                  // ~Scratch: the single synchronisation
    if (n_rows > q.n_rows_cap) return -3;
    return n_rows;
}

}  // namespace skm
