// Device-side construction of the EM problem: the class-major CSR in the
// reference's class order (ascending first-seen unit = collections.Counter
// insertion order under -j1, /root/reference/seekmer/mapper.py:88) and its
// transcript-major transpose cut into rows.  Library primitives (hipCUB radix
// sort and scan) do the ordering; the small kernels here only move indices.
// One-time setup per quantification, not part of the per-step hot loop.
#include "skm_kernels.h"
#include "skm_pool.h"

#include <hipcub/hipcub.hpp>

namespace skm {

namespace {

template <class T>
struct Tmp {
    T *p = nullptr;
    hipError_t alloc(size_t n) { return pool_alloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)); }
    ~Tmp() { pool_free(p); }
};

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
        arena_off[k] = s.arena_offset;
        len[k] = t.arena_len[i];
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

// exclusive scan of n int64 values into out[0..n) and the total into out[n]
int exclusive_scan_with_total(const int64_t *in, int64_t *out, int64_t n, hipStream_t stream)
{
    if (n == 0) {
        QB_TRY(hipMemsetAsync(out, 0, sizeof(int64_t), stream));
        return 0;
    }
    size_t bytes = 0;
    QB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, stream));
    Tmp<char> tmp;
    QB_TRY(tmp.alloc(bytes));
    QB_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, (int)n, stream));
    hipLaunchKernelGGL(set_last_offset_kernel, dim3(1), dim3(64), 0, stream, out, in, n);
    QB_TRY(hipStreamSynchronize(stream));
    return 0;
}

}  // namespace

int device_exclusive_scan_u64(const unsigned long long *in, unsigned long long *out, int64_t n,
                              unsigned long long *total, hipStream_t stream)
{
    *total = 0;
    if (n == 0) return 0;
    if (n >= (1LL << 31)) return -2;
    size_t bytes = 0;
    QB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, stream));
    Tmp<char> tmp;
    QB_TRY(tmp.alloc(bytes));
    QB_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, (int)n, stream));
    unsigned long long last_in = 0, last_out = 0;
    QB_TRY(hipMemcpyAsync(&last_in, in + n - 1, 8, hipMemcpyDeviceToHost, stream));
    QB_TRY(hipMemcpyAsync(&last_out, out + n - 1, 8, hipMemcpyDeviceToHost, stream));
    QB_TRY(hipStreamSynchronize(stream));
    *total = last_in + last_out;
    return 0;
}

int64_t quant_rows_upper_bound(int64_t n_tx, int64_t n_ids)
{
    return n_tx + n_ids / EM_ROW_CAP + 1;
}

int quant_build_from_table(const ClassTable &t, int64_t n_classes, int64_t n_ids, QuantBuild &q,
                           hipStream_t stream)
{
    if (n_classes == 0) {
        QB_TRY(hipMemsetAsync(q.cls_offset, 0, sizeof(int64_t), stream));
        return 0;
    }
    if (n_classes >= (1LL << 31) || n_ids >= (1LL << 31)) return -2;
    Tmp<int64_t> arena_off, len, len_sorted;
    Tmp<double> count;
    Tmp<unsigned long long> first, first_sorted;
    Tmp<int32_t> iota, perm;
    QB_TRY(arena_off.alloc(n_classes)); QB_TRY(len.alloc(n_classes)); QB_TRY(len_sorted.alloc(n_classes));
    QB_TRY(count.alloc(n_classes)); QB_TRY(first.alloc(n_classes)); QB_TRY(first_sorted.alloc(n_classes));
    QB_TRY(iota.alloc(n_classes)); QB_TRY(perm.alloc(n_classes));
    hipLaunchKernelGGL(table_dump_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream,
                       t, n_classes, arena_off.p, len.p, count.p, first.p);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream, iota.p, n_classes);
    size_t bytes = 0;
    QB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, first.p, first_sorted.p, iota.p, perm.p,
                                             (int)n_classes, 0, 64, stream));
    Tmp<char> tmp;
    QB_TRY(tmp.alloc(bytes));
    QB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, first.p, first_sorted.p, iota.p, perm.p,
                                             (int)n_classes, 0, 64, stream));
    hipLaunchKernelGGL(gather_classes_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream, perm.p,
                       n_classes, len.p, count.p, len_sorted.p, q.cls_count);
    if (exclusive_scan_with_total(len_sorted.p, q.cls_offset, n_classes, stream)) return -1;
    hipLaunchKernelGGL(copy_tuples_kernel, dim3(blocks_for(n_classes)), dim3(256), 0, stream, perm.p,
                       n_classes, arena_off.p, t.arena, q.cls_offset, q.ids);
    QB_TRY(hipGetLastError());
    QB_TRY(hipStreamSynchronize(stream));
    (void)n_ids;
    return 0;
}

// Reorder the classes (stably) by their smallest transcript id.  The EM result
// does not depend on class order beyond floating-point association, but the
// transcript-major pass gathers inner[c] over a transcript's classes, and
// classes that share transcripts then sit in neighbouring cache sectors
// instead of being spread over the whole first-seen order.  perm[k] = index of
// internal class k in the caller's order.
int quant_localize(QuantBuild &q, int32_t *perm, hipStream_t stream)
{
    const int64_t C = q.n_classes, M = q.n_ids;
    if (C == 0) return 0;
    if (C >= (1LL << 31) || M >= (1LL << 31)) return -2;
    Tmp<uint32_t> key, key_sorted;
    Tmp<int32_t> iota, ids_copy;
    Tmp<int64_t> len, len_sorted, old_offset;
    Tmp<double> count_copy;
    QB_TRY(key.alloc(C)); QB_TRY(key_sorted.alloc(C)); QB_TRY(iota.alloc(C)); QB_TRY(ids_copy.alloc(M));
    QB_TRY(len.alloc(C)); QB_TRY(len_sorted.alloc(C)); QB_TRY(old_offset.alloc(C + 1)); QB_TRY(count_copy.alloc(C));
    hipLaunchKernelGGL(class_min_id_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, q.cls_offset, q.ids, C,
                       key.p, len.p);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, iota.p, C);
    int end_bit = 1;
    while ((1LL << end_bit) < q.n_tx && end_bit < 32) ++end_bit;
    size_t bytes = 0;
    QB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, key.p, key_sorted.p, iota.p, perm, (int)C, 0,
                                             end_bit, stream));
    Tmp<char> tmp;
    QB_TRY(tmp.alloc(bytes));
    QB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, key.p, key_sorted.p, iota.p, perm, (int)C, 0,
                                             end_bit, stream));
    QB_TRY(hipMemcpyAsync(old_offset.p, q.cls_offset, (C + 1) * 8, hipMemcpyDeviceToDevice, stream));
    QB_TRY(hipMemcpyAsync(ids_copy.p, q.ids, M * 4, hipMemcpyDeviceToDevice, stream));
    QB_TRY(hipMemcpyAsync(count_copy.p, q.cls_count, C * 8, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(gather_classes_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, perm, C, len.p,
                       count_copy.p, len_sorted.p, q.cls_count);
    if (exclusive_scan_with_total(len_sorted.p, q.cls_offset, C, stream)) return -1;
    hipLaunchKernelGGL(copy_tuples_by_offset_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, perm, C,
                       old_offset.p, ids_copy.p, q.cls_offset, q.ids);
    QB_TRY(hipGetLastError());
    QB_TRY(hipStreamSynchronize(stream));
    return 0;
}

int64_t quant_build_transpose(QuantBuild &q, hipStream_t stream)
{
    const int64_t C = q.n_classes, M = q.n_ids, T = q.n_tx;
    if (M >= (1LL << 31) || C >= (1LL << 31)) return -2;
    Tmp<int32_t> pair_cls, sorted_tx;
    Tmp<int64_t> tx_offset, rows;
    QB_TRY(pair_cls.alloc(M)); QB_TRY(sorted_tx.alloc(M));
    QB_TRY(tx_offset.alloc(T + 1)); QB_TRY(rows.alloc(T));
    if (M > 0) {
        hipLaunchKernelGGL(pair_class_kernel, dim3(blocks_for(C)), dim3(256), 0, stream, q.cls_offset, C,
                           pair_cls.p);
        int end_bit = 1;
        while ((1LL << end_bit) < T && end_bit < 31) ++end_bit;
        size_t bytes = 0;
        QB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, q.ids, sorted_tx.p, pair_cls.p, q.tx_cls,
                                                 (int)M, 0, end_bit, stream));
        Tmp<char> tmp;
        QB_TRY(tmp.alloc(bytes));
        QB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, q.ids, sorted_tx.p, pair_cls.p, q.tx_cls,
                                                 (int)M, 0, end_bit, stream));
        QB_TRY(hipStreamSynchronize(stream));
    }
    hipLaunchKernelGGL(segment_starts_kernel, dim3(blocks_for(M + 1)), dim3(256), 0, stream, sorted_tx.p, M,
                       T, tx_offset.p);
    hipLaunchKernelGGL(row_count_kernel, dim3(blocks_for(T)), dim3(256), 0, stream, tx_offset.p, T, rows.p);
    if (exclusive_scan_with_total(rows.p, q.tx_row, T, stream)) return -1;
    int64_t n_rows = 0;
    QB_TRY(hipMemcpyAsync(&n_rows, q.tx_row + T, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    QB_TRY(hipStreamSynchronize(stream));
    if (n_rows > q.n_rows_cap) return -3;
    hipLaunchKernelGGL(row_fill_kernel, dim3(blocks_for(T)), dim3(256), 0, stream, tx_offset.p, q.tx_row, T,
                       M, q.row_start, q.row_tx);
    QB_TRY(hipGetLastError());
    QB_TRY(hipStreamSynchronize(stream));
    return n_rows;
}

}  // namespace skm
