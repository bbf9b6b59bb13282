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

#include <algorithm>
#include <cstdio>
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
        // (eight ids in flight at a time: a tuple is 5.4 ids on average, somewhere in the arena)
        for (int64_t i = 0; i < len; i += 8) {
            int32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = i + k < len ? arena[src + i + k] : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (i + k < len) ids[dst + i + k] = v[k];
        }
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

// ---- hand-written scan and stable radix sort (no library code on the path) -----------------------
// Exclusive prefix sums of n values, three small launches: per-tile sums, one block over the tile
// sums, per-tile scan with the tile's base.  SCAN_TILE values per block; the middle launch handles
// any number of tiles (each of its 1024 lanes walks a run of them).
constexpr int SCAN_TILE = 2048;               // 256 lanes x 8 values

template <class T>
__global__ void __launch_bounds__(256)
scan_tile_sums_kernel(const T *__restrict__ in, int64_t n, T *__restrict__ tile_sums)
{
    __shared__ T s_wave[4];
    const int64_t base = blockIdx.x * (int64_t)SCAN_TILE + threadIdx.x * 8;
    T sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += base + k < n ? in[base + k] : (T)0;
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// tile_sums[i] <- sum of the tiles before i; *total (optional) <- the sum of all
template <class T>
__global__ void __launch_bounds__(1024)
scan_of_sums_kernel(T *tile_sums, int64_t n_tiles, T *total)
{
    __shared__ T s_part[1024];
    const int64_t per = (n_tiles + 1023) / 1024;
    const int64_t first = threadIdx.x * per, last = min(n_tiles, first + per);
    T sum = 0;
    for (int64_t i = first; i < last; ++i) sum += tile_sums[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int step = 1; step < 1024; step <<= 1) {
        const T add = (int)threadIdx.x >= step ? s_part[threadIdx.x - step] : (T)0;
        __syncthreads();
        s_part[threadIdx.x] += add;
        __syncthreads();
    }
    T at = s_part[threadIdx.x] - sum;
    for (int64_t i = first; i < last; ++i) { const T v = tile_sums[i]; tile_sums[i] = at; at += v; }
    if (total && threadIdx.x == 1023) *total = s_part[1023];
}

template <class T>
__global__ void __launch_bounds__(256)
scan_tiles_kernel(const T *__restrict__ in, int64_t n, const T *__restrict__ tile_base, T *__restrict__ out)
{
    __shared__ T s_wave[4];
    const int64_t base = blockIdx.x * (int64_t)SCAN_TILE + threadIdx.x * 8;
    T v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? in[base + k] : (T)0; sum += v[k]; }
    T incl = sum;                                  // inclusive scan of the lanes' sums over the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    T at = tile_base[blockIdx.x] + incl - sum;
    for (int w = 0; w < wave; ++w) at += s_wave[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) out[base + k] = at;
        at += v[k];
    }
}

// out[0..n) = exclusive prefix sums of in, `total` (device, optional) = the sum; in == out allowed
template <class T>
int exclusive_scan(Scratch &scratch, const T *in, T *out, int64_t n, T *total)
{
    hipStream_t stream = scratch.stream;
    if (n <= 0) {
        if (total) QB_TRY(hipMemsetAsync(total, 0, sizeof(T), stream));
        return 0;
    }
    const int64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    QB_ALLOC(tile_sums, T, n_tiles);
    hipLaunchKernelGGL(scan_tile_sums_kernel<T>, dim3((unsigned)n_tiles), dim3(256), 0, stream, in, n, tile_sums);
    hipLaunchKernelGGL(scan_of_sums_kernel<T>, dim3(1), dim3(1024), 0, stream, tile_sums, n_tiles, total);
    hipLaunchKernelGGL(scan_tiles_kernel<T>, dim3((unsigned)n_tiles), dim3(256), 0, stream, in, n, tile_sums, out);
    return 0;
}

// exclusive scan of n int64 values into out[0..n) and the total into out[n] (all on the stream)
int exclusive_scan_with_total(Scratch &scratch, const int64_t *in, int64_t *out, int64_t n)
{
    if (n >= (1LL << 40)) return -2;
    return exclusive_scan<int64_t>(scratch, in, out, n, out + n);
}

// Stable least-significant-digit radix sort of (key, value) pairs on key bits [0, end_bit), 9 bits
// (512 bins) per pass: the 18-bit keys of the class views take two passes.  A pass: a histogram per
// tile of 4096 pairs (digit-major, so that ONE exclusive scan over it gives every (digit, tile) its
// place), then the scatter -- a tile's four waves each own a quarter of it in order; a wave ranks 64
// pairs at a time with nine ballots (the lanes that share my digit) against its running per-digit
// base in LDS, the pairs are put in digit order in LDS and leave in runs.  Stable by construction:
// tiles, waves, 64-pair chunks and lanes are all taken in input order within a digit.
constexpr int RS_BITS = 9, RS_BINS = 1 << RS_BITS, RS_ITEMS = 16, RS_TILE = 256 * RS_ITEMS;

template <class K>
__global__ void __launch_bounds__(256)
radix_hist_kernel(const K *__restrict__ keys, int64_t n, int shift, uint32_t *__restrict__ hist, int64_t n_tiles)
{
    __shared__ uint32_t h[RS_BINS];
    for (int d = threadIdx.x; d < RS_BINS; d += 256) h[d] = 0;
    __syncthreads();
    const int64_t base = blockIdx.x * (int64_t)RS_TILE;
#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t at = base + i * 256 + threadIdx.x;
        if (at < n) atomicAdd(&h[(uint32_t)((unsigned long long)keys[at] >> shift) & (RS_BINS - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RS_BINS; d += 256) hist[d * n_tiles + blockIdx.x] = h[d];
}

// one block per digit: hist[d][0 .. n_tiles) <- its exclusive prefix sums, digit_total[d] <- its sum
// (the scatter adds the digits before d: 512 totals, scanned by every block for itself)
__global__ void __launch_bounds__(256)
radix_digit_scan_kernel(uint32_t *hist, int64_t n_tiles, uint32_t *digit_total)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    uint32_t *row = hist + blockIdx.x * n_tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_tiles; base += 256 * 4) {
        const int64_t at = base + threadIdx.x * 4;
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = at + k < n_tiles ? row[at + k] : 0u; sum += v[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t run = s_carry + incl - sum;
        for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (at + k < n_tiles) row[at + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = run;           // (the last lane's running sum = everything so far)
        __syncthreads();
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = s_carry;
}

template <class K>
__global__ void __launch_bounds__(256)
radix_scatter_kernel(const K *__restrict__ keys_in, const int32_t *__restrict__ vals_in, K *__restrict__ keys_out,
                     int32_t *__restrict__ vals_out, int64_t n, int shift, const uint32_t *__restrict__ offsets,
                     const uint32_t *__restrict__ digit_total, int64_t n_tiles)
{
    __shared__ uint32_t wave_base[4][RS_BINS];      // a wave's count per digit, then its running place in the tile
    __shared__ uint32_t tile_start[RS_BINS + 1];    // where a digit's pairs start in the tile's digit order
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t digit_base[RS_BINS];        // pairs of the whole input with a smaller digit
    __shared__ K s_keys[RS_TILE];
    __shared__ int32_t s_vals[RS_TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < 4 * RS_BINS; d += 256) (&wave_base[0][0])[d] = 0;
    {   // digit_base = exclusive scan of the 512 digit totals (two digits per lane)
        const uint32_t t0 = digit_total[2 * threadIdx.x], t1 = digit_total[2 * threadIdx.x + 1];
        uint32_t incl = t0 + t1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        uint32_t at = incl - (t0 + t1);
        for (int w = 0; w < wave; ++w) at += s_scan[w];
        digit_base[2 * threadIdx.x] = at;
        digit_base[2 * threadIdx.x + 1] = at + t0;
    }
    __syncthreads();
    // the wave's quarter of the tile, 64 consecutive pairs per chunk
    const int64_t first = blockIdx.x * (int64_t)RS_TILE + wave * (RS_TILE / 4);
    K key[RS_ITEMS];
    int32_t val[RS_ITEMS];
    uint32_t digit[RS_ITEMS];
#pragma unroll
    for (int c = 0; c < RS_ITEMS; ++c) {
        const int64_t at = first + c * 64 + lane;
        const bool valid = at < n;
        key[c] = valid ? keys_in[at] : (K)0;
        val[c] = valid ? vals_in[at] : 0;
        digit[c] = (uint32_t)((unsigned long long)key[c] >> shift) & (RS_BINS - 1);
        if (valid) atomicAdd(&wave_base[wave][digit[c]], 1u);
    }
    __syncthreads();
    // tile_start = exclusive scan over the digits of the tile's counts (two digits per lane)
    const int d0 = 2 * threadIdx.x;
    const uint32_t c0 = wave_base[0][d0] + wave_base[1][d0] + wave_base[2][d0] + wave_base[3][d0];
    const uint32_t c1 = wave_base[0][d0 + 1] + wave_base[1][d0 + 1] + wave_base[2][d0 + 1] + wave_base[3][d0 + 1];
    uint32_t incl = c0 + c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    uint32_t at0 = incl - (c0 + c1);
    for (int w = 0; w < wave; ++w) at0 += s_scan[w];
    tile_start[d0] = at0;
    tile_start[d0 + 1] = at0 + c0;
    if (threadIdx.x == 255) tile_start[RS_BINS] = at0 + c0 + c1;
    // a wave's running place for a digit: behind the earlier waves' pairs of that digit
    uint32_t run0 = at0, run1 = at0 + c0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t n0 = wave_base[w][d0], n1 = wave_base[w][d0 + 1];
        wave_base[w][d0] = run0; wave_base[w][d0 + 1] = run1;
        run0 += n0; run1 += n1;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < RS_ITEMS; ++c) {
        const bool valid = first + c * 64 + lane < n;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RS_BITS; ++b) {
            const bool bit = (digit[c] >> b) & 1u;
            const unsigned long long with = __ballot(valid && bit);
            same &= bit ? with : ~with;
        }
        if (valid) {
            const uint32_t before = (uint32_t)__popcll(same & ((1ULL << lane) - 1ULL));
            const uint32_t place = wave_base[wave][digit[c]] + before;
            s_keys[place] = key[c];
            s_vals[place] = val[c];
            if (before == 0) wave_base[wave][digit[c]] += (uint32_t)__popcll(same);      // (the digit's first lane)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next chunk reads the bases this one moved
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    const int64_t tile_first = blockIdx.x * (int64_t)RS_TILE;
    const uint32_t tile_n = (uint32_t)min((int64_t)RS_TILE, n - tile_first);
#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint32_t j = i * 256 + threadIdx.x;
        if (j >= tile_n) continue;
        const K k = s_keys[j];
        const uint32_t d = (uint32_t)((unsigned long long)k >> shift) & (RS_BINS - 1);
        const int64_t to = (int64_t)digit_base[d] + offsets[d * n_tiles + blockIdx.x] + (j - tile_start[d]);
        keys_out[to] = k;
        vals_out[to] = s_vals[j];
    }
}

// Stable radix sort of (key, value) pairs on bits [0, end_bit); the result lands in keys_out / vals_out.
template <class K>
int sort_pairs(Scratch &scratch, const K *keys_in, K *keys_out, const int32_t *vals_in, int32_t *vals_out,
               int64_t n, int end_bit)
{
    hipStream_t stream = scratch.stream;
    if (n <= 0) return 0;
    if (n >= (1LL << 32)) return -2;                                  // (offsets are 32-bit)
    const int passes = std::max(1, (end_bit + RS_BITS - 1) / RS_BITS);
    const int64_t n_tiles = (n + RS_TILE - 1) / RS_TILE;
    QB_ALLOC(hist, uint32_t, RS_BINS * n_tiles);
    QB_ALLOC(digit_total, uint32_t, RS_BINS);
    K *tmp_keys = nullptr;
    int32_t *tmp_vals = nullptr;
    if (passes > 1) {
        QB_ALLOC(tk, K, n); QB_ALLOC(tv, int32_t, n);
        tmp_keys = tk; tmp_vals = tv;
    }
    const K *src_k = keys_in;
    const int32_t *src_v = vals_in;
    for (int pass = 0; pass < passes; ++pass) {
        // (the last pass writes the caller's arrays; the passes before it alternate so that it can)
        const bool to_out = ((passes - 1 - pass) & 1) == 0;
        K *dst_k = to_out ? keys_out : tmp_keys;
        int32_t *dst_v = to_out ? vals_out : tmp_vals;
        const int shift = pass * RS_BITS;
        hipLaunchKernelGGL(radix_hist_kernel<K>, dim3((unsigned)n_tiles), dim3(256), 0, stream, src_k, n, shift, hist, n_tiles);
        hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(RS_BINS), dim3(256), 0, stream, hist, n_tiles, digit_total);
        hipLaunchKernelGGL(radix_scatter_kernel<K>, dim3((unsigned)n_tiles), dim3(256), 0, stream, src_k, src_v, dst_k, dst_v, n,
                           shift, hist, digit_total, n_tiles);
        src_k = dst_k;
        src_v = dst_v;
    }
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
