// Pseudoalignment kernels for gfx950: read packing and the contig-jumping
// mapper (one read or read pair per lane, 64 units per wavefront).
//
// What is computed is the reference's per-read state machine
// (/root/reference/seekmer/_mapper.pyx:111-343, 350-501) -- every branch is
// cited below -- but the data flow is built for CDNA4: reads are first packed
// by a coalesced streaming kernel to 2 bits per base + an "is upper-case
// ACGT" bit plane, k-mers and 8-base windows are then funnel-shifted out of
// registers instead of being re-encoded byte by byte, contig bases are
// fetched from a 2-bit pool, and the running target list of a lane lives in a
// lane-interleaved HBM workspace (word i of every lane is contiguous, so the
// lock-step part of list copies coalesces).
#include "skm_device.h"
#include "skm_kernels.h"

namespace skm {

// ---------------------------------------------------------------- pack_reads
// One lane per 32-base word of a read.  ASCII -> (2-bit code word, 32-bit
// ACGT mask word).  Word index space is [n_reads][words_per_read].
__global__ void __launch_bounds__(256)
pack_reads_kernel(const uint8_t *__restrict__ bases, const int64_t *__restrict__ offsets,
                  int64_t n_reads, int words_per_read,
                  uint64_t *__restrict__ codes, uint32_t *__restrict__ acgt)
{
    const int64_t total = n_reads * (int64_t)words_per_read;
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = g / words_per_read;
        const int w = (int)(g - r * words_per_read);
        const int64_t begin = offsets[r];
        const int len = (int)(offsets[r + 1] - begin);
        uint64_t c = 0;
        uint32_t m = 0;
        const int first = w * 32;
        if (first < len) {
            const int n = min(32, len - first);
            const uint8_t *p = bases + begin + first;
            for (int i = 0; i < n; ++i) {
                const uint32_t ch = p[i];
                c |= (uint64_t)two_bit_encode(ch) << (62 - 2 * i);
                const bool up = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
                m |= (uint32_t)up << (31 - i);
            }
        }
        codes[g] = c;
        acgt[g] = m;
    }
}

// ------------------------------------------------------------------- mapper
// Running target list of one lane: element i at p[i * stride].
struct TList {
    int32_t *p;
    size_t stride;
    __device__ __forceinline__ int32_t get(int i) const { return p[(size_t)i * stride]; }
    __device__ __forceinline__ void set(int i, int32_t v) const { p[(size_t)i * stride] = v; }
};

struct Span {               // MappedSpan, _common.pxd:31-35 (targets = list + n)
    int32_t begin, end;
    Coord anchor;
    int32_t n;
};

// KMerIndex.map_contig, _common.pyx:143-179.  Only `entry` of a target is ever
// read by the mapper, so the list holds entries alone.
template <bool STATS>
__device__ __forceinline__ void map_contig(const DevIndex &ix, Coord c, const TList &list,
                                           Span &span, LaneStats *st)
{
    const bool forward = c.entry >= 0;
    const int32_t index = forward ? c.entry : ~c.entry;
    const int32_t start = (int32_t)ix.contigs[index].target_offset;
    int32_t length = (int32_t)ix.contigs[index].target_length;
    if (length > ix.max_target_count) length = ix.max_target_count;   // workspace bound
    if (STATS) { st->contig_reads++; st->targets_copied += length; }
    span.n = length;
    if (forward) {
        for (int i = 0; i < length; ++i) list.set(i, ix.targets[start + i].entry);
    } else {
        for (int i = 0; i < length; ++i) list.set(i, ~ix.targets[start + length - 1 - i].entry);
    }
}

// KMerIndex._filter_on_contig, _common.pyx:185-235
template <bool STATS>
__device__ __forceinline__ bool filter_on_contig(const DevIndex &ix, const TList &list, Span &span,
                                                 LaneStats *st)
{
    if (STATS) st->merges++;
    if (span.n == 0) return true;
    const bool forward = span.anchor.entry >= 0;
    const int32_t contig = forward ? span.anchor.entry : ~span.anchor.entry;
    const int32_t start = (int32_t)ix.contigs[contig].target_offset;
    const int32_t length = (int32_t)ix.contigs[contig].target_length;
    if (STATS) st->contig_reads++;
    int read_index = 0, write_index = 0;
    int track = forward ? start : start + length - 1;
    const int bound = forward ? start + length : start - 1;
    const int step = forward ? 1 : -1;
    const int first_track = track;
    while (read_index != span.n && track != bound) {
        const int32_t target_entry = list.get(read_index);
        int32_t index_entry = ix.targets[track].entry;
        if (!forward) index_entry = ~index_entry;
        if (target_entry == index_entry) {
            list.set(write_index, target_entry);
            ++read_index; ++write_index; track += step;
        } else if (target_entry < index_entry) {
            ++read_index;
        } else {
            track += step;
        }
    }
    if (STATS) st->targets_merged += (uint32_t)((track - first_track) * step + (track != bound ? 1 : 0));
    if (write_index == 0) return false;
    span.n = write_index;
    return true;
}

// 10-base neighbourhood of the read that one SIFT4 scan can touch
struct QWindow {
    uint32_t codes, acgt;
    int base;
    __device__ __forceinline__ bool match(uint32_t ref_code, int pos) const   // _mapper.pyx:500-501
    {
        const int j = pos - base;
        const bool is_acgt = (acgt >> (15 - j)) & 1u;
        return !is_acgt || ((codes >> (30 - 2 * j)) & 3u) == ref_code;
    }
};
__device__ __forceinline__ uint32_t ref_code(uint32_t ref8, int i) { return (ref8 >> (14 - 2 * i)) & 3u; }

// sift4_align_left, _mapper.pyx:404-445 (the query cursor starts one base
// short of the reference cursor, lines 406-408)
__device__ __forceinline__ int sift4_left(uint32_t ref8, const ReadView &r, int offset)
{
    QWindow q;
    q.base = offset > 0 ? offset - 1 : 0;
    read_window16(r, q.base, q.codes, q.acgt);
    int rc = ALIGN_LENGTH - 1;
    int qc = offset + ALIGN_LENGTH - 2;
    int distance = 0;
    for (int guard = 0; guard < 64 && rc >= 0 && qc >= offset; ++guard) {
        if (q.match(ref_code(ref8, rc), qc)) { --rc; --qc; continue; }
        if (rc != qc - offset) { rc = min(qc - offset, rc); qc = rc + offset; }
#pragma unroll
        for (int i = 0; i < MAX_OFFSET; ++i) {
            if (qc - i >= offset - 1 && qc - i >= 0 && q.match(ref_code(ref8, rc), qc - i)) {
                distance += i - 1; qc -= i - 1; rc += 1;
                break;
            }
            if (rc - i >= 0 && q.match(ref_code(ref8, rc - i), qc)) {
                distance += i - 1; qc += 1; rc -= i - 1;
                break;
            }
        }
        distance += 1; --qc; --rc;
        if (distance > MAX_DISTANCE) return INVALID_SHIFT;
    }
    if (rc >= 0) return rc + 1;
    if (qc >= offset) return -1 - qc + offset;
    return 0;
}

// sift4_align_right, _mapper.pyx:452-493
__device__ __forceinline__ int sift4_right(uint32_t ref8, const ReadView &r, int offset)
{
    QWindow q;
    q.base = offset;
    read_window16(r, q.base, q.codes, q.acgt);
    int rc = 0;
    int qc = offset;
    int distance = 0;
    for (int guard = 0; guard < 64 && rc < ALIGN_LENGTH && qc < offset + ALIGN_LENGTH; ++guard) {
        if (q.match(ref_code(ref8, rc), qc)) { ++rc; ++qc; continue; }
        if (rc != qc - offset) { rc = max(qc - offset, rc); qc = rc + offset; }
#pragma unroll
        for (int i = 0; i < MAX_OFFSET; ++i) {
            if (qc + i < offset + ALIGN_LENGTH + 1 && qc + i < r.len
                    && q.match(ref_code(ref8, rc), qc + i)) {
                distance += i - 1; qc += i - 1; rc -= 1;
                break;
            }
            if (rc + i < ALIGN_LENGTH && q.match(ref_code(ref8, rc + i), qc)) {
                distance += i - 1; qc -= 1; rc += i - 1;
                break;
            }
        }
        distance += 1; ++qc; ++rc;
        if (distance > MAX_DISTANCE) return INVALID_SHIFT;
    }
    if (rc < ALIGN_LENGTH) return ALIGN_LENGTH - rc;
    if (qc < offset + ALIGN_LENGTH) return qc - offset - ALIGN_LENGTH;
    return 0;
}

// _find_first_kmer, _mapper.pyx:199-216
template <bool STATS>
__device__ __forceinline__ void find_first_kmer(const DevIndex &ix, const ReadView &r,
                                                const TList &list, Span &span, LaneStats *st)
{
    uint64_t kmer = read_kmer(r, span.begin);
    span.anchor = map_kmer<STATS>(ix, kmer, st);
    if (span.anchor.offset >= 0) {
        span.end = span.begin;
        map_contig<STATS>(ix, span.anchor, list, span, st);
        return;
    }
    for (int i = span.begin + K; i < r.len; ++i) {
        kmer = ((kmer << 2) | read_code(r, i)) & KMER_MASK;        // _kmer.append
        span.anchor = map_kmer<STATS>(ix, kmer, st);
        if (span.anchor.offset < 0) continue;
        span.begin = i + 1 - K;
        span.end = span.begin;
        map_contig<STATS>(ix, span.anchor, list, span, st);
        return;
    }
}

__device__ __forceinline__ int left_move(const DevIndex &ix, Coord a)
{
    const bool forward = a.entry >= 0;
    const int32_t contig = forward ? a.entry : ~a.entry;
    return forward ? a.offset : (int)ix.contigs[contig].length - a.offset - K;
}
__device__ __forceinline__ int right_move(const DevIndex &ix, Coord a)
{
    const bool forward = a.entry >= 0;
    const int32_t contig = forward ? a.entry : ~a.entry;
    return forward ? (int)ix.contigs[contig].length - a.offset - K : a.offset;
}

// _filter_targets_to_left, _mapper.pyx:222-275
template <bool STATS>
__device__ __forceinline__ void filter_left(const DevIndex &ix, const ReadView &r, const TList &list,
                                            Span &span, LaneStats *st)
{
    bool forward = span.anchor.entry >= 0;
    int move = left_move(ix, span.anchor);
    if (STATS) st->contig_reads++;
    while (span.begin > move) {
        span.begin -= move;
        span.anchor.offset -= forward ? move : -move;
        int shift = sift4_left(contig8<STATS>(ix, span.anchor, true, st), r, span.begin);
        if (shift == INVALID_SHIFT || shift + 1 + move <= 0) { span.n = 0; return; }
        span.begin -= shift + 1;
        if (span.begin < 0) { span.begin = 0; return; }
        // _kmer.prepend(get_tail_kmer(anchor), read[begin])
        uint64_t kmer = (tail_kmer<STATS>(ix, span.anchor, st) >> 2)
                        | ((uint64_t)read_code(r, span.begin) << (2 * K - 2));
        span.anchor = map_kmer<STATS>(ix, kmer, st);
        if (!(span.anchor.offset >= 0) || !filter_on_contig<STATS>(ix, list, span, st)) {
            if (span.begin < K) { span.begin = 0; return; }
            span.begin -= K;
            kmer = read_kmer(r, span.begin);
            span.anchor = map_kmer<STATS>(ix, kmer, st);
            if (!(span.anchor.offset >= 0) || !filter_on_contig<STATS>(ix, list, span, st)) {
                span.n = 0;
                return;
            }
        }
        forward = span.anchor.entry >= 0;
        move = left_move(ix, span.anchor);
        if (STATS) st->contig_reads++;
    }
    span.anchor.offset -= forward ? span.begin : -span.begin;
    if (sift4_left(contig8<STATS>(ix, span.anchor, true, st), r, 0) == INVALID_SHIFT) span.n = 0;
}

// _filter_targets_to_right, _mapper.pyx:281-343 (lines 316-329 are dead code)
template <bool STATS>
__device__ __forceinline__ void filter_right(const DevIndex &ix, const ReadView &r, const TList &list,
                                             Span &span, LaneStats *st)
{
    span.anchor = map_kmer<STATS>(ix, read_kmer(r, span.end), st);
    bool forward = span.anchor.entry >= 0;
    int move = right_move(ix, span.anchor);
    if (STATS) st->contig_reads++;
    while (r.len - span.end - K > move) {
        span.end += move;
        span.anchor.offset += forward ? move : -move;
        int shift = sift4_right(contig8<STATS>(ix, span.anchor, false, st), r,
                                span.end + K - ALIGN_LENGTH);
        if (shift == INVALID_SHIFT || shift + 1 + move <= 0) { span.n = 0; return; }
        span.end += shift + 1;
        if (span.end + K > r.len) { span.end = r.len - K; return; }
        // _kmer.append(get_tail_kmer(anchor), read[end + k - 1])
        uint64_t kmer = ((tail_kmer<STATS>(ix, span.anchor, st) << 2)
                         | read_code(r, span.end + K - 1)) & KMER_MASK;
        span.anchor = map_kmer<STATS>(ix, kmer, st);
        if (!(span.anchor.offset >= 0) || !filter_on_contig<STATS>(ix, list, span, st)) {
            span.n = 0;
            return;
        }
        forward = span.anchor.entry >= 0;
        move = right_move(ix, span.anchor);
        if (STATS) st->contig_reads++;
    }
    const int rest = r.len - span.end - K;
    span.anchor.offset += forward ? rest : -rest;
    if (sift4_right(contig8<STATS>(ix, span.anchor, false, st), r, r.len - ALIGN_LENGTH)
            == INVALID_SHIFT)
        span.n = 0;
}

// map_read, _mapper.pyx:151-193.  Reads shorter than k (undefined behaviour in
// the reference) are reported unmapped with the initial span.
template <bool STATS>
__device__ __forceinline__ Span map_read(const DevIndex &ix, const ReadView &r, const TList &list,
                                      LaneStats *st)
{
    Span span;
    span.anchor = invalid_coord();
    span.begin = 0;
    span.end = 0;
    span.n = 0;
    if (r.len < K) return span;
    for (int attempt = 0; attempt < 2; ++attempt) {
        find_first_kmer<STATS>(ix, r, list, span, st);
        if (span.n == 0) return span;
        if (span.begin > 0) filter_left<STATS>(ix, r, list, span, st);
        if (span.n != 0 && span.end < r.len - K) filter_right<STATS>(ix, r, list, span, st);
        if (span.n != 0 || attempt == 1) return span;
        span.anchor = invalid_coord();                    // single retry, lines 179-184
        span.begin += K;
        if (span.begin + K > r.len) span.begin = r.len - K;
        span.end = span.begin;
    }
    return span;
}

// _intersect, _mapper.pyx:350-397
__device__ __forceinline__ bool intersect(const TList &l1, Span &s1, const TList &l2, const Span &s2)
{
    if (s1.n == 0) return true;
    if (s2.n == 0) return false;
    int read1 = 0, write1 = 0, cursor2 = s2.n - 1;
    while (read1 != s1.n && cursor2 != -1) {
        const int32_t e1 = l1.get(read1);
        const int32_t e2 = ~l2.get(cursor2);
        if (e1 == e2) { l1.set(write1, e1); ++read1; ++write1; --cursor2; }
        else if (e1 < e2) ++read1;
        else --cursor2;
    }
    if (write1 == 0) return false;
    s1.n = write1;
    return true;
}

// 64-bit key of a class tuple (unsigned ids in list order).  Never 0 (0 marks
// an empty table slot).  Full tuples are compared later; this is only the tag.
__device__ __forceinline__ uint64_t tuple_key_step(uint64_t h, uint32_t id)
{
    h ^= id;
    h *= 0x9E3779B97F4A7C15ULL;
    h ^= h >> 32;
    return h;
}

template <bool STATS>
__global__ void __launch_bounds__(256)
map_units_kernel(DevIndex ix, MapBatch b)
{
    __shared__ uint32_t fld_lds[MAX_FRAGMENT_LENGTH];
    for (int i = threadIdx.x; i < MAX_FRAGMENT_LENGTH; i += blockDim.x) fld_lds[i] = 0;
    __syncthreads();

    const int64_t total_threads = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const size_t stride = (size_t)total_threads;
    const TList list1{b.workspace + gtid, stride};
    const TList list2{b.workspace + (size_t)ix.max_target_count * stride + gtid, stride};
    LaneStats ls = {0, 0, 0, 0, 0, 0, 0};
    uint64_t read_bases = 0, n_reads = 0;

    // all 64 lanes of a wave run the same trip count (wave-wide scan below)
    for (int64_t first = gtid - lane; first < b.n_units; first += total_threads) {
        const int64_t u = first + lane;
        const bool active = u < b.n_units;
        Span s1, s2;
        s1.begin = 0; s1.end = 0; s1.n = 0; s1.anchor = invalid_coord();
        s2 = s1;
        if (active) {
            const int64_t r1 = b.paired ? 2 * u : u;
            int len1 = 0, len2 = 0;
            const int mates = b.paired ? 2 : 1;
            for (int m = 0; m < mates; ++m) {          // single inlined copy of the state machine
                const int64_t r = r1 + m;
                ReadView v{b.codes + r * b.words_per_read, b.acgt + r * b.words_per_read,
                           (int)(b.offsets[r + 1] - b.offsets[r])};
                const Span s = map_read<STATS>(ix, v, m ? list2 : list1, &ls);
                if (STATS) { read_bases += v.len; n_reads++; }
                if (m == 0) { s1 = s; len1 = v.len; } else { s2 = s; len2 = v.len; }
            }
            if (b.paired) {
                // map_read_pair, _mapper.pyx:111-145
                if (!intersect(list1, s1, list2, s2)) {
                    s1.n = 0;
                    s1.begin = 0;
                    s1.end = -K;
                } else if (s1.anchor.entry != ~s2.anchor.entry) {
                    s1.begin = 0;
                    s1.end = -K;
                } else {
                    s1.end = len1 - K;
                    s2.end = len2 - K;
                    int interval = s2.anchor.offset - s1.anchor.offset;
                    if (s1.anchor.entry < 0) interval = -interval;
                    s1.end += interval + s2.end - s2.begin;
                }
            }
            // fragment length rule, _mapper.pyx:90-94
            int length = s1.end - s1.begin + K;
            if (length > 0) {
                if (length >= MAX_FRAGMENT_LENGTH) length = MAX_FRAGMENT_LENGTH - 1;
                atomicAdd(&fld_lds[length], 1u);
            }
        }
        // one arena allocation per wave: exclusive scan of the list lengths
        int n = active ? s1.n : 0;
        int scan = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(scan, d, 64);
            if (lane >= d) scan += up;
        }
        const int wave_total = __shfl(scan, 63, 64);
        unsigned long long wave_base = 0;
        if (lane == 63 && wave_total > 0)
            wave_base = atomicAdd(b.ids_cursor, (unsigned long long)wave_total);
        wave_base = __shfl(wave_base, 63, 64);
        if (active) {
            const int64_t off = (int64_t)wave_base + scan - n;
            uint64_t key = 0x243F6A8885A308D3ULL ^ (uint64_t)n;
            const bool fits = off + n <= b.ids_capacity;
            for (int i = 0; i < n; ++i) {
                const int32_t e = list1.get(i);
                if (fits) b.unit_entries[off + i] = e;
                key = tuple_key_step(key, (uint32_t)(e < 0 ? ~e : e));   // _get_ids, :533-536
            }
            if (key == 0) key = 1;
            b.unit_offset[u] = off;
            b.unit_count[u] = n;
            b.unit_key[u] = n ? key : 0;
            b.unit_begin[u] = s1.begin;
            b.unit_end[u] = s1.end;
            b.unit_anchor[u] = s1.anchor;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MAX_FRAGMENT_LENGTH; i += blockDim.x)
        if (fld_lds[i]) atomicAdd(&b.fld[i], (unsigned long long)fld_lds[i]);
    if (STATS) {
        unsigned long long *o = b.stats;
        atomicAdd(&o[0], (unsigned long long)n_reads);
        atomicAdd(&o[1], (unsigned long long)read_bases);
        atomicAdd(&o[2], (unsigned long long)ls.lookups);
        atomicAdd(&o[3], (unsigned long long)ls.slots);
        atomicAdd(&o[4], (unsigned long long)ls.contig_reads);
        atomicAdd(&o[5], (unsigned long long)ls.targets_copied);
        atomicAdd(&o[6], (unsigned long long)ls.targets_merged);
        atomicAdd(&o[7], (unsigned long long)ls.seq_fetches);
        atomicAdd(&o[8], (unsigned long long)ls.merges);
    }
}

void launch_pack_reads(const uint8_t *bases, const int64_t *offsets, int64_t n_reads,
                       int words_per_read, uint64_t *codes, uint32_t *acgt, hipStream_t stream)
{
    const int64_t total = n_reads * (int64_t)words_per_read;
    if (total == 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pack_reads_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       bases, offsets, n_reads, words_per_read, codes, acgt);
}

void launch_map_units(const DevIndex &ix, const MapBatch &b, int grid_blocks, bool stats,
                      hipStream_t stream)
{
    if (b.n_units == 0) return;
    if (stats)
        hipLaunchKernelGGL(map_units_kernel<true>, dim3(grid_blocks), dim3(256), 0, stream, ix, b);
    else
        hipLaunchKernelGGL(map_units_kernel<false>, dim3(grid_blocks), dim3(256), 0, stream, ix, b);
}

}  // namespace skm
