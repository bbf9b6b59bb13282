// Pseudoalignment kernels for gfx950: read packing and the contig-jumping
// mapper (one read or read pair per lane at a time, lanes refilled as they finish).
//
// What is computed is the reference's per-read state machine
// (/root/reference/seekmer/_mapper.pyx:111-343, 350-501) -- every branch is
// cited below -- but the data flow is built for CDNA4: reads are first packed
// by a coalesced streaming kernel to 2 bits per base + an "is upper-case
// ACGT" bit plane, k-mers and 8-base windows are then funnel-shifted out of
// registers instead of being re-encoded byte by byte, contig bases are
// fetched from a 2-bit pool, and the running target list of a lane lives in a
// small per-lane slice of an HBM workspace.
#include "skm_device.h"
#include "skm_kernels.h"

namespace skm {

// ---------------------------------------------------------------- pack_reads
// One lane per 32-base word of a read: ASCII -> (2-bit code word, 32-bit ACGT
// mask word).  The 32 bases come in as two (unaligned) 16-byte loads, so a
// wave reads 2 KiB of consecutive FASTQ bases per instruction pair.  Word
// index space is [n_reads][words_per_read].
struct __attribute__((packed, aligned(1))) Bytes16 { uint32_t w[4]; };

__device__ __forceinline__ void pack_dword(uint32_t w, int first, int n, uint64_t &codes, uint32_t &acgt)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = first + k;
        if (i < n) {
            const uint32_t ch = (w >> (8 * k)) & 0xffu;
            codes |= (uint64_t)two_bit_encode(ch) << (62 - 2 * i);
            const bool up = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
            acgt |= (uint32_t)up << (31 - i);
        }
    }
}

__global__ void __launch_bounds__(256)
pack_reads_kernel(const uint8_t *__restrict__ bases, const int64_t *__restrict__ offsets,
                  int64_t n_reads, int words_per_read,
                  uint64_t *__restrict__ codes, uint32_t *__restrict__ acgt)
{
    const int64_t total = n_reads * (int64_t)words_per_read;
    const int64_t end_of_bases = offsets[n_reads];
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
            const int64_t at = begin + first;
            if (at + 32 <= end_of_bases) {
                const Bytes16 lo = *reinterpret_cast<const Bytes16 *>(bases + at);
                const Bytes16 hi = *reinterpret_cast<const Bytes16 *>(bases + at + 16);
#pragma unroll
                for (int d = 0; d < 4; ++d) pack_dword(lo.w[d], 4 * d, n, c, m);
#pragma unroll
                for (int d = 0; d < 4; ++d) pack_dword(hi.w[d], 16 + 4 * d, n, c, m);
            } else {                                   // last bytes of the batch: stay in bounds
                for (int i = 0; i < n; ++i) pack_dword(bases[at + i], i, i + 1, c, m);
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

// The reference maps a read with nested loops (map_read -> _find_first_kmer /
// _filter_targets_to_left / _filter_targets_to_right, _mapper.pyx:151-343).
// On a 64-wide wave that shape is ruinous: the 200-instruction hash sits at
// seven call sites and the whole wave idles while a few lanes roll their
// first k-mer past a sequencing error or take the single retry.  Here every
// lane runs the SAME state machine, but explicitly: one round = at most one
// index lookup per lane at ONE call site, one list merge site, one SIFT4 site
// per direction; a lane that finishes its unit is refilled at once from its
// wave's private unit range, so all 64 lanes keep doing useful lookups.
// States that wait for a lookup result:
//   Y_FIRST  first-hit scan (_find_first_kmer)            Y_RA  right re-anchor (:283-284)
//   Y_LJ/Y_LS left junction / skip-a-k lookup (:247-263)  Y_RJ  right junction (:309-315)
// States that run without a lookup:
//   N_LEFT (:229-246, :270-275)  N_RIGHT_ENTER (:174-176)  N_RIGHT (:285-308, :335-343)
//   N_AFTER (:177-193)  N_MATE_DONE  UNIT_DONE (map_read_pair :129-144 + batch loop :89-94)
// Scheduling.  With 64 lanes spread over half a dozen heavy actions, running
// every action every round keeps each at ~15 % lane utilisation (measured:
// the kernel is VALU-issue bound, not memory bound).  So lanes WAIT in their
// state and an action is executed only when enough lanes want it (or when it
// is the most wanted and nothing else ran): convergence is restored by
// voting, the price being a longer per-unit latency that the refill hides.
enum : int { ST_IDLE = 0,
             Y_FIRST, Y_LJ, Y_LS, Y_RA, Y_RJ,          // want a lookup
             C_COPY,                                   // want map_contig
             M_LJ, M_LS, M_RJ,                         // want a list merge
             N_LEFT, N_RIGHT,                          // want an 8-base alignment step
             N_RIGHT_ENTER, N_AFTER, N_MATE_DONE,      // cheap transitions, run every round
             ST_UNIT_DONE };                           // want emission

constexpr int ARENA_CHUNK = 2048;     // ids a wave takes from the entry arena per atomic

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
    // each lane owns two contiguous lists (mate 1, mate 2): with lanes at
    // different points of their state machines nothing coalesces across lanes,
    // so a list must stay inside as few 64-byte sectors as possible
    const size_t stride = 1;
    int32_t *const ws1 = b.workspace + (size_t)gtid * 2 * (size_t)ix.max_target_count;
    int32_t *const ws2 = ws1 + ix.max_target_count;
    LaneStats ls = {0, 0, 0, 0, 0, 0, 0};
    uint64_t read_bases = 0, n_reads = 0, tuple_ids = 0;
    // scheduler census (STATS build): [0]=rounds, then executions / lanes per action
    uint32_t census[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // this wave's private range of units: no atomics for work distribution
    const int64_t n_waves = total_threads >> 6;
    const int64_t per_wave = (b.n_units + n_waves - 1) / n_waves;
    int64_t next = (gtid >> 6) * per_wave;
    const int64_t limit = min(b.n_units, next + per_wave);
    int64_t chunk_pos = 0, chunk_end = 0;      // wave-uniform slice of the entry arena
    const int th_lookup = b.vote[0], th_copy = b.vote[1], th_merge = b.vote[2],
              th_align = b.vote[3], th_emit = b.vote[4];

    // per-lane machine state
    int state = ST_IDLE;
    int64_t u = 0;
    int mate = 0, attempt = 0, scan_i = 0, len1 = 0;
    uint64_t kmer = 0;
    ReadView rv{b.codes, b.acgt, 0};
    Span span{0, 0, invalid_coord(), 0};
    Span s1 = span;

    for (;;) {
        // ---------------------------------------------------------- refill
        const unsigned long long idle = __ballot(state == ST_IDLE);
        if (idle != 0 && next < limit) {
            const int rank = __popcll(idle & ((1ULL << lane) - 1));
            if (state == ST_IDLE && next + rank < limit) {
                u = next + rank;
                mate = 0;
                attempt = 0;
                const int64_t r = b.paired ? 2 * u : u;
                rv.codes = b.codes + r * b.words_per_read;
                rv.acgt = b.acgt + r * b.words_per_read;
                rv.len = (int)(b.offsets[r + 1] - b.offsets[r]);
                span.begin = 0; span.end = 0; span.n = 0; span.anchor = invalid_coord();
                if (STATS) { read_bases += rv.len; n_reads++; }
                if (rv.len < K) {
                    state = N_MATE_DONE;              // shorter than k: unmapped (documented deviation)
                } else {
                    kmer = read_kmer(rv, 0);
                    scan_i = K;
                    state = Y_FIRST;
                }
            }
            next += __popcll(idle);
        }
        if (__ballot(state != ST_IDLE) == 0) break;

        // ------------------------------------------------ cheap transitions
        while (__ballot(state >= N_RIGHT_ENTER && state <= N_MATE_DONE) != 0) {
            if (state == N_RIGHT_ENTER) {                 // map_read, :174-176 / :190-192
                if (span.n != 0 && span.end < rv.len - K) {
                    kmer = read_kmer(rv, span.end);
                    state = Y_RA;
                } else {
                    state = N_AFTER;
                }
            } else if (state == N_AFTER) {
                if (span.n != 0 || attempt == 1) {
                    state = N_MATE_DONE;
                } else {                                  // the single retry, :179-185
                    attempt = 1;
                    span.anchor = invalid_coord();
                    span.begin += K;
                    if (span.begin + K > rv.len) span.begin = rv.len - K;
                    span.end = span.begin;
                    kmer = read_kmer(rv, span.begin);
                    scan_i = span.begin + K;
                    state = Y_FIRST;
                }
            } else if (state == N_MATE_DONE) {
                if (b.paired && mate == 0) {
                    s1 = span;
                    len1 = rv.len;
                    mate = 1;
                    attempt = 0;
                    const int64_t r = 2 * u + 1;
                    rv.codes = b.codes + r * b.words_per_read;
                    rv.acgt = b.acgt + r * b.words_per_read;
                    rv.len = (int)(b.offsets[r + 1] - b.offsets[r]);
                    span.begin = 0; span.end = 0; span.n = 0; span.anchor = invalid_coord();
                    if (STATS) { read_bases += rv.len; n_reads++; }
                    if (rv.len >= K) {
                        kmer = read_kmer(rv, 0);
                        scan_i = K;
                        state = Y_FIRST;
                    }                                     // else: stays N_MATE_DONE, closed next pass
                } else {
                    state = ST_UNIT_DONE;
                }
            }
        }

        // --------------------------------------------------------- the vote
        const int n_lookup = __popcll(__ballot(state >= Y_FIRST && state <= Y_RJ));
        const int n_copy = __popcll(__ballot(state == C_COPY));
        const int n_merge = __popcll(__ballot(state >= M_LJ && state <= M_RJ));
        const int n_left = __popcll(__ballot(state == N_LEFT));
        const int n_right = __popcll(__ballot(state == N_RIGHT));
        const int n_emit = __popcll(__ballot(state == ST_UNIT_DONE));
        bool do_lookup = n_lookup >= th_lookup, do_copy = n_copy >= th_copy,
             do_merge = n_merge >= th_merge, do_left = n_left >= th_align,
             do_right = n_right >= th_align, do_emit = n_emit >= th_emit;
        if (!(do_lookup || do_copy || do_merge || do_left || do_right || do_emit)) {
            // nothing reached its quorum: run the most wanted action
            const int best = max(max(max(n_lookup, n_copy), max(n_merge, n_left)), max(n_right, n_emit));
            if (n_lookup == best) do_lookup = true;
            else if (n_merge == best) do_merge = true;
            else if (n_left == best) do_left = true;
            else if (n_right == best) do_right = true;
            else if (n_copy == best) do_copy = true;
            else do_emit = true;
        }

        if (STATS) {
            census[0]++;
            if (do_lookup) { census[1]++; census[2] += n_lookup; }
            if (do_copy) { census[3]++; census[4] += n_copy; }
            if (do_merge) { census[5]++; census[6] += n_merge; }
            if (do_left) { census[7]++; census[8] += n_left; }
            if (do_right) { census[9]++; census[10] += n_right; }
            if (do_emit) { census[11]++; census[12] += n_emit; }
        }

        const TList list{mate ? ws2 : ws1, stride};

        // ------------------------------------------ the one index lookup site
        if (do_lookup && state >= Y_FIRST && state <= Y_RJ) {
            const Coord pos = map_kmer<STATS>(ix, kmer, &ls);
            span.anchor = pos;
            if (state == Y_FIRST) {                       // _find_first_kmer, :199-216
                if (pos.offset >= 0) {
                    span.begin = scan_i - K;
                    span.end = span.begin;
                    state = C_COPY;
                } else if (scan_i < rv.len) {
                    kmer = ((kmer << 2) | read_code(rv, scan_i)) & KMER_MASK;     // _kmer.append
                    ++scan_i;
                } else {
                    state = N_MATE_DONE;                  // no hit: returned as is, no retry
                }
            } else if (state == Y_RA) {
                state = N_RIGHT;
            } else if (pos.offset >= 0) {
                state = state == Y_LJ ? M_LJ : (state == Y_LS ? M_LS : M_RJ);
            } else if (state == Y_LJ) {                   // miss at the junction, :250-259
                if (span.begin < K) { span.begin = 0; state = N_RIGHT_ENTER; }
                else { span.begin -= K; kmer = read_kmer(rv, span.begin); state = Y_LS; }
            } else {                                      // Y_LS :260-263, Y_RJ :312-315
                span.n = 0;
                state = N_AFTER;
            }
        }

        // ---------------------------------------------- KMerIndex.map_contig
        if (do_copy && state == C_COPY) {
            map_contig<STATS>(ix, span.anchor, list, span, &ls);
            state = span.n == 0 ? N_MATE_DONE : (span.begin > 0 ? N_LEFT : N_RIGHT_ENTER);
        }

        // ------------------------------------- the one _filter_on_contig site
        if (do_merge && state >= M_LJ && state <= M_RJ) {
            const bool ok = filter_on_contig<STATS>(ix, list, span, &ls);
            if (state == M_LJ) {
                if (ok) state = N_LEFT;
                else if (span.begin < K) { span.begin = 0; state = N_RIGHT_ENTER; }
                else { span.begin -= K; kmer = read_kmer(rv, span.begin); state = Y_LS; }
            } else if (state == M_LS) {
                if (ok) state = N_LEFT; else { span.n = 0; state = N_AFTER; }
            } else {
                if (ok) state = N_RIGHT; else { span.n = 0; state = N_AFTER; }
            }
        }

        // ------------------- _filter_targets_to_left: loop head + alignment step
        if (do_left && state == N_LEFT) {
            const bool forward = span.anchor.entry >= 0;
            const int move = left_move(ix, span.anchor);
            if (STATS) ls.contig_reads++;
            const bool in_loop = span.begin > move;
            int at;
            if (in_loop) {
                span.begin -= move;
                span.anchor.offset -= forward ? move : -move;
                at = span.begin;
            } else {                                      // closing check, :270-275
                span.anchor.offset -= forward ? span.begin : -span.begin;
                at = 0;
            }
            const int shift = sift4_left(contig8<STATS>(ix, span.anchor, true, &ls), rv, at);
            if (!in_loop) {
                if (shift == INVALID_SHIFT) span.n = 0;
                state = N_RIGHT_ENTER;
            } else if (shift == INVALID_SHIFT || shift + 1 + move <= 0) {
                span.n = 0;
                state = N_AFTER;
            } else {
                span.begin -= shift + 1;
                if (span.begin < 0) {
                    span.begin = 0;
                    state = N_RIGHT_ENTER;
                } else {
                    kmer = (tail_kmer<STATS>(ix, span.anchor, &ls) >> 2)          // _kmer.prepend
                           | ((uint64_t)read_code(rv, span.begin) << (2 * K - 2));
                    state = Y_LJ;
                }
            }
        }

        // ------------------ _filter_targets_to_right: loop head + alignment step
        if (do_right && state == N_RIGHT) {
            const bool forward = span.anchor.entry >= 0;
            const int move = right_move(ix, span.anchor);
            if (STATS) ls.contig_reads++;
            const int rest = rv.len - span.end - K;
            const bool in_loop = rest > move;
            int at;
            if (in_loop) {
                span.end += move;
                span.anchor.offset += forward ? move : -move;
                at = span.end + K - ALIGN_LENGTH;
            } else {                                      // closing check, :335-343
                span.anchor.offset += forward ? rest : -rest;
                at = rv.len - ALIGN_LENGTH;
            }
            const int shift = sift4_right(contig8<STATS>(ix, span.anchor, false, &ls), rv, at);
            if (!in_loop) {
                if (shift == INVALID_SHIFT) span.n = 0;
                state = N_AFTER;
            } else if (shift == INVALID_SHIFT || shift + 1 + move <= 0) {
                span.n = 0;
                state = N_AFTER;
            } else {
                span.end += shift + 1;
                if (span.end + K > rv.len) {
                    span.end = rv.len - K;
                    state = N_AFTER;
                } else {
                    kmer = ((tail_kmer<STATS>(ix, span.anchor, &ls) << 2)          // _kmer.append
                            | read_code(rv, span.end + K - 1)) & KMER_MASK;
                    state = Y_RJ;
                }
            }
        }

        // --------------------------------------------------- finished units
        if (!do_emit) continue;
        int n_out = 0;
        if (state == ST_UNIT_DONE) {
            if (b.paired) {
                // map_read_pair, _mapper.pyx:129-144 (span = mate 2, s1 = mate 1)
                const TList l1{ws1, stride}, l2{ws2, stride};
                const Span s2 = span;
                if (!intersect(l1, s1, l2, s2)) {
                    s1.n = 0;
                    s1.begin = 0;
                    s1.end = -K;
                } else if (s1.anchor.entry != ~s2.anchor.entry) {
                    s1.begin = 0;
                    s1.end = -K;
                } else {
                    int interval = s2.anchor.offset - s1.anchor.offset;
                    if (s1.anchor.entry < 0) interval = -interval;
                    s1.end = (len1 - K) + interval + (rv.len - K) - s2.begin;
                }
                span = s1;
            }
            // fragment length rule, _mapper.pyx:90-94
            int length = span.end - span.begin + K;
            if (length > 0) {
                if (length >= MAX_FRAGMENT_LENGTH) length = MAX_FRAGMENT_LENGTH - 1;
                atomicAdd(&fld_lds[length], 1u);
            }
            n_out = span.n;
        }
        // one slice of the entry arena for all lanes that finish this round
        int scan = n_out;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(scan, d, 64);
            if (lane >= d) scan += up;
        }
        const int wave_total = __shfl(scan, 63, 64);
        if (chunk_pos + wave_total > chunk_end) {
            const int64_t want = wave_total > ARENA_CHUNK ? wave_total : ARENA_CHUNK;
            unsigned long long got = 0;
            if (lane == 0) got = atomicAdd(b.ids_cursor, (unsigned long long)want);
            chunk_pos = (int64_t)__shfl(got, 0, 64);
            chunk_end = chunk_pos + want;
        }
        if (state == ST_UNIT_DONE) {
            const int64_t off = chunk_pos + scan - n_out;
            uint64_t key = 0x243F6A8885A308D3ULL ^ (uint64_t)n_out;
            const bool fits = off + n_out <= b.ids_capacity;
            for (int i = 0; i < n_out; ++i) {
                const int32_t e = ws1[(size_t)i * stride];
                if (fits) b.unit_entries[off + i] = e;
                key = tuple_key_step(key, (uint32_t)(e < 0 ? ~e : e));   // _get_ids, :533-536
            }
            if (key == 0) key = 1;
            b.unit_offset[u] = off;
            b.unit_count[u] = n_out;
            b.unit_key[u] = n_out ? key : 0;
            b.unit_begin[u] = span.begin;
            b.unit_end[u] = span.end;
            b.unit_anchor[u] = span.anchor;
            if (STATS) tuple_ids += n_out;
            state = ST_IDLE;
        }
        chunk_pos += wave_total;
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
        atomicAdd(&o[9], (unsigned long long)tuple_ids);
        if (lane == 0)
            for (int i = 0; i < 13; ++i) atomicAdd(&o[16 + i], (unsigned long long)census[i]);
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
