// Pseudoalignment kernels for gfx950: read packing and the contig-jumping
// mapper (one read or read pair per lane at a time, lanes refilled as they finish).
//
// What is computed is the reference's per-read state machine
// (/root/reference/seekmer/_mapper.pyx:111-343, 350-501) -- every branch is
// cited below -- but the data flow is built for CDNA4: reads are first packed
// by a coalesced streaming kernel to 2 bits per base + an "is upper-case
// ACGT" bit plane, k-mers and 8-base windows are then funnel-shifted out of
// registers instead of being re-encoded byte by byte, contig bases are
// fetched from a 2-bit pool, the k-mer table is probed through a bucketised
// device copy (skm_device.h: DevBucket), and the running target list of a unit
// is never materialised: it is a slice of the index plus a keep-mask.
#include "../../include/seekmer_hip.h"
#include "skm_device.h"
#include "skm_kernels.h"

namespace skm {

// ---------------------------------------------------------------- pack_reads
// One lane per 32-base word of a read: ASCII -> (2-bit code word, 32-bit ACGT
// mask word).  The 32 bases come in as two (unaligned) 16-byte loads, so a
// wave reads 2 KiB of consecutive FASTQ bases per instruction pair.  Word
// index space is [n_reads][words_per_read].
struct __attribute__((packed, aligned(1))) Bytes16 { uint32_t w[4]; };

// one base (tail of the batch only)
__device__ __forceinline__ void pack_byte(uint32_t ch, int i, uint64_t &codes, uint32_t &acgt)
{
    codes |= (uint64_t)two_bit_encode(ch) << (62 - 2 * i);
    const bool up = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
    acgt |= (uint32_t)up << (31 - i);
}

// four bases at once (bases `first` .. `first`+3 of the word, `first` a multiple
// of 4; only those below `n` count).  SWAR: (c >> 1) & 3 is 0,1,2,3 for A,C,T,G
// in either case, a byte permute looks the letter up again to tell ACGT from
// everything else, and two multiplies gather the 2-bit codes / flag bits.
__device__ __forceinline__ void pack_dword(uint32_t w, int first, int n, uint64_t &codes, uint32_t &acgt)
{
    const int cnt = n - first;
    if (cnt <= 0) return;
    const uint32_t keep = cnt >= 4 ? 0xffffffffu : ((1u << (8 * cnt)) - 1u);
    const uint32_t up = w & 0xDFDFDFDFu;                              // fold case
    const uint32_t idx = (up >> 1) & 0x03030303u;
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, idx);   // 'A','C','T','G' by idx
    const uint32_t t = up ^ expect;
    uint32_t valid = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);   // 0x80 where ACGT/acgt
    valid &= keep;
    uint32_t code = idx ^ ((idx >> 1) & 0x01010101u);                 // A0 C1 G2 T3 (_kmer.pxd:253-273)
    code &= (valid >> 7) * 3u;                                        // anything else encodes as 0
    const uint32_t upper = valid & ~((w << 2) & 0x80808080u);         // and is upper case
    const uint32_t code_byte = (code * 0x40100401u) >> 24;            // first base in the top bits
    const uint32_t flag_nibble = (((upper >> 7) * 0x08040201u) >> 24) & 0xFu;
    const int d = first >> 2;
    codes |= (uint64_t)code_byte << (56 - 8 * d);
    acgt |= flag_nibble << (28 - 4 * d);
}

// The same four bases when they are expected to be upper-case ACGT (nearly every word of a
// FASTQ file): the byte of their four codes in the TOP byte of the result, and `bad` collects
// anything that is not such a letter -- a word with bad == 0 has the all-ones flag pattern and
// needs no more work; any other word is redone by pack_dword.  No count: the caller cuts the
// word to the read's length afterwards (what lies behind the read's end in the 32 bytes are the
// next read's bases; should THEY hold something else, the word takes the exact path for nothing).
__device__ __forceinline__ uint32_t pack_dword_plain(uint32_t w, uint32_t &bad)
{
    const uint32_t idx = (w >> 1) & 0x03030303u;
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, idx);      // 'A','C','T','G' by idx
    bad |= w ^ expect;
    const uint32_t code = idx ^ ((idx >> 1) & 0x01010101u);                   // A0 C1 G2 T3
    return code * 0x40100401u;                                                // first base in the top bits of byte 3
}
// the top bytes of four such products as one word, the first on top
__device__ __forceinline__ uint32_t top_bytes(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3)
{
    const uint32_t a = __builtin_amdgcn_perm(p0, p1, 0x07030000u);            // byte 3 of p0, byte 3 of p1, -, -
    const uint32_t b = __builtin_amdgcn_perm(p2, p3, 0x00000703u);            // -, -, byte 3 of p2, byte 3 of p3
    return __builtin_amdgcn_perm(a, b, 0x07060100u);
}

// (4.2 TB/s of the ~6.3 a copy reaches: two, or four, words per lane with their loads issued phase
// by phase -- offsets, then bases -- ran in the same time, and so did this fast path with half the
// instructions of the one before it: profiles/r04_ab_map.log)
__global__ void __launch_bounds__(256)
pack_reads_kernel(const uint8_t *__restrict__ bases, const int64_t *__restrict__ offsets,
                  int64_t n_reads, int words_per_read, int record_words, uint32_t *__restrict__ records)
{
    const int64_t total = n_reads * (int64_t)words_per_read;
    const int64_t end_of_bases = offsets[n_reads];
    const bool shifting = (words_per_read & (words_per_read - 1)) == 0;      // (4 words for reads of up to 128 bases)
    const int shift = __builtin_ctz((unsigned)words_per_read);
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        // (a shift, or a 32-bit division when the index fits: the 64-bit one is a long software routine)
        const int64_t r = shifting ? g >> shift
                                   : total <= 0xffffffffLL ? (int64_t)((uint32_t)g / (uint32_t)words_per_read)
                                                           : g / words_per_read;
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
                uint32_t bad = 0;
                const uint32_t first16 = top_bytes(pack_dword_plain(lo.w[0], bad), pack_dword_plain(lo.w[1], bad),
                                                   pack_dword_plain(lo.w[2], bad), pack_dword_plain(lo.w[3], bad));
                const uint32_t last16 = top_bytes(pack_dword_plain(hi.w[0], bad), pack_dword_plain(hi.w[1], bad),
                                                  pack_dword_plain(hi.w[2], bad), pack_dword_plain(hi.w[3], bad));
                if (bad == 0) {
                    m = n >= 32 ? 0xffffffffu : ~(0xffffffffu >> n);      // n upper-case ACGT bases
                    c = (((uint64_t)first16 << 32) | last16) & (n >= 32 ? ~0ULL : ~(~0ULL >> (2 * n)));
                } else {                                   // N, lower case, anything else: the full rule
#pragma unroll
                    for (int d = 0; d < 4; ++d) pack_dword(lo.w[d], 4 * d, n, c, m);
#pragma unroll
                    for (int d = 0; d < 4; ++d) pack_dword(hi.w[d], 16 + 4 * d, n, c, m);
                }
            } else {                                   // last bytes of the batch: stay in bounds
                for (int i = 0; i < n; ++i) pack_byte(bases[at + i], i, c, m);
            }
        }
        uint32_t *rec = records + r * (int64_t)record_words;
        reinterpret_cast<uint64_t *>(rec)[w] = c;
        rec[2 * words_per_read + w] = m;
        if (w == 0) rec[3 * words_per_read] = (uint32_t)len;
        // the record's padding too, so that its sectors leave the L2 whole (0.76 against 0.78 ms;
        // nobody reads these words)
        for (int k = 3 * words_per_read + 1 + w; k < record_words; k += words_per_read) rec[k] = 0;
    }
}

// ------------------------------------------------------------- unpack_reads
// Reads that arrive already packed by the host (skm_packed_reads: code words only): one lane per
// (read, record word) writes the 64-byte record the map kernel reads -- codes as they came, the
// ACGT bit plane all ones up to the read's length, the length.  The destination stride places
// mate m of every unit (paired batches interleave the mates: record 2 u + m).
__global__ void __launch_bounds__(256)
unpack_reads_kernel(const uint64_t *__restrict__ codes, int64_t stride, int code_words,
                    const uint32_t *__restrict__ lengths, uint32_t uniform_len, int64_t n_reads,
                    int words_per_read, uint32_t *__restrict__ dst, int64_t dst_stride, int *error)
{
    const int64_t total = n_reads * (int64_t)words_per_read;
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = total <= 0xffffffffLL ? (int64_t)((uint32_t)g / (uint32_t)words_per_read)
                                                : g / words_per_read;
        const int w = (int)(g - r * words_per_read);
        uint32_t len = lengths ? lengths[r] : uniform_len;
        if (len > 32u * (uint32_t)code_words) {           // (the host promised otherwise)
            len = 32u * (uint32_t)code_words;
            if (w == 0) atomicExch(error, SKM_ERR_ARG);
        }
        const uint64_t c = w < code_words ? codes[r * stride + w] : 0;
        const int n = (int)len - 32 * w;
        const uint32_t m = n >= 32 ? 0xffffffffu : (n <= 0 ? 0u : ~(0xffffffffu >> n));
        uint32_t *rec = dst + r * dst_stride;
        reinterpret_cast<uint64_t *>(rec)[w] = c;
        rec[2 * words_per_read + w] = m;
        if (w == 0) rec[3 * words_per_read] = len;
    }
}

// the reads that hold a character other than upper-case ACGT get their own bit plane
__global__ void __launch_bounds__(256)
unpack_exceptions_kernel(const uint32_t *__restrict__ exc_reads, const uint32_t *__restrict__ exc_masks,
                         int64_t n_exceptions, int code_words, int64_t first_read, int words_per_read,
                         uint32_t *__restrict__ dst, int64_t dst_stride)
{
    const int64_t total = n_exceptions * (int64_t)code_words;
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = g / code_words;
        const int w = (int)(g - e * code_words);
        const int64_t r = (int64_t)exc_reads[e] - first_read;
        if (w < words_per_read) dst[r * dst_stride + 2 * words_per_read + w] = exc_masks[g];
    }
}

// A host batch's offsets as they arrived in HBM: rebase to the first byte copied, find the
// longest read and count the places where the offsets step backwards (out[0] = max length,
// out[1] = descents) -- the host then never walks the n_reads + 1 offsets itself.
__global__ void __launch_bounds__(256)
offsets_scan_kernel(int64_t *offsets, int64_t n_reads, int64_t base, unsigned long long *out)
{
    long long longest = 0;
    unsigned long long descents = 0;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_reads;
         r += (int64_t)gridDim.x * blockDim.x) {
        const long long len = offsets[r + 1] - offsets[r];
        if (len < 0) ++descents;
        longest = max(longest, len);
    }
    for (int d = 32; d > 0; d >>= 1) {
        longest = max(longest, __shfl_xor(longest, d, 64));
        descents += __shfl_xor(descents, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&out[0], (unsigned long long)longest);
        if (descents) atomicAdd(&out[1], descents);
    }
}
// offsets of a batch whose reads all have the same length: never sent over PCIe
__global__ void __launch_bounds__(256)
offsets_uniform_kernel(int64_t *offsets, int64_t n, int64_t read_len)
{
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n;
         r += (int64_t)gridDim.x * blockDim.x) offsets[r] = r * read_len;
}
__global__ void __launch_bounds__(256)
offsets_rebase_kernel(int64_t *offsets, int64_t n, int64_t base)
{
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n;
         r += (int64_t)gridDim.x * blockDim.x) offsets[r] -= base;
}

// ------------------------------------------------- bucket table construction
__global__ void __launch_bounds__(256)
bucket_init_kernel(DevBucketBuild *buckets, uint64_t n_buckets)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_buckets * 4;
         i += (uint64_t)gridDim.x * blockDim.x) {
        buckets[i >> 2].kmer[i & 3] = KMER_INVALID;
        buckets[i >> 2].pos[i & 3] = Coord{0, -1};
    }
}

// every occupied slot of the reference table goes into the first free entry from its bucket
// on; report[0] = entries placed, [1] = placed outside their home bucket, [2] = k-mers met
// twice or with bits above the 2k a k-mer has (the table is not a set of k-mers: the caller
// drops the bucket table)
__global__ void __launch_bounds__(256)
bucket_fill_kernel(const IndexEntry *__restrict__ kmers, uint64_t n_slots, DevBucketBuild *buckets,
                   uint32_t bucket_mask, uint32_t bucket_shift, unsigned long long *report)
{
    unsigned long long placed = 0, moved = 0, twice = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_slots;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const IndexEntry e = kmers[i];
        if (e.kmer == KMER_INVALID) continue;
        if (e.kmer & ~KMER_MASK) { ++twice; continue; }
        const uint64_t rc = kmer_revcomp(e.kmer);
        const uint32_t home = bucket_hash(e.kmer < rc ? e.kmer : rc) >> bucket_shift;
        uint32_t b = home;
        bool done = false;
        for (uint32_t n = 0; n <= bucket_mask && !done; ++n) {
            for (int j = 0; j < 4 && !done; ++j) {
                const unsigned long long old = atomicCAS(
                    reinterpret_cast<unsigned long long *>(&buckets[b].kmer[j]), KMER_INVALID, e.kmer);
                if (old == KMER_INVALID) {
                    buckets[b].pos[j] = e.pos;
                    ++placed;
                    if (b != home) ++moved;
                    done = true;
                } else if (old == e.kmer) {
                    ++twice;
                    done = true;
                }
            }
            b = (b + 1) & bucket_mask;
        }
    }
    if (placed) atomicAdd(&report[0], placed);
    if (moved) atomicAdd(&report[1], moved);
    if (twice) atomicAdd(&report[2], twice);
}

// The filled buckets in the form the mapper reads (DevBucket), in place: a lane per bucket.
__global__ void __launch_bounds__(256)
bucket_pack_kernel(DevBucketBuild *buckets, uint64_t n_buckets)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_buckets;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t stored[4];
        for (int j = 0; j < 4; ++j) stored[j] = buckets[i].kmer[j];
        DevBucket *packed = reinterpret_cast<DevBucket *>(&buckets[i]);
        for (int j = 0; j < 4; ++j) {
            uint32_t low = BUCKET_FREE, high = BUCKET_FREE;
            if (stored[j] != KMER_INVALID) {
                const uint64_t rc = kmer_revcomp(stored[j]);
                const uint64_t canonical = stored[j] < rc ? stored[j] : rc;
                low = (uint32_t)canonical & BUCKET_LOW_MASK;
                high = (uint32_t)(canonical >> 31) | (stored[j] != canonical ? BUCKET_STORED_RC : 0u);
            }
            packed->low[j] = low;
            packed->high[j] = high;
        }
    }
}

// The signatures of the k-mers in the packed buckets (skm_device.h: kmer_min_hash): a lane per entry.
__global__ void __launch_bounds__(256)
signature_build_kernel(const DevBucket *buckets, uint64_t n_buckets, unsigned long long *signatures, uint32_t shift)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_buckets * 4;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const DevBucket &bucket = buckets[i >> 2];
        const uint32_t low = bucket.low[i & 3];
        if (low == BUCKET_FREE) continue;
        const uint64_t canonical = ((uint64_t)(bucket.high[i & 3] & BUCKET_HIGH_MASK) << 31) | low;
        const Signature bits = signature_bits(bucket_hash(canonical));
        unsigned long long *slot = signatures + 2 * (size_t)signature_slot(kmer_min_hash(canonical), shift);
        if (bits.lo) atomicOr(slot, bits.lo);
        if (bits.hi) atomicOr(slot + 1, bits.hi);
    }
}

// The bucket table answers "is this k-mer in the set, and with which position"; the
// reference's probe (_common.pyx:75-97) answers the same question exactly when it finds every
// stored k-mer in the slot that stores it.  report[3] counts the slots where it does not
// (an empty slot or an equal k-mer between the home slot and the slot).
__global__ void __launch_bounds__(256)
probe_check_kernel(DevIndex ix, uint64_t n_slots, unsigned long long *report)
{
    unsigned long long bad = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_slots;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t kmer = ix.kmers[i].kmer;
        if (kmer == KMER_INVALID) continue;
        const uint64_t rc = kmer_revcomp(kmer);
        uint32_t slot = kmer_hash(kmer < rc ? kmer : rc) & ix.slot_mask;
        bool reached = false;
        for (uint64_t n = 0; n <= ix.slot_mask; ++n) {
            const uint64_t stored = ix.kmers[slot].kmer;
            if (stored == KMER_INVALID) break;
            if (stored == kmer || stored == rc) { reached = slot == (uint32_t)i; break; }
            slot = (slot + 1) & ix.slot_mask;
        }
        if (!reached) ++bad;
    }
    if (bad) atomicAdd(&report[3], bad);
}

// The junction successors of every contig record (DevSide::succ), by the device's own lookup over
// the bucket table that has just passed its check: eight lanes per contig.  `force_lookup` (a test
// hook) marks every successor that exists SUCC_LOOKUP, so that the hops take the fall-back path.
__global__ void __launch_bounds__(256)
successor_build_kernel(DevIndex ix, DevContig *records, int64_t n_contigs, int force_lookup)
{
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n_contigs * 8;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = g >> 3;
        const int side = (int)(g & 7) >> 2, b = (int)(g & 3);
        bool found;
        const Coord pos = map_kmer_buckets(ix, successor_query(ix.edge_kmers[2 * c], ix.edge_kmers[2 * c + 1],
                                                               records[c].side[0].length, side, b), found);
        // (a k-mer stored without a position, offset < 0, is a miss that leaves its own trace in the
        // unit's anchor: SURVEY A6 -- left to the lookup)
        uint32_t kind = found ? SUCC_LOOKUP : SUCC_ABSENT;
        if (found && pos.offset >= 0 && !force_lookup) {
            const int64_t landing = pos.entry < 0 ? ~pos.entry : pos.entry;
            if (landing < n_contigs) {
                if (pos.offset == 0) kind = SUCC_AT_START;
                else if (pos.offset == records[landing].side[0].length - K) kind = SUCC_AT_END;
            }
        }
        const bool placed = kind == SUCC_AT_START || kind == SUCC_AT_END;
        // SUCC_WHOLE: the landing contig's list holds every entry of this contig's (as multisets, in
        // the orientation the hop arrives in).  A read's running list is a part of the list of the
        // contig its anchor is on, so _filter_on_contig on the landing contig (_common.pyx:185-235)
        // keeps all of it: the hop needs no merge.  Ascending slices only (two-pointer check).
        bool whole = false, masked = false;
        uint32_t kept = 0;
#ifndef SKM_NO_WHOLE
        if (placed && ix.sorted_targets) {
            const int64_t landing = pos.entry < 0 ? ~pos.entry : pos.entry;
            const int32_t *all = reinterpret_cast<const int32_t *>(records);
            auto slice = [&](int64_t contig, int32_t &start, int32_t &length) {
                const DevSide &s0 = records[contig].side[0];
                const int32_t count = (int32_t)(s0.count_edge >> 16);
                if (count <= CONTIG_INLINE_TARGETS) { start = (int32_t)(contig * (2 * SIDE_WORDS) + SIDE_TARGETS_WORD); length = count; }
                else { start = s0.targets[0]; length = s0.targets[1]; }
            };
            int32_t from, n_from, to, n_to;
            slice(c, from, n_from);
            slice(landing, to, n_to);
            // the landing list in the arriving orientation, ascending
            auto at = [&](int32_t k) { return pos.entry >= 0 ? all[to + k] : ~all[to + n_to - 1 - k]; };
            if (n_from <= n_to && n_to <= 4096) {
                whole = true;
                int32_t j = 0;
                for (int32_t i = 0; i < n_from && whole; ++i) {
                    const int32_t want = all[from + i];
                    while (j < n_to && at(j) < want) ++j;
                    if (j < n_to && at(j) == want) ++j; else whole = false;
                }
            }
            // SUCC_MASKED: otherwise, for a short list without a transcript listed twice (which copies
            // of one the walk keeps depends on the direction it is read in), the entries it keeps --
            // all a read needs whose running list is still a part of THIS contig's list
#ifndef SKM_NO_MASKED
            if (!whole && n_from <= CONTIG_INLINE_TARGETS && n_to <= 4096) {
                masked = true;
                for (int32_t i = 1; i < n_from; ++i) masked = masked && all[from + i] != all[from + i - 1];
                int32_t j = 0;
                for (int32_t i = 0; i < n_from && masked; ++i) {
                    const int32_t want = all[from + i];
                    while (j < n_to && at(j) < want) ++j;
                    if (j < n_to && at(j) == want) { kept |= 1u << i; ++j; }
                }
            }
#endif
        }
#endif
        records[c].side[side].kept[b] = (uint8_t)(masked ? kept : 0u);
        records[c].side[side].succ[b] = (int32_t)((placed ? ((uint32_t)pos.entry << SUCC_ENTRY_SHIFT) | (whole ? SUCC_WHOLE : 0u)
                                                            | (masked ? SUCC_MASKED : 0u) : 0u) | kind);
    }
}

// ------------------------------------------------------------------- mapper
// Running target list of a context.  KMerIndex.map_contig (_common.pyx:143-179)
// copies the first contig's target slice (reversed and complemented for a
// reverse hit) and every later step only deletes entries from it
// (_filter_on_contig, _intersect).  So the list is kept as
//   (start, length, orientation of the slice in ix.targets) + a keep-mask,
// element i of the reference's list being
//   forward:  targets[start + i].entry        reverse: ~targets[start + length - 1 - i].entry
// Nothing is copied or stored while mapping; merges are read-only walks over
// the index (L1-cacheable).  Word 0 of the mask lives with the context, words
// 1.. (slices longer than 64 targets) in a per-context HBM extension with a
// staging copy of the same size behind it.
// targets[i] through a 32-bit byte offset from the (wave-uniform) base: one
// global_load with scalar base + vector offset instead of a 64-bit address
// computation per element (skm_index_create bounds n_targets by 2^30)
__device__ __forceinline__ int32_t target_at(const DevIndex &ix, int32_t i)
{
    return *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(ix.targets)
                                              + ((uint32_t)i << 2));
}

struct TSet {
    int32_t start;            // first element of the slice in ix.targets
    int32_t length;           // slice length = list positions 0 .. length-1
    bool forward;
    uint64_t word0;
    uint64_t *ext;            // words 1.. (live), then the staging words
    int32_t ext_words;        // capacity of each of the two regions

    __device__ __forceinline__ int words() const { return (length + 63) >> 6; }
    __device__ __forceinline__ uint64_t word(int w) const { return w == 0 ? word0 : ext[w - 1]; }
    __device__ __forceinline__ void set_word(int w, uint64_t v) { if (w == 0) word0 = v; else ext[w - 1] = v; }
    __device__ __forceinline__ int32_t entry(const DevIndex &ix, int i) const
    {
        return forward ? target_at(ix, start + i) : ~target_at(ix, start + length - 1 - i);
    }
};

struct Span {               // MappedSpan, _common.pxd:31-35 (targets = TSet, n = its size)
    int32_t begin, end;
    Coord anchor;
    int32_t n;
};

// KMerIndex.map_contig, _common.pyx:143-179: the list is the whole slice
// (from the side of the record that the step after the first hit reads: SideVisit)
template <bool STATS>
__device__ __forceinline__ void map_contig(const DevIndex &ix, Coord c, const SideVisit &at_side, TSet &set, Span &span,
                                           LaneStats *st)
{
    const bool forward = c.entry >= 0;
    set.start = at_side.slice.start;
    int32_t length = at_side.slice.length;
    if (length > ix.max_target_count) length = ix.max_target_count;
    set.length = length;
    set.forward = forward;
    if (STATS) { st->contig_reads++; st->targets_copied += length; }
    span.n = length;
    set.word0 = 0;
    const int words = set.words();
    for (int w = 0; w < words; ++w) {
        const int bits = min(64, length - 64 * w);
        set.set_word(w, bits == 64 ? ~0ULL : ((1ULL << bits) - 1));
    }
}

// ---- short lists in registers --------------------------------------------
// The two-pointer walks below cost one dependent load per step and the wave
// waits for its longest lane (measured: 44k cycles per emission round in
// _intersect alone, 12k per index lookup round).  Every slice of a built index
// is sorted by signed entry (the builder sorts (contig, entry, offset);
// skm_index_create verifies it, DevIndex.sorted_targets), so both walks are
// multiset intersections that keep the earliest duplicates of the left list --
// and for slices of at most LIST_REGS targets that is computed on registers:
// both lists are fetched with independent loads (one round trip) and compared
// all-pairs with fully unrolled code.  Longer slices, unsorted indices and the
// counting build take the walks.
constexpr int LIST_REGS = 16;
constexpr uint32_t LIST_ALL = (1u << LIST_REGS) - 1u;
static_assert(LIST_REGS % 2 == 0 && LIST_REGS <= 16, "keep_common pairs the right-hand entries");
#define LIST_FAST(count) (!(count))
constexpr int32_t NO_ENTRY = INT32_MIN;      // (would be transcript 2^31-1: cannot occur, n_targets < 2^30)
constexpr int32_t NO_ENTRY_B = INT32_MIN + 1; // the same for the other side of a comparison: never equal to NO_ENTRY

// entries at list positions 0..15 of a slice of 1..16 targets; `absent` where the position is
// outside the slice or cleared in `keep`.  `twice` = two neighbours of the (whole, ascending)
// slice are equal: a transcript listed twice (two such classes in the reference's chr21 data).
__device__ __forceinline__ void load_list(const DevIndex &ix, int32_t start, int32_t length, bool forward,
                                          uint32_t keep, int32_t absent, int32_t (&e)[LIST_REGS], bool &twice)
{
    const int last = length - 1;
    int32_t v[LIST_REGS];
#pragma unroll
    for (int i = 0; i < LIST_REGS; ++i) {
        const int p = min(i, last);
        const int32_t raw = target_at(ix, forward ? start + p : start + last - p);
        v[i] = forward ? raw : ~raw;
    }
    bool same = false;
#pragma unroll
    for (int i = 1; i < LIST_REGS; ++i) same |= (i <= last) & (v[i] == v[i - 1]);
    twice = same;
#pragma unroll
    for (int i = 0; i < LIST_REGS; ++i) e[i] = (i <= last && ((keep >> i) & 1u)) ? v[i] : absent;
}
__device__ __forceinline__ void load_list(const DevIndex &ix, int32_t start, int32_t length, bool forward,
                                          uint32_t keep, int32_t (&e)[LIST_REGS])
{
    bool twice;
    load_list(ix, start, length, forward, keep, NO_ENTRY, e, twice);
}

// positions of `a` (ascending, NO_ENTRY = absent) that a two-pointer walk against the
// multiset `b` (NO_ENTRY_B = absent) keeps: the r-th copy of a value survives iff b holds more
// than r copies
template <int N>
__device__ __forceinline__ uint32_t keep_common_exact(const int32_t (&a)[N], const int32_t (&b)[N])
{
    uint32_t keep = 0;
    int32_t prev = NO_ENTRY;
    int run = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int32_t v = a[i];
        int copies = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) copies += (b[j] == v) ? 1 : 0;
        const bool present = v != NO_ENTRY;
        run = (present && v == prev) ? run + 1 : (present ? 0 : run);
        if (present) prev = v;
        if (present && run < copies) keep |= 1u << i;
    }
    return keep;
}
// The same when no value occurs twice on either side (`twice` of load_list, all but a handful
// of slices): a position survives iff its value occurs in b.  All pairs on the vector ALU only:
// the smallest a[i] ^ b[j] over j is 0 iff a[i] is in b -- v_xor + half a v_min3_u32 per pair,
// no lane masks through scalar registers (compare + s_or was two instructions per pair and
// compare + add-with-carry costs two wait states per pair on gfx950).
template <int N>
__device__ __forceinline__ uint32_t keep_common(const int32_t (&a)[N], const int32_t (&b)[N], bool twice)
{
    if (twice) return keep_common_exact<N>(a, b);
    uint32_t dropped = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint32_t nearest = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < N; j += 2)
            nearest = min(nearest, min((uint32_t)(a[i] ^ b[j]), (uint32_t)(a[i] ^ b[j + 1])));
        dropped |= min(nearest, 1u) << i;                 // 1 = a[i] is not in b
    }
    return ~dropped & ((1u << N) - 1u);
}

// KMerIndex._filter_on_contig, _common.pyx:185-235: two-pointer merge of the
// list (ascending signed entries) with the anchor contig's slice; an empty
// intersection leaves the list as it was and returns false.
// `right`: the merge belongs to a hop to the right; the step that follows reads the same side of the
// record.  A successor that landed on its contig's last k-mer gets its offset here (OFFSET_AT_END).
template <bool STATS>
__device__ __forceinline__ bool filter_on_contig(const DevIndex &ix, bool right, TSet &set, Span &span, LaneStats *st)
{
    if (STATS) st->merges++;
    const bool forward = span.anchor.entry >= 0;
    const int32_t contig = forward ? span.anchor.entry : ~span.anchor.entry;
    const int side = visit_side(right, forward);
    const DevSide &record = side_at(ix, contig, side);
    if (span.anchor.offset == OFFSET_AT_END) span.anchor.offset = record.length - K;
    if (span.n == 0) return true;
    const Slice slice = side_slice(record, contig, side);
    const int32_t start = slice.start;
    int32_t length = slice.length;
    if (length > ix.max_target_count) length = ix.max_target_count;
    if (STATS) st->contig_reads++;
    if (LIST_FAST(STATS) && ix.sorted_targets && set.length <= LIST_REGS && length <= LIST_REGS) {
        if (length == 0) return false;
        int32_t a[LIST_REGS], t[LIST_REGS];
        bool twice_a, twice_t;
        load_list(ix, set.start, set.length, set.forward, (uint32_t)set.word0, NO_ENTRY, a, twice_a);
        load_list(ix, start, length, forward, LIST_ALL, NO_ENTRY_B, t, twice_t);   // (order within the slice is immaterial)
        const uint32_t keep = keep_common<LIST_REGS>(a, t, twice_a | twice_t);
        if (keep == 0) return false;
        set.word0 = keep;
        span.n = __builtin_popcount(keep);
        return true;
    }
    int track = forward ? start : start + length - 1;
    const int bound = forward ? start + length : start - 1;
    const int step = forward ? 1 : -1;
    const int first_track = track;
    const int words = set.words();
    uint64_t *const stage = set.ext + set.ext_words;
    uint64_t new0 = 0;
    int kept = 0;
    for (int w = 0; w < words; ++w) {
        uint64_t m = track != bound ? set.word(w) : 0, keep = 0;
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            const int32_t target_entry = set.entry(ix, 64 * w + bit);
            int32_t index_entry = 0;
            while (track != bound) {                 // skip index entries below the list entry
                index_entry = target_at(ix, track);
                if (!forward) index_entry = ~index_entry;
                if (index_entry >= target_entry) break;
                track += step;
            }
            if (track == bound) break;
            if (index_entry == target_entry) { keep |= 1ULL << bit; ++kept; track += step; }
        }
        if (w == 0) new0 = keep; else stage[w - 1] = keep;
    }
    if (STATS) st->targets_merged += (uint32_t)((track - first_track) * step + (track != bound ? 1 : 0));
    if (kept == 0) return false;
    set.word0 = new0;
    for (int w = 1; w < words; ++w) set.ext[w - 1] = stage[w - 1];
    span.n = kept;
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
__device__ __forceinline__ QWindow left_window(const ReadView &r, int offset)
{
    QWindow q;
    q.base = offset > 0 ? offset - 1 : 0;
    read_window16(r, q.base, q.codes, q.acgt);
    return q;
}
// the windows of the two closing checks (read bases 0..7 and len-8..len-1, _mapper.pyx:270-275,
// :335-343) from the copy kept with the context: 8 codes (16 bits) + 8 "is ACGT" bits each
__device__ __forceinline__ QWindow edge_window(uint32_t codes16, uint32_t acgt8, int base)
{
    return QWindow{codes16 << 16, acgt8 << 8, base};
}
__device__ __forceinline__ int sift4_left(uint32_t ref8, const QWindow &q, int offset)
{
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
__device__ __forceinline__ QWindow right_window(const ReadView &r, int offset)
{
    QWindow q;
    q.base = offset;
    read_window16(r, q.base, q.codes, q.acgt);
    return q;
}
__device__ __forceinline__ int sift4_right(uint32_t ref8, const QWindow &q, int offset, int read_len)
{
    int rc = 0;
    int qc = offset;
    int distance = 0;
    for (int guard = 0; guard < 64 && rc < ALIGN_LENGTH && qc < offset + ALIGN_LENGTH; ++guard) {
        if (q.match(ref_code(ref8, rc), qc)) { ++rc; ++qc; continue; }
        if (rc != qc - offset) { rc = max(qc - offset, rc); qc = rc + offset; }
#pragma unroll
        for (int i = 0; i < MAX_OFFSET; ++i) {
            if (qc + i < offset + ALIGN_LENGTH + 1 && qc + i < read_len
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

// _intersect, _mapper.pyx:350-397: mate 1 ascending against mate 2 walked from
// its end with complemented entries; matches are consumed one to one.
template <bool STATS>
__device__ __forceinline__ bool intersect(const DevIndex &ix, TSet &a, Span &s1, const TSet &b2, const Span &s2)
{
    if (s1.n == 0) return true;
    if (s2.n == 0) return false;
    if (LIST_FAST(STATS) && ix.sorted_targets && a.length <= LIST_REGS && b2.length <= LIST_REGS) {
        int32_t e1[LIST_REGS], e2[LIST_REGS];
        bool twice1, twice2;
        load_list(ix, a.start, a.length, a.forward, (uint32_t)a.word0, NO_ENTRY, e1, twice1);
        // mate 2 is compared complemented: ~entry(i).  load_list with the orientation flipped
        // yields exactly that at mirrored positions, and positions do not matter on this side.
        const uint32_t mirrored = __brev((uint32_t)b2.word0) >> (32 - b2.length);
        load_list(ix, b2.start, b2.length, !b2.forward, mirrored, NO_ENTRY_B, e2, twice2);
        const uint32_t keep = keep_common<LIST_REGS>(e1, e2, twice1 | twice2);
        a.word0 = keep;
        if (keep == 0) return false;
        s1.n = __builtin_popcount(keep);
        return true;
    }
    int w2 = b2.words() - 1;
    uint64_t m2 = b2.word(w2);
    // cursor over mate 2, highest list position first
    auto next2 = [&](int32_t &e2) -> bool {
        while (m2 == 0) {
            if (w2 == 0) return false;
            --w2;
            m2 = b2.word(w2);
        }
        const int bit = 63 - __builtin_clzll(m2);
        e2 = ~b2.entry(ix, 64 * w2 + bit);
        return true;
    };
    auto drop2 = [&]() { m2 &= ~(1ULL << (63 - __builtin_clzll(m2))); };
    int32_t e2 = 0;
    bool have2 = next2(e2);
    int kept = 0;
    const int words = a.words();
    for (int w = 0; w < words; ++w) {
        uint64_t m = have2 ? a.word(w) : 0, keep = 0;
        while (m && have2) {
            const int bit = __builtin_ctzll(m);
            const int32_t e1 = a.entry(ix, 64 * w + bit);
            if (e1 == e2) { keep |= 1ULL << bit; ++kept; m &= m - 1; drop2(); have2 = next2(e2); }
            else if (e1 < e2) m &= m - 1;
            else { drop2(); have2 = next2(e2); }
        }
        a.set_word(w, keep);
    }
    if (kept == 0) return false;
    s1.n = kept;
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
// On a 64-wide wave that shape is ruinous (measured: the kernel is VALU-issue
// bound, not memory bound): the 200-instruction hash sits at seven call sites
// and the wave idles while a few lanes roll their first k-mer past a
// sequencing error or take the single retry.  So the state machine is made
// explicit and scheduled for convergence at BLOCK level:
//   * a block keeps NCTX unit contexts (state, span, pending k-mer ...) in LDS
//     and refills finished ones from its private range of units;
//   * every context sits in exactly one of six per-action queues (LDS rings);
//     each of the block's waves, on its own and without block barriers, pops
//     up to 64 contexts from the most backed-up queue, runs that ONE action --
//     the index lookup, the list merge, the left / right 8-base alignment
//     step, the start of a unit or its emission, each one piece of straight
//     code -- for all of them, and pushes every context to the queue of its
//     next action (measured with synchronous rounds: a third of the wave time
//     was spent at the round barrier behind the slowest chunk);
//   * a short queue is left to fill up while other waves are still producing.
// States that wait for a lookup result:
//   Y_FIRST  first-hit scan (_find_first_kmer)            Y_RA  right re-anchor (:283-284)
//   Y_SCAN   the same scan once its first k-mer has missed: the read is being rolled past a
//            sequencing error (13 further misses on average, 76 for a read that maps nowhere).
//            These contexts have a queue of their own whose action does up to SCAN_ROUNDS
//            lookups in a row, so the fixed cost of a scheduling round is paid once per
//            SCAN_ROUNDS misses and the wave stays converged (every lane is rolling)
//   Y_LJ/Y_LS left junction / skip-a-k lookup (:247-263)  Y_RJ  right junction (:309-315)
// Other waiting states: M_* (_filter_on_contig), N_LEFT (:229-246,
// :270-275), N_RIGHT (:285-308, :335-343), ST_NEW, ST_UNIT_DONE (map_read_pair :129-144 +
// batch loop :89-94).  Cheap transitions (:174-193) run right after the action that
// caused them.
enum : int { ST_IDLE = 0, ST_NEW,
             Y_FIRST, Y_LJ, Y_LS, Y_RA, Y_RJ,          // want a lookup
             M_LJ, M_LS, M_RJ, M_FIRST,                // want a list merge (M_FIRST: only the first hit's list)
             N_LEFT, N_RIGHT,                          // want an 8-base alignment step
             N_RIGHT_ENTER, N_AFTER, N_MATE_DONE,      // cheap transitions
             ST_UNIT_DONE,                             // want emission
             Y_SCAN,                                   // want a run of lookups
             ST_HALF };                                // a mate that is done and waits for the other one
enum : int { A_START = 0, A_LOOKUP, A_MERGE, A_LEFT, A_RIGHT, A_EMIT, A_SCAN, N_ACTIONS };
#ifndef SKM_SCAN_ROUNDS
#define SKM_SCAN_ROUNDS 5
#endif
constexpr int SCAN_ROUNDS = SKM_SCAN_ROUNDS;    // k-mers one round of the first-hit roll looks up together
#ifndef SKM_SIGNED_CANDIDATES
#define SKM_SIGNED_CANDIDATES 8
#endif
constexpr int SIGNED_CANDIDATES = SKM_SIGNED_CANDIDATES;            // ... with the signatures in front: a window of k + 6 bases of the read


constexpr int NCTX = MAP_CONTEXTS;    // unit contexts per block (LDS)
constexpr int ARENA_CHUNK = 2048;     // ids a wave takes from the entry arena per atomic

constexpr int FLD_WINDOW = 512;       // fragment lengths below this are counted in LDS

__device__ __forceinline__ int action_of(int state)
{
    if (state >= Y_FIRST && state <= Y_RJ) return A_LOOKUP;
    if (state == Y_SCAN) return A_SCAN;
    if (state == ST_NEW) return A_START;
    if (state >= M_LJ && state <= M_FIRST) return A_MERGE;
    if (state == N_LEFT) return A_LEFT;
    if (state == N_RIGHT) return A_RIGHT;
    return A_EMIT;                                    // ST_UNIT_DONE
}

// the one place the k-mer table is asked: bucket table (BUCKETS) or the reference's layout
template <bool STATS, bool BUCKETS>
__device__ __forceinline__ Coord lookup_kmer(const DevIndex &ix, uint64_t kmer, LaneStats *st)
{
    if (BUCKETS) return map_kmer_buckets_whole(ix, kmer);
    return map_kmer<STATS>(ix, kmer, st);
}

// <false, *>: the production kernels (BUCKETS: the index carries a bucket table, as every built
// index does).  <true, false>: the counting build -- every access of the reference's algorithm is
// performed in the reference's table layout and counted (these counts are the algorithmic bytes),
// plus the scheduler census.  <true, true>: the census build, a profiling aid -- the production
// code paths with the scheduler census and cycle stamps (its access counters undercount).
template <bool STATS, bool BUCKETS>
__global__ void __launch_bounds__(MAP_THREADS, SKM_MAP_WAVES_PER_EU)
map_units_kernel(DevIndex ix, MapBatch b)
{
    constexpr bool COUNT = STATS && !BUCKETS;
    // junction lookups answered by the contig records (DevContig::succ; the counting build performs
    // and counts the reference's lookups).  Uniform over the launch.
    const bool SUCCESSORS = BUCKETS && ix.successors != 0;
    __shared__ uint32_t fld_lds[FLD_WINDOW];
    // contexts, structure of arrays
    __shared__ int32_t c_state[NCTX], c_unit[NCTX], c_begin[NCTX], c_end[NCTX], c_aentry[NCTX],
                       c_aoffset[NCTX], c_scan[NCTX], c_len[NCTX], c_tstart[NCTX], c_tlen[NCTX];
    // c_look: the 16 read bases (2-bit codes) of the aligned half word that holds base c_scan,
    // so that rolling the first k-mer forward touches the read record once per 16 bases
    __shared__ uint32_t c_kmer_lo[NCTX], c_kmer_hi[NCTX], c_mask_lo[NCTX], c_mask_hi[NCTX], c_look[NCTX];
    // c_edge: the first 8 (low half) and the last 8 (high half) bases of the current read as
    // 2-bit codes -- what the two closing checks compare (_mapper.pyx:270-275, :335-343); their
    // "is ACGT" bits ride in bits 20-27 of c_scan (head) and c_len (tail).  The read record
    // is in L2 for ~10 us after a touch and a unit lives 15 rounds: without the copy every
    // closing check fetches the record's sector again.
    __shared__ uint32_t c_edge[NCTX];
    // one MPMC ring per action: entry = context | 0x8000 once written, 0 while empty
    __shared__ uint16_t ring[N_ACTIONS][NCTX];
    __shared__ uint32_t q_head[N_ACTIONS], q_tail[N_ACTIONS], next_unit, done_units, busy, stalled;
    // Paired batches: the two mates of a unit are mapped SIDE BY SIDE by the contexts 2p (mate 1)
    // and 2p + 1 (mate 2) of pair slot p -- map_read_pair maps them independently and only then
    // intersects (_mapper.pyx:119-128).  The mate that finishes first stays in its context
    // (ST_HALF); the one that finishes second takes the unit to the emission, which reads both
    // contexts.  pair_done[p] counts the finished mates of the slot.  (Mapping the mates one after
    // the other in one context meant parking mate 1's span and set in HBM: a sector written and
    // a sector read back per pair.)
    __shared__ uint32_t pair_done[NCTX / 2];

    for (int i = threadIdx.x; i < FLD_WINDOW; i += blockDim.x) fld_lds[i] = 0;
    for (int i = threadIdx.x; i < N_ACTIONS * NCTX; i += blockDim.x) (&ring[0][0])[i] = 0;
    if (threadIdx.x < N_ACTIONS) { q_head[threadIdx.x] = 0; q_tail[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { next_unit = 0; done_units = 0; busy = 0; stalled = 0; }
    for (int i = threadIdx.x; i < NCTX / 2; i += blockDim.x) pair_done[i] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    // this block's private range of units: no global atomics for work distribution
    const int64_t per_block = (b.n_units + gridDim.x - 1) / gridDim.x;
    const int64_t block_first = blockIdx.x * per_block;
    const int64_t block_units = max((int64_t)0, min(b.n_units, block_first + per_block) - block_first);
    // mask extension words (live + staging, per mate) for slices longer than 64 targets
    const int ext_words = max(0, (ix.max_target_count + 63) / 64 - 1);
    uint64_t *const ws_block = reinterpret_cast<uint64_t *>(b.workspace)
                               + (size_t)blockIdx.x * NCTX * 4 * (size_t)ext_words;
    // the block's records, addressed with 32-bit byte offsets (the host checks the range fits)
    const uint32_t record_bytes = (uint32_t)b.record_words << 2;
    const char *const block_records = reinterpret_cast<const char *>(b.records)
                                      + (size_t)(b.paired ? 2 * block_first : block_first) * record_bytes;
    int64_t chunk_pos = 0, chunk_end = 0;      // wave-uniform slice of the entry arena
    LaneStats ls = {0, 0, 0, 0, 0, 0, 0};
    uint64_t read_bases = 0, n_reads = 0, tuple_ids = 0;
    uint32_t census[1 + 2 * N_ACTIONS];
    for (int i = 0; i < 1 + 2 * N_ACTIONS; ++i) census[i] = 0;
    // cycle stamps (STATS build): [0]=idle / choosing [1]=unused [2..]=per action
    unsigned long long cyc[2 + N_ACTIONS + 6];      // the last six: phases of the emission
    for (int i = 0; i < 2 + N_ACTIONS + 6; ++i) cyc[i] = 0;
    unsigned long long t_mark = STATS ? clock64() : 0;

    // seed: every context (pair of contexts) takes a unit and queues up for A_START
    const int slot_width = b.paired ? 2 : 1;
    for (int c = slot_width * (int)threadIdx.x; c + slot_width <= NCTX; c += slot_width * (int)blockDim.x) {
        const uint32_t k = atomicAdd(&next_unit, 1u);
        if ((int64_t)k < block_units) {
            for (int m = 0; m < slot_width; ++m) {
                c_unit[c + m] = (int32_t)k;
                c_state[c + m] = ST_NEW;
                ring[A_START][atomicAdd(&q_tail[A_START], 1u) % NCTX] = (uint16_t)((c + m) | 0x8000);
            }
        }
    }
    __syncthreads();

    uint32_t idle_spins = 0;
    for (;;) {
        // ------------------------------------------------ pick the most backed-up action
        // (lane a reads queue a: one LDS round trip for all six, then scalar compares)
        uint32_t my_head = 0, my_tail = 0;
        if (lane < N_ACTIONS) {
            my_head = *(volatile uint32_t *)&q_head[lane];
            my_tail = *(volatile uint32_t *)&q_tail[lane];
        }
        int action = -1;
        uint32_t avail = 0, head = 0;
#pragma unroll
        for (int a = 0; a < N_ACTIONS; ++a) {
            const uint32_t h = __builtin_amdgcn_readlane(my_head, a);
            const uint32_t t = __builtin_amdgcn_readlane(my_tail, a);
            if (t - h > avail && (int)(t - h) >= 0) { avail = t - h; action = a; head = h; }
        }
        if (*(volatile uint32_t *)&stalled) break;        // a bounded spin gave up: drain, host reports it
        if (avail == 0) {
            if (*(volatile uint32_t *)&done_units >= (uint32_t)block_units) break;
            if (++idle_spins > (1u << 26)) { stalled = 1; break; }
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        // a short queue is worth waiting for while other waves are still producing
        if ((int)avail < b.vote[action] && *(volatile uint32_t *)&busy != 0) {
            if (++idle_spins > (1u << 26)) { stalled = 1; break; }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        idle_spins = 0;
        const uint32_t take = avail < 64u ? avail : 64u;
        uint32_t won = 0;
        if (lane == 0) {
            won = atomicCAS(&q_head[action], head, head + take) == head;
            if (won) atomicAdd(&busy, 1u);
        }
        if (!__shfl(won, 0, 64)) continue;
        if (STATS) { const unsigned long long t = clock64(); cyc[0] += t - t_mark; t_mark = t; }
        const bool valid = (uint32_t)lane < take;
        if (STATS && lane == 0) { census[0]++; census[1 + 2 * action]++; census[2 + 2 * action] += take; }
        int c = 0;
        if (valid) {
            volatile uint16_t *slot = &ring[action][(head + lane) % NCTX];
            uint16_t e;
            uint32_t spins = 0;
            do { e = *slot; } while (!(e & 0x8000) && ++spins < (1u << 24));   // reserved by a producer, written in a moment
            if (!(e & 0x8000)) stalled = 1;
            *slot = 0;
            c = e & 0x7fff;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        int next_action = -1;                             // queue the context goes to afterwards
        int partner = -1;                                 // (emission of a pair: the other context starts over too)
        {
            const unsigned long long t_action = STATS ? clock64() : 0;
            // load the context
            int word = valid ? c_state[c] : ST_IDLE;
            int state = word & 0xff, attempt = (word >> 9) & 1;
            // the k-mer words hold the first hit's position while the left filter runs (see N_RIGHT_ENTER)
            int first_hit_kept = (word >> 8) & 1;
            const int mate = b.paired ? (c & 1) : 0;
            const int64_t u = block_first + c_unit[c];
            Span span{c_begin[c], c_end[c], Coord{c_aentry[c], c_aoffset[c]}, (int32_t)((uint32_t)word >> 10)};
            uint32_t look = c_look[c];
            uint32_t edge = c_edge[c];
            const uint32_t scan_word = (uint32_t)c_scan[c], len_word = (uint32_t)c_len[c];
            int scan_i = (int)(scan_word & 0xffffffu);
            uint32_t head_acgt = scan_word >> 24, tail_acgt = len_word >> 24;
            uint64_t kmer = ((uint64_t)c_kmer_hi[c] << 32) | c_kmer_lo[c];
            uint64_t *const ext1 = ws_block + (size_t)(b.paired ? (c & ~1) : c) * 4 * (size_t)ext_words;
            uint64_t *const ext2 = ext1 + 2 * (size_t)ext_words;       // (mate 2 of the pair slot)
            TSet set{c_tstart[c], c_tlen[c] >> 1, (c_tlen[c] & 1) != 0,
                     ((uint64_t)c_mask_hi[c] << 32) | c_mask_lo[c], mate ? ext2 : ext1, ext_words};
            const uint32_t first_read = b.paired ? 2u * (uint32_t)c_unit[c] : (uint32_t)c_unit[c];
            ReadView rv{block_records, (first_read + (uint32_t)mate) * record_bytes, b.words_per_read,
                        (int)(len_word & 0xffffffu)};   // (the length is kept with the context: saves touching the record)
            // first and last 8 bases of a read that is about to be mapped -> edge, head_acgt, tail_acgt
            auto keep_edges = [&]() {
                uint32_t codes, acgt;
                read_window16(rv, 0, codes, acgt);
                edge = codes >> 16;
                head_acgt = acgt >> 8;
                read_window16(rv, rv.len - ALIGN_LENGTH, codes, acgt);
                edge |= codes & 0xffff0000u;
                tail_acgt = acgt >> 8;
            };
            bool anchored = false;      // span.anchor is map_kmer(k-mer at span.end) already
            // _find_first_kmer has found its k-mer at read position scan_i - K (:203-206, :211-214).
            // KMerIndex.map_contig's list is NOT fetched here: the record of the contig would be a
            // second dependent access behind the bucket's, and the alignment step that follows reads
            // the same sector anyway -- it takes the list along (set.length < 0 = "list to come"; the
            // counting build fetches it here, as the reference does).
            auto first_hit = [&](Coord pos) {
                span.begin = scan_i - K;
                span.end = span.begin;
                if (COUNT) {
                    map_contig<STATS>(ix, pos, span.begin > 0 ? visit<false>(ix, pos, false, true)
                                                              : visit<true>(ix, pos, false, true), set, span, &ls);
                } else {
                    set.start = 0; set.length = -1; set.forward = pos.entry >= 0; set.word0 = 0;
                    span.n = 1;
                }
                state = span.n == 0 ? N_MATE_DONE : (span.begin > 0 ? N_LEFT : N_RIGHT_ENTER);
                anchored = true;
                if (SUCCESSORS && state == N_LEFT) {
                    kmer = ((uint64_t)(uint32_t)pos.offset << 32) | (uint32_t)pos.entry;
                    first_hit_kept = 1;
                }
            };

            // The merge of a hop (_filter_on_contig on the landing contig, _common.pyx:185-235) settled by
            // the record of the contig the hop LEAVES (span.anchor is still on it): nothing to do when
            // the landing list holds this contig's whole list (SUCC_WHOLE: the running list is a part
            // of it); a stored mask when the running list is still made of positions of THIS contig's
            // list (SUCC_MASKED: no hop since the first hit in this direction).  false: the merge
            // action does it -- also when the mask leaves nothing, so that the failing merge ends
            // with the anchor the reference ends with.
            auto merge_by_record = [&](const Hop &hop, bool forward) -> bool {
                if (hop.whole) return true;
                const int32_t contig = forward ? span.anchor.entry : ~span.anchor.entry;
                if (!hop.masked || (set.start >> 5) != contig || set.forward != forward || set.length > CONTIG_INLINE_TARGETS)
                    return false;
                const uint32_t in_read_order = set.forward ? hop.kept : __brev(hop.kept) >> (32 - set.length);
                const uint32_t keep = (uint32_t)set.word0 & in_read_order;
                if (keep == 0) return false;
                set.word0 = keep;
                span.n = __builtin_popcount(keep);
                return true;
            };

            if (valid && action == A_START) {
                rv = read_view(block_records, b.record_words, b.words_per_read, first_read + (uint32_t)mate);
                attempt = 0;
                first_hit_kept = 0;
                set.ext = mate ? ext2 : ext1;
                set.start = 0; set.length = 0; set.word0 = 0;   // (a context starts with whatever LDS held)
                span.begin = 0; span.end = 0; span.n = 0; span.anchor = invalid_coord();
                if (STATS) { read_bases += rv.len; n_reads++; }
                if (rv.len < K) {
                    state = N_MATE_DONE;              // shorter than k: unmapped (documented deviation)
                } else {
                    kmer = read_kmer(rv, 0);
                    scan_i = K;
                    look = read_half(rv, scan_i >> 4);
                    keep_edges();
                    state = Y_FIRST;
                }
            } else if (valid && action == A_LOOKUP) {
                // ---------------------------------- the one index lookup site
                const Coord pos = lookup_kmer<STATS, BUCKETS>(ix, kmer, &ls);
                span.anchor = pos;
                if (state == Y_FIRST) {                       // _find_first_kmer, :199-216
                    if (pos.offset >= 0) {
                        first_hit(pos);
                    } else if (scan_i < rv.len) {
                        kmer = ((kmer << 2) | ((look >> (30 - 2 * (scan_i & 15))) & 3u)) & KMER_MASK;   // _kmer.append
                        ++scan_i;
                        if ((scan_i & 15) == 0 && scan_i < rv.len) look = read_half(rv, scan_i >> 4);
                        state = Y_SCAN;                       // the rest of the roll: A_SCAN
                    } else {
                        state = N_MATE_DONE;                  // no hit: returned as is, no retry
                    }
                } else if (state == Y_RA) {
                    state = N_RIGHT;
                } else if (pos.offset >= 0) {
                    state = state == Y_LJ ? M_LJ : (state == Y_LS ? M_LS : M_RJ);
                } else if (state == Y_LJ) {                   // miss at the junction, :250-259
                    if (span.begin < K) { span.begin = 0; state = N_RIGHT_ENTER; }
                    else { span.begin -= K; kmer = read_kmer(rv, span.begin); state = Y_LS; first_hit_kept = 0; }
                } else {                                      // Y_LS :260-263, Y_RJ :312-315
                    span.n = 0;
                    state = N_AFTER;
                }
            } else if (valid && action == A_SCAN && BUCKETS && ix.signatures != nullptr) {
                // ------------- _find_first_kmer's roll, :207-216, with the signatures in front of the bucket
                // table (skm_device.h: kmer_min_hash): up to SIGNED_CANDIDATES k-mers of the read -- the one
                // at hand and those its next bases make, a window of k + SIGNED_CANDIDATES - 1 bases --
                // ask the signature of their minimizer (a run of k-mers shares it: one or two sectors for
                // the lot) whether they can be in the table at all; those that can are looked up in read
                // order, so the first hit is the one the reference's one-by-one roll stops at.
                constexpr int N = SIGNED_CANDIDATES;
                uint32_t more = 0;
                int m = 1;
#pragma unroll
                for (int j = 1; j < N; ++j) {
                    const int at = scan_i + j - 1;                 // the base that makes candidate j
                    const bool ok = m == j && at < rv.len && (at >> 4) == (scan_i >> 4);
                    more = (more << 2) | (ok ? (look >> (30 - 2 * (at & 15))) & 3u : 0u);
                    m += ok ? 1 : 0;
                }
                const uint64_t window = (kmer << (2 * (N - 1))) | more;   // candidate j = bases j .. j + k - 1 of it
                const uint64_t mirror = revcomp32(window) >> (2 * (32 - (K + N - 1)));   // its reverse complement = bases N-1-j .. of this
                uint32_t forward[K - MINIMIZER_BASES + N], backward[K - MINIMIZER_BASES + N];
#pragma unroll
                for (int s = 0; s < K - MINIMIZER_BASES + N; ++s) {
                    forward[s] = mmer_hash((uint32_t)(window >> (2 * (K - MINIMIZER_BASES + N - 1 - s))) & ((1u << (2 * MINIMIZER_BASES)) - 1u));
                    backward[s] = mmer_hash((uint32_t)(mirror >> (2 * (K - MINIMIZER_BASES + N - 1 - s))) & ((1u << (2 * MINIMIZER_BASES)) - 1u));
                }
                // (a run of candidates shares its minimizer: the signature is asked for once per run)
                u64x2 signature[N];
                uint32_t slot = 0xffffffffu;
                u64x2 word{0, 0};
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    uint32_t least = 0xffffffffu;
#pragma unroll
                    for (int i = 0; i < MINIMIZERS_PER_KMER; ++i)
                        least = min(least, min(forward[j + i], backward[N - 1 - j + i]));
                    const uint32_t mine = signature_slot(least, ix.signature_shift);
                    if (j < m && mine != slot) {
                        slot = mine;
                        word = *reinterpret_cast<const u64x2 *>(ix.signatures + 2 * (size_t)slot);
                    }
                    signature[j] = word;
                }
                uint32_t possible = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const uint64_t candidate = (window >> (2 * (N - 1 - j))) & KMER_MASK;
                    const uint64_t rc = (mirror >> (2 * j)) & KMER_MASK;
                    const Signature wanted = signature_bits(bucket_hash(candidate < rc ? candidate : rc));
                    if (j < m && (signature[j].x & wanted.lo) == wanted.lo && (signature[j].y & wanted.hi) == wanted.hi)
                        possible |= 1u << j;
                }
                if (STATS) { ls.lookups += (uint32_t)m; ls.slots += (uint32_t)__builtin_popcount(possible); }   // (census: asked / passed)
                // In read order, the candidates that can be in the table are looked up one after the other:
                // the first of them is nearly always the hit the roll is after (the k-mers before it hold
                // the sequencing error and were turned away by their signatures), and the k-mers behind a
                // hit are in the table too -- asking for their buckets up front fetched three sectors per
                // round for nothing.  A candidate that was turned away is a miss (invalid_coord, as
                // map_kmer returns it); span.anchor is the result of the LAST k-mer the roll looked at.
                int last = m - 1;
                Coord pos = invalid_coord();
                for (uint32_t todo = possible; todo != 0; todo &= todo - 1) {
                    const int j = __builtin_ctz(todo);
                    const Coord got = map_kmer_buckets_whole(ix, (window >> (2 * (N - 1 - j))) & KMER_MASK);
                    if (got.offset >= 0) { pos = got; last = j; break; }
                    if (j == m - 1) pos = got;         // (stored without a position, offset < 0: a miss that leaves its trace)
                }
                kmer = (window >> (2 * (N - 1 - last))) & KMER_MASK;      // the k-mer of candidate `last`
                scan_i += last;
                span.anchor = pos;
                if (pos.offset >= 0) {
                    first_hit(pos);
                } else if (scan_i < rv.len) {
                    if ((scan_i >> 4) != ((scan_i - last) >> 4)) look = read_half(rv, scan_i >> 4);
                    kmer = ((kmer << 2) | ((look >> (30 - 2 * (scan_i & 15))) & 3u)) & KMER_MASK;
                    ++scan_i;
                    if ((scan_i & 15) == 0 && scan_i < rv.len) look = read_half(rv, scan_i >> 4);
                } else {
                    state = N_MATE_DONE;
                }
            } else if (valid && action == A_SCAN && BUCKETS) {
                // ------------- _find_first_kmer's roll, :207-216, over the bucket table: the next
                // SCAN_ROUNDS k-mers of the read are looked up TOGETHER (their buckets are
                // independent loads: one memory latency for the lot) and then judged in read
                // order, so the first hit is the one the reference's one-by-one roll stops at.
                // A k-mer joins the group while its last base is in the 16-base look-ahead.
                // Every candidate asks for the 16 bytes of `low` words of its bucket (DevBucket) and keeps
                // the 31 bits it is looking for: five registers per candidate.  All of them but the last
                // are expected to miss -- the read is being rolled past a sequencing error -- and a miss
                // is settled by those words alone; the entry that matches them is fetched whole.
                uint32_t want[SCAN_ROUNDS];
                u32x4 low[SCAN_ROUNDS];
                int m = 0;
                {
                    uint64_t rolling = kmer;
#pragma unroll
                    for (int j = 0; j < SCAN_ROUNDS; ++j) {
                        const int at = scan_i + j - 1;             // the base that makes candidate j
                        const bool ok = j == 0 || (m == j && at < rv.len && (at >> 4) == (scan_i >> 4));
                        if (ok && j > 0)
                            rolling = ((rolling << 2) | ((look >> (30 - 2 * (at & 15))) & 3u)) & KMER_MASK;
                        m += ok ? 1 : 0;
                        const uint64_t rc = kmer_revcomp(rolling);
                        const uint64_t canonical = rolling < rc ? rolling : rc;
                        want[j] = (uint32_t)canonical;
                        low[j] = bucket_low(ix, bucket_hash(canonical) >> ix.bucket_shift);   // (j >= m repeats the last one)
                    }
                }
                // judged in read order: the roll stops at the first candidate that is not a plain miss
                // (a likely hit, a full bucket, two entries alike) and looks that one up in full -- its
                // sector is in the cache.  Should that be a miss after all (or a k-mer stored without a
                // position, offset < 0: a miss too, :203, :211), the judging goes on behind it.
                int last = 0;                                      // candidate the roll stopped at
                Coord pos = invalid_coord();                       // (span.anchor is the LAST lookup's result)
                const uint64_t first = kmer;
                for (int from = 0; from < m;) {
                    int stop = -1;
#pragma unroll
                    for (int j = 0; j < SCAN_ROUNDS; ++j) {
                        if (j >= from && j < m && stop < 0) {
                            last = j;
                            if (bucket_screen(low[j], want[j]) != -1) stop = j;
                        }
                    }
                    kmer = first;
                    for (int r = 0; r < last; ++r)                 // the k-mer of candidate `last`
                        kmer = ((kmer << 2) | ((look >> (30 - 2 * ((scan_i + r) & 15))) & 3u)) & KMER_MASK;
                    pos = invalid_coord();
                    if (stop < 0) break;
                    pos = map_kmer_buckets(ix, kmer);
                    if (pos.offset >= 0) break;
                    from = stop + 1;
                }
                scan_i += last;
                span.anchor = pos;
                if (pos.offset >= 0) {
                    first_hit(pos);
                } else if (scan_i < rv.len) {
                    if ((scan_i >> 4) != ((scan_i - last) >> 4)) look = read_half(rv, scan_i >> 4);
                    kmer = ((kmer << 2) | ((look >> (30 - 2 * (scan_i & 15))) & 3u)) & KMER_MASK;
                    ++scan_i;
                    if ((scan_i & 15) == 0 && scan_i < rv.len) look = read_half(rv, scan_i >> 4);
                } else {
                    state = N_MATE_DONE;
                }
            } else if (valid && action == A_SCAN) {
                // ------------- the same roll over the reference's table layout, SCAN_ROUNDS k-mers per round
                for (int round = 0; round < SCAN_ROUNDS && state == Y_SCAN; ++round) {
                    const Coord pos = lookup_kmer<STATS, BUCKETS>(ix, kmer, &ls);
                    span.anchor = pos;
                    if (pos.offset >= 0) {
                        first_hit(pos);
                    } else if (scan_i < rv.len) {
                        kmer = ((kmer << 2) | ((look >> (30 - 2 * (scan_i & 15))) & 3u)) & KMER_MASK;
                        ++scan_i;
                        if ((scan_i & 15) == 0 && scan_i < rv.len) look = read_half(rv, scan_i >> 4);
                    } else {
                        state = N_MATE_DONE;
                    }
                }
            } else if (valid && action == A_MERGE) {
                // ---------------------------------- the one _filter_on_contig site
                if (state == M_FIRST) {                       // a first hit that no step follows (see N_RIGHT_ENTER)
                    map_contig<STATS>(ix, span.anchor, visit<true>(ix, span.anchor, false, true), set, span, &ls);
                    state = span.n == 0 ? N_MATE_DONE : N_AFTER;
                } else {
                    const bool ok = filter_on_contig<COUNT>(ix, state == M_RJ, set, span, &ls);
                    if (state == M_LJ) {
                        if (ok) state = N_LEFT;
                        else if (span.begin < K) { span.begin = 0; state = N_RIGHT_ENTER; }
                        else { span.begin -= K; kmer = read_kmer(rv, span.begin); state = Y_LS; first_hit_kept = 0; }
                    } else if (state == M_LS) {
                        if (ok) state = N_LEFT; else { span.n = 0; state = N_AFTER; }
                    } else {
                        if (ok) state = N_RIGHT; else { span.n = 0; state = N_AFTER; }
                    }
                }
            } else if (valid && action == A_LEFT) {
                // --------------- _filter_targets_to_left: loop head + alignment step
                const bool forward = span.anchor.entry >= 0;
                // Everything the step reads of its contig is in ONE sector of the record (DevSide) and
                // is asked for at once: place and length, the 8 bases at this end, the four junction
                // successors (a forward anchor's distance to the contig's left edge is its offset: whether
                // this is a hop is known before the sector arrives, and a closing check asks for none) and,
                // straight after a first hit, the contig's list (see first_hit).
                // (an anchor that a merge-free hop has just put on its contig's last k-mer learns its offset here)
                const bool at_end = span.anchor.offset == OFFSET_AT_END;
                const SideVisit at_side = visit<false>(ix, span.anchor,
                                                       SUCCESSORS && (at_end || !forward || span.begin > span.anchor.offset),
                                                       set.length < 0);
                if (at_end) span.anchor.offset = at_side.length - K;
                if (set.length < 0) map_contig<STATS>(ix, span.anchor, at_side, set, span, &ls);
                if (span.n == 0) {                            // (a first hit without targets: as map_contig left it)
                    state = N_MATE_DONE;
                } else {
                    const int move = forward ? span.anchor.offset : at_side.length - span.anchor.offset - K;
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
                    // (the closing check compares read bases 0..7: kept with the context)
                    const QWindow q = in_loop ? left_window(rv, at) : edge_window(edge & 0xffffu, head_acgt, 0);
                    const int shift = sift4_left(in_loop ? contig8_edge<STATS>(ix, span.anchor, at_side, true, &ls)
                                                         : contig8<STATS>(ix, span.anchor, at_side.offset, true, &ls), q, at);
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
                        } else if (SUCCESSORS) {
                            // the junction lookup (:247-249), answered by the record of the contig the hop
                            // leaves: what A_LOOKUP does for Y_LJ, without the visit to the k-mer table
                            const Hop hop = junction_successor(at_side, forward, read_code(rv, span.begin));
                            const uint32_t kind = hop.kind;
                            if (kind == SUCC_LOOKUP) {            // (the k-mer words are needed: the first hit goes)
                                kmer = (tail_kmer<STATS>(ix, span.anchor, &ls) >> 2)
                                       | ((uint64_t)read_code(rv, span.begin) << (2 * K - 2));
                                state = Y_LJ;
                                first_hit_kept = 0;
                            } else if (kind != SUCC_ABSENT) {
                                state = merge_by_record(hop, forward) ? N_LEFT : M_LJ;
                                span.anchor = hop.landing;
                            } else {
                                span.anchor = invalid_coord();
                                if (span.begin < K) { span.begin = 0; state = N_RIGHT_ENTER; }      // :250-259
                                else { span.begin -= K; kmer = read_kmer(rv, span.begin); state = Y_LS; first_hit_kept = 0; }
                            }
                        } else {
                            kmer = (tail_kmer<STATS>(ix, span.anchor, &ls) >> 2)          // _kmer.prepend
                                   | ((uint64_t)read_code(rv, span.begin) << (2 * K - 2));
                            state = Y_LJ;
                        }
                    }
                }
            } else if (valid && action == A_RIGHT) {
                // -------------- _filter_targets_to_right: loop head + alignment step
                const bool forward = span.anchor.entry >= 0;
                const int rest = rv.len - span.end - K;
                const bool at_end = span.anchor.offset == OFFSET_AT_END;
                const SideVisit at_side = visit<true>(ix, span.anchor,
                                                      SUCCESSORS && (at_end || forward || rest > span.anchor.offset),
                                                      set.length < 0);
                if (at_end) span.anchor.offset = at_side.length - K;
                if (set.length < 0) map_contig<STATS>(ix, span.anchor, at_side, set, span, &ls);
                if (span.n == 0) {                            // (a first hit without targets: as map_contig left it)
                    state = N_MATE_DONE;
                } else {
                    const int move = forward ? at_side.length - span.anchor.offset - K : span.anchor.offset;
                    if (STATS) ls.contig_reads++;
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
                    // (the closing check compares the last 8 read bases: kept with the context)
                    const QWindow q = in_loop ? right_window(rv, at) : edge_window(edge >> 16, tail_acgt, at);
                    const int shift = sift4_right(in_loop ? contig8_edge<STATS>(ix, span.anchor, at_side, false, &ls)
                                                          : contig8<STATS>(ix, span.anchor, at_side.offset, false, &ls),
                                                  q, at, rv.len);
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
                        } else if (SUCCESSORS) {
                            // the junction lookup (:309-311) from the record: Y_RJ's part of A_LOOKUP
                            const Hop hop = junction_successor(at_side, forward, read_code(rv, span.end + K - 1));
                            const uint32_t kind = hop.kind;
                            if (kind == SUCC_LOOKUP) {
                                kmer = ((tail_kmer<STATS>(ix, span.anchor, &ls) << 2)
                                        | read_code(rv, span.end + K - 1)) & KMER_MASK;
                                state = Y_RJ;
                            } else if (kind != SUCC_ABSENT) {
                                state = merge_by_record(hop, forward) ? N_RIGHT : M_RJ;
                                span.anchor = hop.landing;
                            } else {                                                                  // :312-315
                                span.anchor = invalid_coord();
                                span.n = 0;
                                state = N_AFTER;
                            }
                        } else {
                            kmer = ((tail_kmer<STATS>(ix, span.anchor, &ls) << 2)          // _kmer.append
                                    | read_code(rv, span.end + K - 1)) & KMER_MASK;
                            state = Y_RJ;
                        }
                    }
                }
            } else if (action == A_EMIT) {
                // ---------------------------------------------- A_EMIT: finished units
                unsigned long long t_e = STATS ? clock64() : 0;
                auto phase = [&](int k) {
                    if (STATS) { const unsigned long long t = clock64(); cyc[2 + N_ACTIONS + k] += t - t_e; t_e = t; }
                };
                int n_out = 0;
                // (counted as done before its records are stored: a wave that sees the block
                // finished leaves the loop, and the kernel's end orders the stores)
                const unsigned long long finished = __ballot(valid);
                uint32_t rec_base = 0;
                if (lane == 0) rec_base = atomicAdd(&done_units, (uint32_t)__popcll(finished));
                rec_base = __shfl(rec_base, 0, 64);
                if (valid) {
                    if (b.paired) {
                        // map_read_pair, _mapper.pyx:129-144: both mates' contexts of the pair slot
                        const int c1 = c & ~1, c2 = c | 1;
                        // (only the lists and their sizes before the intersection: what the fragment
                        // length needs is read from the contexts behind it -- registers are short here)
                        Span s1{0, 0, Coord{0, 0}, (int32_t)((uint32_t)c_state[c1] >> 10)};
                        TSet set1{c_tstart[c1], c_tlen[c1] >> 1, (c_tlen[c1] & 1) != 0,
                                  ((uint64_t)c_mask_hi[c1] << 32) | c_mask_lo[c1], ext1, ext_words};
                        span = Span{0, 0, Coord{0, 0}, (int32_t)((uint32_t)c_state[c2] >> 10)};
                        set = TSet{c_tstart[c2], c_tlen[c2] >> 1, (c_tlen[c2] & 1) != 0,
                                   ((uint64_t)c_mask_hi[c2] << 32) | c_mask_lo[c2], ext2, ext_words};
                        phase(0);
                        const bool common = intersect<COUNT>(ix, set1, s1, set, span);
                        s1.begin = c_begin[c1];
                        s1.end = c_end[c1];
                        s1.anchor = Coord{c_aentry[c1], c_aoffset[c1]};
                        const Coord anchor2{c_aentry[c2], c_aoffset[c2]};
                        if (!common) {
                            s1.n = 0;
                            s1.begin = 0;
                            s1.end = -K;
                        } else if (s1.anchor.entry != ~anchor2.entry) {
                            s1.begin = 0;
                            s1.end = -K;
                        } else {
                            const int len1 = (int)((uint32_t)c_len[c1] & 0xffffffu), len2 = (int)((uint32_t)c_len[c2] & 0xffffffu);
                            int interval = anchor2.offset - s1.anchor.offset;
                            if (s1.anchor.entry < 0) interval = -interval;
                            s1.end = (len1 - K) + interval + (len2 - K) - c_begin[c2];
                        }
                        span = s1;
                        set = set1;
                    }
                    phase(1);
                    // fragment length rule, _mapper.pyx:90-94
                    int length = span.end - span.begin + K;
                    if (length > 0) {
                        if (length >= MAX_FRAGMENT_LENGTH) length = MAX_FRAGMENT_LENGTH - 1;
                        if (length < FLD_WINDOW) atomicAdd(&fld_lds[length], 1u);
                        else atomicAdd(&b.fld[length], 1ULL);
                    }
                    n_out = span.n;
                }
                phase(2);
                // one slice of the entry arena for the whole chunk
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
                phase(3);
                if (valid) {
                    const int64_t off = chunk_pos + scan - n_out;
                    uint64_t key = 0x243F6A8885A308D3ULL ^ (uint64_t)n_out;
                    const bool fits = off + n_out <= b.ids_capacity;
                    int i = 0;
                    int words = n_out ? set.words() : 0;
                    if (LIST_FAST(COUNT) && n_out && set.length <= LIST_REGS) {      // short list: one round trip
                        int32_t e[LIST_REGS];
                        load_list(ix, set.start, set.length, set.forward, (uint32_t)set.word0, e);
#pragma unroll
                        for (int k = 0; k < LIST_REGS; ++k) {
                            if (e[k] != NO_ENTRY) {
                                if (fits) b.unit_entries[off + i] = e[k];
                                key = tuple_key_step(key, (uint32_t)(e[k] < 0 ? ~e[k] : e[k]));
                                ++i;
                            }
                        }
                        words = 0;
                    }
                    for (int w = 0; w < words; ++w) {
                        uint64_t m = set.word(w);
                        while (m) {
                            const int32_t e = set.entry(ix, 64 * w + __builtin_ctzll(m));
                            m &= m - 1;
                            if (fits) b.unit_entries[off + i] = e;
                            key = tuple_key_step(key, (uint32_t)(e < 0 ? ~e : e));   // _get_ids, :533-536
                            ++i;
                        }
                    }
                    if (key == 0) key = 1;
                    phase(4);
                    // the wave's finished units take consecutive records of the block's range
                    const int64_t r = block_first + rec_base + __popcll(finished & ((1ULL << lane) - 1));
                    b.rec_unit[r] = (int32_t)u;
                    b.rec_key[r] = n_out ? key : 0;
                    b.rec_tuple[r] = (unsigned long long)off | ((unsigned long long)n_out << 40);
                    if (b.keep_spans) {
                        b.unit_begin[u] = span.begin;
                        b.unit_end[u] = span.end;
                        b.unit_anchor[u] = span.anchor;
                    }
                    if (STATS) tuple_ids += n_out;
                }
                chunk_pos += wave_total;
                phase(5);
                // each of these contexts takes the block's next unit, or retires
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&next_unit, (uint32_t)__popcll(finished));
                base = __shfl(base, 0, 64);
                if (valid) {
                    const uint32_t k = base + __popcll(finished & ((1ULL << lane) - 1));
                    const bool more = (int64_t)k < block_units;
                    if (b.paired) {                       // both contexts of the pair slot start over
                        const int c1 = c & ~1, c2 = c | 1;
                        c_unit[c1] = (int32_t)k; c_unit[c2] = (int32_t)k;
                        c_state[c1] = more ? ST_NEW : ST_IDLE;
                        c_state[c2] = more ? ST_NEW : ST_IDLE;
                        pair_done[c >> 1] = 0;
                        if (more) { next_action = A_START; partner = c ^ 1; }
                    } else if (more) {
                        c_unit[c] = (int32_t)k;
                        c_state[c] = ST_NEW;
                        next_action = A_START;
                    } else {
                        c_state[c] = ST_IDLE;
                    }
                }
            }

            // ------------------------------ cheap transitions until the context waits again
            while (valid && action != A_EMIT && state >= N_RIGHT_ENTER && state <= N_MATE_DONE) {
                if (state == N_RIGHT_ENTER) {                 // map_read, :174-176 / :190-192
                    if (span.n != 0 && span.end < rv.len - K) {
                        // :283-284 re-anchors on the k-mer at span.end.  Straight after the
                        // first hit that is the k-mer just looked up and the anchor is its
                        // result, so the repeat is skipped (the counting build performs it).
                        if (anchored && !COUNT) {
                            state = N_RIGHT;
                        } else if (first_hit_kept && !COUNT) {
                            // after a left filter the k-mer at span.end is still the first hit (the
                            // filter moves span.begin only), whose position was kept in the k-mer
                            // words: they were free, the junction lookups of a record-answered hop
                            // need no k-mer (a skip-a-k lookup, :257-263, takes them back)
                            span.anchor = Coord{(int32_t)(uint32_t)kmer, (int32_t)(uint32_t)(kmer >> 32)};
                            first_hit_kept = 0;
                            state = N_RIGHT;
                        } else {
                            kmer = read_kmer(rv, span.end);
                            state = Y_RA;
                        }
                    } else {
                        state = set.length < 0 ? M_FIRST : N_AFTER;   // (a first hit no step follows: its list)
                    }
                } else if (state == N_AFTER) {
                    if (span.n != 0 || attempt == 1) {
                        state = N_MATE_DONE;
                    } else {                                  // the single retry, :179-185
                        attempt = 1;
                        first_hit_kept = 0;
                        span.anchor = invalid_coord();
                        span.begin += K;
                        if (span.begin + K > rv.len) span.begin = rv.len - K;
                        span.end = span.begin;
                        kmer = read_kmer(rv, span.begin);
                        scan_i = span.begin + K;
                        if (scan_i < rv.len) look = read_half(rv, scan_i >> 4);
                        state = Y_FIRST;
                    }
                } else {                                      // N_MATE_DONE
                    state = b.paired ? ST_HALF : ST_UNIT_DONE;
                }
            }

            // store the context
            if (valid && action != A_EMIT) {
                c_state[c] = (int32_t)((uint32_t)state | ((uint32_t)first_hit_kept << 8) | ((uint32_t)attempt << 9)
                                       | ((uint32_t)span.n << 10));
                c_begin[c] = span.begin;
                c_end[c] = span.end;
                c_aentry[c] = span.anchor.entry;
                c_aoffset[c] = span.anchor.offset;
                c_look[c] = look;
                c_edge[c] = edge;
                c_scan[c] = (int32_t)((uint32_t)scan_i | (head_acgt << 24));
                c_len[c] = (int32_t)((uint32_t)rv.len | (tail_acgt << 24));
                c_kmer_lo[c] = (uint32_t)kmer;
                c_kmer_hi[c] = (uint32_t)(kmer >> 32);
                c_tstart[c] = set.start;
                c_tlen[c] = (set.length << 1) | (set.forward ? 1 : 0);
                c_mask_lo[c] = (uint32_t)set.word0;
                c_mask_hi[c] = (uint32_t)(set.word0 >> 32);
                if (state == ST_HALF) {
                    // this mate is done and its context complete: whoever finds the other mate done
                    // already takes the unit to the emission, the first one just stays put
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    const uint32_t before = atomicAdd(&pair_done[c >> 1], 1u);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    next_action = before == 1u ? A_EMIT : -1;
                } else {
                    next_action = action_of(state);
                }
            }
            if (STATS) cyc[2 + action] += clock64() - t_action;
        }
        // ------------------------------------------------ hand every context to its next queue
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        for (int a = 0; a < N_ACTIONS; ++a) {
            const unsigned long long going = __ballot(next_action == a);
            if (going == 0) continue;
            const int leader = __builtin_ctzll(going);
            uint32_t pos = 0;
            if (lane == leader) pos = atomicAdd(&q_tail[a], (uint32_t)__popcll(going));
            pos = __shfl(pos, leader, 64);
            if (next_action == a) {
                volatile uint16_t *slot = &ring[a][(pos + __popcll(going & ((1ULL << lane) - 1))) % NCTX];
                uint32_t spins = 0;
                while (*slot != 0 && ++spins < (1u << 24)) { }   // (a reader that reserved it is about to clear it)
                if (*slot != 0) stalled = 1;
                *slot = (uint16_t)(c | 0x8000);
            }
        }
        {   // the other context of a pair slot that starts over
            const unsigned long long going = __ballot(partner >= 0);
            if (going) {
                const int leader = __builtin_ctzll(going);
                uint32_t pos = 0;
                if (lane == leader) pos = atomicAdd(&q_tail[A_START], (uint32_t)__popcll(going));
                pos = __shfl(pos, leader, 64);
                if (partner >= 0) {
                    volatile uint16_t *slot = &ring[A_START][(pos + __popcll(going & ((1ULL << lane) - 1))) % NCTX];
                    uint32_t spins = 0;
                    while (*slot != 0 && ++spins < (1u << 24)) { }
                    if (*slot != 0) stalled = 1;
                    *slot = (uint16_t)(partner | 0x8000);
                }
            }
        }
        if (lane == 0) atomicSub(&busy, 1u);
        if (STATS) t_mark = clock64();
    }
    __syncthreads();
    if (threadIdx.x == 0 && stalled) atomicExch(b.ids_cursor + 1, 1ULL);   // scheduler gave up
    for (int i = threadIdx.x; i < FLD_WINDOW; i += blockDim.x)
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
        if (lane == 0) {
            for (int i = 0; i < 1 + 2 * N_ACTIONS; ++i) atomicAdd(&o[16 + i], (unsigned long long)census[i]);
            for (int i = 0; i < 2 + N_ACTIONS + 6; ++i) atomicAdd(&o[32 + i], cyc[i]);
        }
    }
}

void launch_bucket_build(const DevIndex &ix, uint64_t n_slots, DevBucket *buckets, uint32_t bucket_mask,
                         uint32_t bucket_shift, unsigned long long *report, hipStream_t stream)
{
    DevBucketBuild *filling = reinterpret_cast<DevBucketBuild *>(buckets);
    hipLaunchKernelGGL(bucket_init_kernel, dim3(4096), dim3(256), 0, stream, filling, (uint64_t)bucket_mask + 1);
    hipLaunchKernelGGL(bucket_fill_kernel, dim3(4096), dim3(256), 0, stream, ix.kmers, n_slots, filling,
                       bucket_mask, bucket_shift, report);
    hipLaunchKernelGGL(bucket_pack_kernel, dim3(4096), dim3(256), 0, stream, filling, (uint64_t)bucket_mask + 1);
    hipLaunchKernelGGL(probe_check_kernel, dim3(4096), dim3(256), 0, stream, ix, n_slots, report);
}

void launch_signature_build(const DevBucket *buckets, uint64_t n_buckets, uint64_t *signatures, uint32_t shift,
                            hipStream_t stream)
{
    hipLaunchKernelGGL(signature_build_kernel, dim3(4096), dim3(256), 0, stream, buckets, n_buckets,
                       reinterpret_cast<unsigned long long *>(signatures), shift);
}

void launch_successor_build(const DevIndex &ix, DevContig *records, int64_t n_contigs, int force_lookup,
                            hipStream_t stream)
{
    hipLaunchKernelGGL(successor_build_kernel, dim3(2048), dim3(256), 0, stream, ix, records, n_contigs, force_lookup);
}

void launch_offsets_scan(int64_t *offsets, int64_t n_reads, int64_t base, unsigned long long *out,
                         hipStream_t stream)
{
    if (n_reads <= 0) return;
    int64_t blocks = (n_reads + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(offsets_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, offsets, n_reads, base, out);
    if (base != 0)
        hipLaunchKernelGGL(offsets_rebase_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, offsets,
                           n_reads + 1, base);
}

void launch_offsets_uniform(int64_t *offsets, int64_t n_reads, int64_t read_len, hipStream_t stream)
{
    int64_t blocks = (n_reads + 1 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(offsets_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, offsets, n_reads + 1,
                       read_len);
}

void launch_pack_reads(const uint8_t *bases, const int64_t *offsets, int64_t n_reads,
                       int words_per_read, int record_words, uint32_t *records, hipStream_t stream)
{
    const int64_t total = n_reads * (int64_t)words_per_read;
    if (total == 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pack_reads_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       bases, offsets, n_reads, words_per_read, record_words, records);
}

void launch_unpack_reads(const uint64_t *codes, int64_t stride, int code_words, const uint32_t *lengths,
                         uint32_t uniform_len, int64_t n_reads, int words_per_read, uint32_t *dst,
                         int64_t dst_stride, int *error, hipStream_t stream)
{
    const int64_t total = n_reads * (int64_t)words_per_read;
    if (total == 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(unpack_reads_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, codes, stride, code_words,
                       lengths, uniform_len, n_reads, words_per_read, dst, dst_stride, error);
}

void launch_unpack_exceptions(const uint32_t *exc_reads, const uint32_t *exc_masks, int64_t n_exceptions,
                              int code_words, int64_t first_read, int words_per_read, uint32_t *dst,
                              int64_t dst_stride, hipStream_t stream)
{
    const int64_t total = n_exceptions * (int64_t)code_words;
    if (total == 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(unpack_exceptions_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, exc_reads, exc_masks,
                       n_exceptions, code_words, first_read, words_per_read, dst, dst_stride);
}

void launch_map_units(const DevIndex &ix, const MapBatch &b, int grid_blocks, int stats,
                      hipStream_t stream)
{
    if (b.n_units == 0) return;
    if (stats == 2 && ix.buckets)
        hipLaunchKernelGGL((map_units_kernel<true, true>), dim3(grid_blocks), dim3(MAP_THREADS), 0, stream, ix, b);
    else if (stats)
        hipLaunchKernelGGL((map_units_kernel<true, false>), dim3(grid_blocks), dim3(MAP_THREADS), 0, stream, ix, b);
    else if (ix.buckets)
        hipLaunchKernelGGL((map_units_kernel<false, true>), dim3(grid_blocks), dim3(MAP_THREADS), 0, stream, ix, b);
    else
        hipLaunchKernelGGL((map_units_kernel<false, false>), dim3(grid_blocks), dim3(MAP_THREADS), 0, stream, ix, b);
}

void warm_code_map()
{
    hipFuncAttributes attributes;
    (void)hipFuncGetAttributes(&attributes, reinterpret_cast<const void *>(&map_units_kernel<false, true>));
    (void)hipFuncGetAttributes(&attributes, reinterpret_cast<const void *>(&unpack_reads_kernel));
}

}  // namespace skm
