// Device-side primitives shared by the gfx950 kernels: k-mer arithmetic, the
// reference's SipHash variant, index probing and packed-read access.
// Semantics follow /root/reference/seekmer/_kmer.pxd, _coordinate.pxd and the
// query half of _common.pyx (cited per function); the code is written for
// CDNA4 (64-wide waves, one unit per lane), not translated from the Cython.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace skm {

constexpr int K = 25;                               // _kmer.pxd:9-17
constexpr uint64_t KMER_MASK = (1ULL << 50) - 1;    // _kmer.pxd:31-39
constexpr uint64_t KMER_INVALID = ~0ULL;            // _kmer.pxd:20-28
constexpr int MAX_FRAGMENT_LENGTH = 2000;           // _mapper.pyx:18
constexpr int ALIGN_LENGTH = 8;                     // _mapper.pyx:22
constexpr int MAX_OFFSET = 2;                       // _mapper.pyx:24
constexpr int MAX_DISTANCE = 4;                     // _mapper.pyx:26
constexpr int INVALID_SHIFT = 0x7FFF;               // _mapper.pyx:28

struct Coord { int32_t entry; int32_t offset; };                 // _coordinate.pxd:8-10
struct alignas(16) IndexEntry { uint64_t kmer; Coord pos; };     // _common.pxd:15-17
struct ContigEntry {                                             // _common.pxd:21-27
    int64_t offset, length;
    uint64_t first_kmer, last_kmer;
    int64_t target_offset, target_length;
};

// Contig record as the kernels read it.  The mapper runs at the chip's random sector-request
// ceiling, so what counts is how many 64-byte sectors a contig visit touches and how many
// DEPENDENT round trips a read needs.  The 48-byte reference row is therefore re-packed at upload
// to a 128-byte record of two sectors, ONE PER END OF THE CONTIG, each holding everything a visit
// that works at that end needs:
//   * the contig's place in the pooled bases and its length (both sides carry them);
//   * the 8 bases at this end (the window the alignment step of a hop compares,
//     get_contig_sequence of an edge k-mer, _common.pyx:103-137);
//   * the junction successors of this end.  Every hop of _filter_targets_to_left/right
//     (_mapper.pyx:246-248, :308-310) looks up ONE k-mer that is a pure function of (contig, end,
//     orientation, read base): prepend / append of the anchor contig's edge k-mer (get_tail_kmer,
//     _common.pyx:241-266) with the next base of the read.  There are four such k-mers per end --
//     END side: append(tail k-mer at the contig's end, b); START side: prepend(first_kmer, b) -- and
//     an anchor on the reverse strand asks for the reverse complement of one of the same, whose
//     answer is the stored one with the entry complemented (map_kmer, _common.pyx:84-87).  Their
//     map_kmer results are computed once at upload, with the device's own lookup, and a hop reads
//     its answer from the record of the contig it is leaving: no visit to the k-mer table.  In a
//     built index such a k-mer is the first k-mer of the contig it belongs to or the last one (that
//     is what makes it a junction), so a successor is one word: the signed contig entry and, in the
//     two low bits, SUCC_ABSENT (not in the table: a miss, _mapper.pyx:250, :312), SUCC_AT_START
//     (offset 0), SUCC_AT_END (offset = that contig's length - k, filled in when the hop lands
//     there and its record has arrived) or SUCC_LOOKUP (any other offset, or a k-mer stored without
//     a position: the hop does the table lookup itself, as every hop did before).  Two more bits
//     say what the list merge on the landing contig will do (KMerIndex._filter_on_contig,
//     _common.pyx:185-235), where that is a function of the junction alone: SUCC_WHOLE -- it keeps the
//     read's whole running list, because the landing list holds every entry of this contig's list,
//     of which the running list is a part; SUCC_MASKED -- of THIS contig's list it keeps the entries
//     whose bit is set in `kept` (short lists without a transcript listed twice), which settles the
//     merge as long as the running list is made of positions of this contig's list.  Such a hop
//     goes on to the next alignment step without a merge round;
//   * the contig's first CONTIG_INLINE_TARGETS signed target entries (nine slices in ten have no
//     more): the list arrives with the sector, and a merge (KMerIndex.map_contig /
//     _filter_on_contig, _common.pyx:143-235) is one round trip.  Longer slices live behind the
//     records in the same allocation and the sector holds their place and length instead;
//     `DevIndex::targets` addresses records and overflow alike as one int32 array.
// Which side a visit reads follows from its direction and the anchor's orientation: moving right on
// a forward anchor, or left on a reverse one, works at the contig's END (side 0), the other two at its
// START (side 1) -- and the merge that lands a hop on a contig is followed by the alignment step at
// the same end, so a contig a read hops through costs ONE sector where a record of a row sector + a
// successor sector (the first layout of round 4) cost two.  What only the fall-back paths need -- the
// two edge k-mers themselves, for a hop that must look its junction k-mer up -- lives in an array of
// its own (DevIndex::edge_kmers).
constexpr int CONTIG_INLINE_TARGETS = 8;
constexpr int CONTIG_SHIFT = 7;
constexpr uint32_t SUCC_ABSENT = 0, SUCC_AT_START = 1, SUCC_AT_END = 2, SUCC_LOOKUP = 3;
constexpr uint32_t SUCC_WHOLE = 4;             // bit 2: the landing contig's list holds this contig's whole list
constexpr uint32_t SUCC_MASKED = 8;            // bit 3: DevSide::kept says what the merge on the landing contig keeps
constexpr int SUCC_ENTRY_SHIFT = 4;
constexpr int32_t OFFSET_AT_END = INT32_MAX;   // Coord.offset of a successor until its contig's length is known
struct alignas(64) DevSide {
    int32_t offset, length;            // the contig in the pooled bases
    uint32_t count_edge;               // min(target_length, 0xffff) << 16 | the 8 bases at this end (first on top)
    int32_t succ[4];                   // entry << 4 | SUCC_MASKED | SUCC_WHOLE | SUCC_* of the four junction k-mers of this end
    uint8_t kept[4];                   // SUCC_MASKED: which entries of THIS contig's list (stored order) a hop's merge keeps
    int32_t targets[CONTIG_INLINE_TARGETS];   // the slice; or [0] = its place in DevIndex::targets, [1] = its length
};
struct alignas(128) DevContig { DevSide side[2]; };      // [0] the contig's end, [1] its start
constexpr int SIDE_WORDS = 16, SIDE_TARGETS_WORD = 8;    // int32 words per side / offsetof(DevSide, targets) / 4
static_assert(sizeof(DevSide) == 64 && sizeof(DevContig) == (1u << CONTIG_SHIFT), "a side is one sector");
static_assert(offsetof(DevSide, targets) == 4 * SIDE_TARGETS_WORD, "the inline targets close the sector");

// The k-mer table as the mapper probes it.  The reference's table (linear
// probing from a SipHash home slot, _common.pyx:54-97) is a set: a built index
// stores every k-mer once, under the smaller of itself and its reverse
// complement, and map_kmer returns "the position stored with this k-mer, or
// none" -- which slot holds it does not reach the result.  So the device keeps
// a second copy of the same set in a layout made for 64-byte sectors: buckets
// of four entries (the four k-mers first, then the four positions), addressed
// by a three-multiply hash of the canonical k-mer instead of SipHash-2-4
// (210 of the ~600 instructions of a lookup), overflowing into the next
// bucket.  At one k-mer per bucket on average 98 % of the lookups -- hits and
// misses alike -- end in the bucket they start in: one sector per lookup
// instead of 1.3, and a miss is known from the first 16 bytes.
// A k-mer is kept in canonical form (the smaller of itself and its reverse
// complement: one comparison per entry) with a flag that says the reference's
// table stores the other strand -- map_kmer complements the entry when the
// query is the reverse complement of what is STORED (_common.pyx:84-87) --
// and its 50 bits are split over two words, `low` first for all four entries:
// 31 bits decide "not this one" for all but one query in 2^31, and a k-mer can
// never look like a free entry (bit 31).  The first-hit roll, which looks up
// run after run of k-mers that are not in the table, asks for the 16 bytes of
// `low` only and fetches the rest for the entry that matches them.
// skm_index_create builds it on the device from the reference table and only
// after checking, slot by slot, that the reference's own probe finds every
// stored k-mer where it is stored (true of any built index); a table that
// fails the check is probed in the reference's layout (DevIndex.buckets ==
// nullptr).  The counting build always probes the reference's layout: its
// slot counts define the algorithmic bytes.
struct alignas(64) DevBucket {
    uint32_t low[4];           // bits 0..30 of the canonical k-mer; BUCKET_FREE = free entry
    uint32_t high[4];          // bits 31..49; BUCKET_STORED_RC = the table stores its reverse complement
    Coord pos[4];
};
struct alignas(64) DevBucketBuild {   // the same bytes while the table is being filled (64-bit compare-and-swap)
    uint64_t kmer[4];          // KMER_INVALID = free entry
    Coord pos[4];
};
constexpr uint32_t BUCKET_FREE = 0xFFFFFFFFu, BUCKET_LOW_MASK = 0x7FFFFFFFu;
constexpr uint32_t BUCKET_HIGH_MASK = 0x7FFFFu, BUCKET_STORED_RC = 1u << 19;
static_assert(2 * K == 50, "DevBucket splits a 50-bit k-mer 31 + 19");

// Index as it lives in HBM.  kmers keeps the reference's array layout;
// contigs are re-packed (above), the pooled contig bases go to 2 bits (32
// bases per u64, first base in the top bits) because the mapper only ever
// needs 8-base windows of them, and of each target (entry, offset) only the
// signed transcript entry is kept: nothing on this path reads the offset.
struct DevIndex {
    const IndexEntry *kmers;
    uint32_t slot_mask;
    const DevContig *contigs;
    int64_t n_contigs;
    const uint64_t *seq2;      // 2-bit packed pooled bases, one zero pad word
    int64_t n_bases;
    const int32_t *targets;    // signed transcript entries (Coord.entry of the reference rows); see DevContig
    int64_t n_targets;
    int32_t max_target_count;
    int32_t edge_windows;      // first_kmer/last_kmer agree with the pooled bases on every contig
    int32_t sorted_targets;    // every contig's target slice ascends by signed entry (built indices do)
    int32_t successors;        // every record carries its junction successors (see DevContig)
    const uint64_t *edge_kmers; // [2 * n_contigs]: first_kmer, last_kmer of every contig (fall-back paths)
    const uint64_t *signatures; // [2 << (32 - signature_shift)] or nullptr: which k-mers a minimizer has (see kmer_min_hash)
    uint32_t signature_shift;
    const DevBucket *buckets;  // the same set of k-mers by bucket, or nullptr (see DevBucket)
    uint32_t bucket_mask;      // number of buckets - 1 (a power of two)
    uint32_t bucket_shift;     // bucket = bucket_hash(canonical k-mer) >> bucket_shift
};

__device__ __forceinline__ Coord invalid_coord() { return Coord{0, -1}; }  // _coordinate.pxd:13-24

// Row `index` of the contig table through a 32-bit byte offset from the
// (wave-uniform) base pointer: scalar base + vector offset addressing instead
// of a 64-bit address computation per access (skm_index_create bounds
// n_contigs by 2^26).
__device__ __forceinline__ const DevSide &side_at(const DevIndex &ix, int32_t index, int side)
{
    return *reinterpret_cast<const DevSide *>(reinterpret_cast<const char *>(ix.contigs)
                                              + (((uint32_t)index << CONTIG_SHIFT) + ((uint32_t)side << 6)));
}
// the end of its contig a visit works at: moving right on a forward anchor or left on a reverse one -> 0 (END)
__device__ __forceinline__ int visit_side(bool right, bool forward) { return right == forward ? 0 : 1; }
// a contig's target slice as one of its sides gives it: first element in DevIndex::targets, length
struct Slice { int32_t start, length; };
__device__ __forceinline__ Slice side_slice(const DevSide &s, int32_t index, int side)
{
    const int32_t count = (int32_t)(s.count_edge >> 16);
    if (count <= CONTIG_INLINE_TARGETS)
        return Slice{index * (2 * SIDE_WORDS) + side * SIDE_WORDS + SIDE_TARGETS_WORD, count};
    return Slice{s.targets[0], s.targets[1]};
}

// _kmer.pxd:146-171: reverse the 2-bit groups of the 64-bit word, shift the
// 50 payload bits down, complement.
__device__ __forceinline__ uint64_t kmer_revcomp(uint64_t k)
{
    k = ((k >> 2) & 0x3333333333333333ULL) | ((k & 0x3333333333333333ULL) << 2);
    k = ((k >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((k & 0x0f0f0f0f0f0f0f0fULL) << 4);
    k = __builtin_bswap64(k);
    k >>= (64 - 2 * K);
    return ~k & KMER_MASK;
}

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int s) { return (x << s) | (x >> (64 - s)); }

#define SKM_SIP_HALF(a, b, c, d, s, t) \
    do { a += b; c += d; b = rotl64(b, s) ^ a; d = rotl64(d, t) ^ c; a = rotl64(a, 32); } while (0)
#define SKM_SIP_ROUND(v0, v1, v2, v3) \
    do { SKM_SIP_HALF(v0, v1, v2, v3, 13, 16); SKM_SIP_HALF(v2, v1, v0, v3, 17, 21); } while (0)

// _kmer.pxd:174-219: SipHash-2-4 of one 8-byte word under the fixed key
// (5381, 42) with the reference's non-standard tail (v0 ^= 0 after the length
// block) and truncation to 32 bits.
__device__ __forceinline__ uint32_t kmer_hash(uint64_t kmer)
{
    uint64_t v0 = 5381ULL ^ 0x736f6d6570736575ULL;
    uint64_t v1 = 42ULL ^ 0x646f72616e646f6dULL;
    uint64_t v2 = 5381ULL ^ 0x6c7967656e657261ULL;
    uint64_t v3 = 42ULL ^ 0x7465646279746573ULL;
    v3 ^= kmer;
    SKM_SIP_ROUND(v0, v1, v2, v3);
    SKM_SIP_ROUND(v0, v1, v2, v3);
    v0 ^= kmer;
    v3 ^= (8ULL << 56);
    SKM_SIP_ROUND(v0, v1, v2, v3);
    SKM_SIP_ROUND(v0, v1, v2, v3);
    v2 ^= 0xff;
    SKM_SIP_ROUND(v0, v1, v2, v3);
    SKM_SIP_ROUND(v0, v1, v2, v3);
    SKM_SIP_ROUND(v0, v1, v2, v3);
    SKM_SIP_ROUND(v0, v1, v2, v3);
    return (uint32_t)((v0 ^ v1) ^ (v2 ^ v3));
}

// _kmer.pxd:253-273
__device__ __forceinline__ uint32_t two_bit_encode(uint32_t c)
{
    c &= 0xDFu;                       // fold case (only used on letters below)
    return c == 'T' ? 3u : c == 'G' ? 2u : c == 'C' ? 1u : 0u;
}

// per-lane access counters (only in the STATS instantiation of the map kernel)
struct LaneStats {
    uint32_t lookups, slots, contig_reads, targets_copied, targets_merged,
             seq_fetches, merges;
};

// KMerIndex.map_kmer, _common.pyx:54-97.  Home slot = hash(min(kmer, rc)) &
// (size-1); linear probe with wrap-around; empty slot ends the probe.  The
// table is read one aligned 64-byte sector (four slots) per round trip and the
// slots are examined in probe order from the home slot on: the slots examined
// -- and the result -- are the reference's, a chain that stays inside its
// sector costs one sector visit, and a longer one costs ceil() memory
// latencies instead of one per slot.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
constexpr int PROBE = 4;
template <bool STATS>
__device__ __forceinline__ Coord map_kmer(const DevIndex &ix, uint64_t kmer, LaneStats *st)
{
    const uint64_t rc = kmer_revcomp(kmer);
    const uint32_t home = kmer_hash(kmer < rc ? kmer : rc) & ix.slot_mask;
    if (STATS) st->lookups++;
    uint32_t base = home & ~(uint32_t)(PROBE - 1);
    int skip = (int)(home & (PROBE - 1));
    for (uint64_t n = 0; n <= ix.slot_mask;) {
        u32x4 raw[PROBE];
#pragma unroll
        for (int j = 0; j < PROBE; ++j)
            raw[j] = *reinterpret_cast<const u32x4 *>(&ix.kmers[(base + j) & ix.slot_mask]);
#pragma unroll
        for (int j = 0; j < PROBE; ++j) {
            if (j < skip) continue;
            const uint64_t stored = ((uint64_t)raw[j].y << 32) | raw[j].x;
            if (STATS) st->slots++;
            if (stored == KMER_INVALID) return invalid_coord();
            if (stored == kmer) return Coord{(int32_t)raw[j].z, (int32_t)raw[j].w};
            if (stored == rc) return Coord{~(int32_t)raw[j].z, (int32_t)raw[j].w};
            ++n;
        }
        skip = 0;
        base = (base + PROBE) & ix.slot_mask;
    }
    return invalid_coord();
}

// Hash of the bucket table: its top bits pick the bucket.  (Checked against
// the k-mers of the reference's chr21 test transcriptome: bucket occupancies
// follow the Poisson law to four digits.)
__device__ __forceinline__ uint32_t bucket_hash(uint64_t canonical)
{
    uint32_t h = (uint32_t)canonical * 0x9E3779B1u;
    h = (h ^ (h >> 16)) + (uint32_t)(canonical >> 32) * 0x85EBCA77u;
    return h * 0xC2B2AE3Du;
}

// bucket b of the table through a 32-bit byte offset from the (wave-uniform) base where it fits
__device__ __forceinline__ const DevBucket *bucket_at(const DevIndex &ix, uint32_t b)
{
    return reinterpret_cast<const DevBucket *>(reinterpret_cast<const char *>(ix.buckets) + ((size_t)b << 6));
}

// 16 bytes of a bucket.  (With the non-temporal hint -- a bucket is read once per lookup, 4.3 GB of
// them pass through the caches per sample beside 92 MB of contig records that are read again and
// again -- the launch took 3.5 % longer: profiles/r04_ab_map.log.)
__device__ __forceinline__ u32x4 bucket_load(const u32x4 *p) { return *p; }
// the four k-mers of a bucket (first 32 bytes of its sector)
struct BucketKeys { u32x4 low, high; };
__device__ __forceinline__ BucketKeys bucket_keys(const DevIndex &ix, uint32_t b)
{
    const u32x4 *p = reinterpret_cast<const u32x4 *>(bucket_at(ix, b));
    return BucketKeys{bucket_load(p), bucket_load(p + 1)};
}
__device__ __forceinline__ u32x4 bucket_low(const DevIndex &ix, uint32_t b)
{
    return bucket_load(reinterpret_cast<const u32x4 *>(bucket_at(ix, b)));
}

// Judge the `low` words of a bucket for a canonical k-mer: 0..3 = the one entry that can hold it,
// -1 = not here and the bucket has a free entry (a miss), -2 = not here and the bucket is full (look
// in the next one), -3 = more than one entry shares the 31 bits (judge the whole keys).
__device__ __forceinline__ int bucket_screen(const u32x4 &low, uint64_t canonical)
{
    const uint32_t want = (uint32_t)canonical & BUCKET_LOW_MASK;
    const uint32_t s[4] = {low.x, low.y, low.z, low.w};
    int found = -2, same = 0;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        if (s[j] == BUCKET_FREE) found = -1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (s[j] == want) { found = j; ++same; }
    }
    return same > 1 ? -3 : found;
}

// Judge one bucket for the query (kmer, rc): 0..3 = entry that holds it (flip set when the table
// stores its reverse complement), -1 = not here and the bucket has a free entry (a miss),
// -2 = not here and the bucket is full (look in the next one).
__device__ __forceinline__ int bucket_find(const BucketKeys &k, uint64_t kmer, uint64_t rc, bool &flip)
{
    const uint64_t canonical = kmer < rc ? kmer : rc;
    const uint32_t want_low = (uint32_t)canonical & BUCKET_LOW_MASK, want_high = (uint32_t)(canonical >> 31);
    const uint32_t lo[4] = {k.low.x, k.low.y, k.low.z, k.low.w};
    const uint32_t hi[4] = {k.high.x, k.high.y, k.high.z, k.high.w};
    int found = -2;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        if (lo[j] == BUCKET_FREE) found = -1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (lo[j] == want_low && (hi[j] & BUCKET_HIGH_MASK) == want_high) {
            found = j;
            flip = ((hi[j] & BUCKET_STORED_RC) != 0) != (kmer != canonical);
        }
    }
    return found;
}

// map_kmer over the bucket table (same result as the probe above on every table that passed
// skm_index_create's check); `found` = the k-mer is in the table (whatever position it holds)
__device__ __forceinline__ Coord map_kmer_buckets(const DevIndex &ix, uint64_t kmer, bool &found)
{
    const uint64_t rc = kmer_revcomp(kmer);
    uint32_t b = bucket_hash(kmer < rc ? kmer : rc) >> ix.bucket_shift;
    found = false;
    for (uint32_t n = 0; n <= ix.bucket_mask; ++n) {
        const BucketKeys keys = bucket_keys(ix, b);
        bool flip = false;
        const int j = bucket_find(keys, kmer, rc, flip);
        if (j >= 0) {
            const Coord c = bucket_at(ix, b)->pos[j];
            found = true;
            return flip ? Coord{~c.entry, c.offset} : c;
        }
        if (j == -1) return invalid_coord();
        b = (b + 1) & ix.bucket_mask;
    }
    return invalid_coord();
}
__device__ __forceinline__ Coord map_kmer_buckets(const DevIndex &ix, uint64_t kmer)
{
    bool found;
    return map_kmer_buckets(ix, kmer, found);
}
// The same with the home bucket's four positions asked for together with its keys (the whole
// sector in one go): the position of the entry that matches is then a register select and not a
// second, dependent access -- for the lookups that expect a hit.
struct BucketLoads { BucketKeys keys; u32x4 p01, p23; };       // (asked for; nothing waits for them yet)
__device__ __forceinline__ BucketLoads bucket_loads(const DevIndex &ix, uint64_t kmer)
{
    const uint64_t rc = kmer_revcomp(kmer);
    const uint32_t b = bucket_hash(kmer < rc ? kmer : rc) >> ix.bucket_shift;
    const u32x4 *p = reinterpret_cast<const u32x4 *>(bucket_at(ix, b)->pos);
    return BucketLoads{bucket_keys(ix, b), bucket_load(p), bucket_load(p + 1)};
}
__device__ __forceinline__ Coord map_kmer_in(const DevIndex &ix, const BucketLoads &bucket, uint64_t kmer)
{
    const uint64_t rc = kmer_revcomp(kmer);
    bool flip = false;
    const int j = bucket_find(bucket.keys, kmer, rc, flip);
    if (j >= 0) {
        const uint32_t entry = j == 0 ? bucket.p01.x : j == 1 ? bucket.p01.z : j == 2 ? bucket.p23.x : bucket.p23.z;
        const uint32_t offset = j == 0 ? bucket.p01.y : j == 1 ? bucket.p01.w : j == 2 ? bucket.p23.y : bucket.p23.w;
        return Coord{(int32_t)(flip ? ~entry : entry), (int32_t)offset};
    }
    if (j == -1) return invalid_coord();
    return map_kmer_buckets(ix, kmer);              // a full bucket: the chain from the start
}
__device__ __forceinline__ Coord map_kmer_buckets_whole(const DevIndex &ix, uint64_t kmer)
{
    return map_kmer_in(ix, bucket_loads(ix, kmer), kmer);
}

// The k-mer whose map_kmer result side `side`, successor `b` of a record holds (see DevContig): the
// junction k-mers of _filter_targets_to_right (END side: _kmer.append of the tail k-mer at the
// contig's end with base b, _mapper.pyx:308-310) and of _filter_targets_to_left (START side:
// _kmer.prepend of first_kmer with base b, :246-248) for an anchor on the forward strand.
// get_tail_kmer (_common.pyx:241-266) tells the two edge k-mers apart by `offset == 0`, so a contig
// of exactly k bases has first_kmer at its end too.
__device__ __forceinline__ uint64_t successor_query(uint64_t first_kmer, uint64_t last_kmer, int32_t length,
                                                    int side, int b)
{
    const uint64_t tail_end = length == K ? first_kmer : last_kmer;
    if (side == 0) return ((tail_end << 2) | (uint64_t)b) & KMER_MASK;
    return (first_kmer >> 2) | ((uint64_t)b << (2 * K - 2));
}

// map_kmer of the junction k-mer of a hop that leaves the anchor's contig: to the right
// (append(get_tail_kmer(anchor), base), anchor on the contig's last k-mer in read direction) or to the
// left (prepend(get_tail_kmer(anchor), base), anchor on its first), from the four successors of the
// side the step works at.  A reverse anchor's k-mer is the reverse complement of a forward one with
// the complementary base, and map_kmer of a reverse complement is the same position with the entry
// complemented.  `kind` tells what the offset is (SUCC_*); the Coord's offset is 0 or OFFSET_AT_END.
// What a step of a filter (or the first hit's list) reads of the anchor's contig: one sector, every
// field asked for before the first of them is used.
struct SideVisit {
    int32_t offset, length;
    uint32_t edge8;
    int32_t succ[4];
    uint32_t kept;                 // DevSide::kept, byte b for successor b
    Slice slice;
};
template <bool RIGHT>
__device__ __forceinline__ SideVisit visit(const DevIndex &ix, Coord anchor, bool successors, bool list)
{
    const bool forward = anchor.entry >= 0;
    const int32_t index = forward ? anchor.entry : ~anchor.entry;
    const int side = visit_side(RIGHT, forward);
    const DevSide &record = side_at(ix, index, side);
    SideVisit v;
    v.offset = record.offset;
    v.length = record.length;
    const uint32_t count_edge = record.count_edge;
    v.succ[0] = v.succ[1] = v.succ[2] = v.succ[3] = 0;
    v.kept = 0;
    if (successors) {
        v.succ[0] = record.succ[0]; v.succ[1] = record.succ[1]; v.succ[2] = record.succ[2]; v.succ[3] = record.succ[3];
        v.kept = *reinterpret_cast<const uint32_t *>(record.kept);
    }
    v.slice = Slice{0, 0};
    if (list) {
        const int32_t place = record.targets[0], longer = record.targets[1];
        const int32_t count = (int32_t)(count_edge >> 16);
        v.slice = count <= CONTIG_INLINE_TARGETS
                      ? Slice{index * (2 * SIDE_WORDS) + side * SIDE_WORDS + SIDE_TARGETS_WORD, count}
                      : Slice{place, longer};
    }
    v.edge8 = count_edge & 0xffffu;
    return v;
}
// what the record says about the hop that the read's next base selects
struct Hop {
    Coord landing;                 // map_kmer of the junction k-mer (offset 0 or OFFSET_AT_END), when kind is AT_START / AT_END
    uint32_t kind;                 // SUCC_ABSENT / AT_START / AT_END / LOOKUP
    bool whole, masked;            // SUCC_WHOLE / SUCC_MASKED
    uint32_t kept;                 // SUCC_MASKED: the entries of the leaving contig's list (stored order) that the merge keeps
};
__device__ __forceinline__ Hop junction_successor(const SideVisit &at_side, bool forward, uint32_t base)
{
    const uint32_t b = forward ? base : 3u - base;
    const int32_t word = b == 0 ? at_side.succ[0] : b == 1 ? at_side.succ[1] : b == 2 ? at_side.succ[2] : at_side.succ[3];
    Hop hop;
    hop.kind = (uint32_t)word & 3u;
    hop.whole = ((uint32_t)word & SUCC_WHOLE) != 0;
    hop.masked = ((uint32_t)word & SUCC_MASKED) != 0;
    hop.kept = (at_side.kept >> (8 * b)) & 0xffu;
    const int32_t entry = word >> SUCC_ENTRY_SHIFT;
    hop.landing = Coord{forward ? entry : ~entry, hop.kind == SUCC_AT_START ? 0 : OFFSET_AT_END};
    return hop;
}

// ---- signatures: what the first-hit roll asks before it asks the table --------------------------
// A read with a sequencing error in its first k bases is rolled through up to k k-mers that are not
// in the table, and every one of them costs a sector of a bucket that has nothing to do with the
// one before: the roll is the largest single item of the mapper's sector budget.  Consecutive k-mers
// share their MINIMIZER -- here: the smallest hash among the 15-mers of the k-mer and of its reverse
// complement, 22 values of which a k-mer and its successor share 20 -- for five steps on average,
// so a small table indexed by that hash is read once for a run of k-mers.  (15 bases: with 11 there
// are 4 M possible minimizers of which a random order favours ~300 k, and 65 M k-mers saturate their
// signatures; 13, 14 and 15 measured 5.31, 5.15 and 5.13 ms per launch.)  It holds, per slot, a 128-bit signature: two bits of it (signature_bits, from
// bucket_hash(canonical k-mer)) are set for every k-mer of the table whose minimizer hashes there.
// A k-mer with one of its bits clear is not in the table -- no false negatives: slots that collide
// only add bits --; one with both set is looked up as before.
#ifndef SKM_MINIMIZER_BASES
#define SKM_MINIMIZER_BASES 15
#endif
constexpr int MINIMIZER_BASES = SKM_MINIMIZER_BASES;
static_assert(MINIMIZER_BASES >= 8 && MINIMIZER_BASES <= 15 && MINIMIZER_BASES < K, "a minimizer is hashed as a 32-bit word");
constexpr int MINIMIZERS_PER_KMER = K - MINIMIZER_BASES + 1;      // 15 per strand
__device__ __forceinline__ uint32_t mmer_hash(uint32_t mmer)
{
    uint32_t h = mmer * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return h ^ (h >> 13);
}
__device__ __forceinline__ uint32_t kmer_min_hash(uint64_t kmer)
{
    const uint64_t rc = kmer_revcomp(kmer);
    uint32_t least = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < MINIMIZERS_PER_KMER; ++i) {
        least = min(least, mmer_hash((uint32_t)(kmer >> (2 * i)) & ((1u << (2 * MINIMIZER_BASES)) - 1u)));
        least = min(least, mmer_hash((uint32_t)(rc >> (2 * i)) & ((1u << (2 * MINIMIZER_BASES)) - 1u)));
    }
    return least;
}
struct Signature { uint64_t lo, hi; };           // 128 bits per slot
__device__ __forceinline__ Signature signature_bits(uint32_t bucket_hash_of_kmer)
{
    const uint32_t a = bucket_hash_of_kmer & 127u, b = (bucket_hash_of_kmer >> 7) & 127u;
    Signature s{0, 0};
    if (a < 64) s.lo |= 1ULL << a; else s.hi |= 1ULL << (a - 64);
    if (b < 64) s.lo |= 1ULL << b; else s.hi |= 1ULL << (b - 64);
    return s;
}
// the slot of a minimizer hash: the smallest of 30 hashes is a small number -- mixed again (a
// bijection) before its top bits pick the slot
__device__ __forceinline__ uint32_t signature_slot(uint32_t least, uint32_t shift)
{
    return mmer_hash(least ^ 0x5bd1e995u) >> shift;
}
// reverse complement of 32 bases in a word (first base in the top bits)
__device__ __forceinline__ uint64_t revcomp32(uint64_t k)
{
    k = ((k >> 2) & 0x3333333333333333ULL) | ((k & 0x3333333333333333ULL) << 2);
    k = ((k >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((k & 0x0f0f0f0f0f0f0f0fULL) << 4);
    return ~__builtin_bswap64(k);
}

// 32 consecutive 2-bit codes starting at base `p` of a packed array (first
// base in the top bits).  The arrays carry one pad word, so w+1 is readable.
__device__ __forceinline__ uint64_t packed_window(const uint64_t *words, int64_t p)
{
    const int64_t w = p >> 5;
    const int s = (int)(p & 31) << 1;
    const uint64_t hi = words[w];
    if (s == 0) return hi;
    return (hi << s) | (words[w + 1] >> (64 - s));
}

// reverse-complement of 8 packed bases (16 bits)
__device__ __forceinline__ uint32_t revcomp8(uint32_t v)
{
    v = ((v >> 2) & 0x3333u) | ((v & 0x3333u) << 2);
    v = ((v >> 4) & 0x0f0fu) | ((v & 0x0f0fu) << 4);
    v = ((v >> 8) & 0x00ffu) | ((v & 0x00ffu) << 8);
    return ~v & 0xffffu;
}

// KMerIndex.get_contig_sequence for |length| == 8, _common.pyx:103-137:
// leading (length>0) or trailing (length<0) 8 bases of the anchored k-mer in
// read orientation, as 16 bits (first base on top).  The pool index is
// clamped so that an inconsistent index can never fault the GPU.
template <bool STATS>
__device__ __forceinline__ uint32_t contig8(const DevIndex &ix, Coord c, int32_t contig_offset, bool leading, LaneStats *st)
{
    int64_t offset = (int64_t)contig_offset + c.offset;
    if (STATS) { st->contig_reads++; st->seq_fetches++; }
    if (c.entry >= 0) offset += leading ? ALIGN_LENGTH : K;
    else offset += leading ? K : ALIGN_LENGTH;
    int64_t first = offset - ALIGN_LENGTH;
    if (first < 0) first = 0;
    if (first > ix.n_bases - ALIGN_LENGTH) first = ix.n_bases - ALIGN_LENGTH;
    uint32_t v = (uint32_t)(packed_window(ix.seq2, first) >> 48);
    if (c.entry < 0) v = revcomp8(v);
    return v;
}

// The same window when the anchor sits on the first or last k-mer of its contig
// (every in-loop step of _filter_targets_to_left/right, _mapper.pyx:229-246 and
// :285-308, moves it there): the contig's first 8 bases are the top 16 bits of
// first_kmer and its last 8 the low 16 bits of last_kmer, copied at upload into
// the side the step reads anyway (DevSide::count_edge), so the pool is not
// touched.  skm_index_create checks first_kmer/last_kmer against the pooled
// bases; an index where they disagree takes the pool path (edge_windows = 0).
// The side is the one of the step's direction (SideVisit): leading windows belong
// to left steps, trailing ones to right steps.
template <bool STATS>
__device__ __forceinline__ uint32_t contig8_edge(const DevIndex &ix, Coord c, const SideVisit &at_side, bool leading,
                                                 LaneStats *st)
{
    if (!ix.edge_windows) return contig8<STATS>(ix, c, at_side.offset, leading, st);
    if (STATS) { st->contig_reads++; st->seq_fetches++; }
    return c.entry < 0 ? revcomp8(at_side.edge8) : at_side.edge8;
}

// KMerIndex.get_tail_kmer, _common.pyx:241-266
template <bool STATS>
__device__ __forceinline__ uint64_t tail_kmer(const DevIndex &ix, Coord c, LaneStats *st)
{
    int32_t index = c.entry < 0 ? ~c.entry : c.entry;
    if (STATS) st->contig_reads++;
    uint64_t k = ix.edge_kmers[2 * (int64_t)index + (c.offset == 0 ? 0 : 1)];
    if (c.entry < 0) k = kmer_revcomp(k);
    return k;
}

// A read in packed form: 2-bit codes (N and everything else that is not
// ACGT/acgt encode as 0, _kmer.pxd:253-273) plus one bit per base that says
// "is upper-case ACGT" -- the only property _match_base needs
// (_mapper.pyx:500-501).
// One record per read, 64-byte aligned: W code words (u64), W mask words (u32,
// 32 bases per word, first base in the top bit), then the read length (u32).
// A view addresses its record as (wave-uniform base, 32-bit byte offset): the
// loads then take the scalar-base + vector-offset form and no lane computes a
// 64-bit address.
struct ReadView {
    const char *base;          // wave-uniform: the first record of the block's range
    uint32_t off;              // byte offset of this read's record from `base`
    int words;                 // W
    int len;
    __device__ __forceinline__ uint64_t code_word(int w) const
    {
        return *reinterpret_cast<const uint64_t *>(base + (off + ((uint32_t)w << 3)));
    }
    __device__ __forceinline__ uint32_t mask_word(int w) const
    {
        return *reinterpret_cast<const uint32_t *>(base + (off + ((uint32_t)(2 * words + w) << 2)));
    }
    __device__ __forceinline__ uint32_t stored_length() const
    {
        return *reinterpret_cast<const uint32_t *>(base + (off + ((uint32_t)(3 * words) << 2)));
    }
};
// record `local` (counted from the block's first record) of a batch with record_words u32 per record
__device__ __forceinline__ ReadView read_view(const char *block_records, int record_words, int words_per_read,
                                              uint32_t local)
{
    ReadView r{block_records, local * ((uint32_t)record_words << 2), words_per_read, 0};
    r.len = (int)r.stored_length();
    return r;
}

// 32 consecutive codes of the read starting at base p (the record carries one pad word)
__device__ __forceinline__ uint64_t read_window(const ReadView &r, int p)
{
    const int w = p >> 5;
    const int s = (p & 31) << 1;
    const uint64_t hi = r.code_word(w);
    if (s == 0) return hi;
    return (hi << s) | (r.code_word(w + 1) >> (64 - s));
}
__device__ __forceinline__ uint64_t read_kmer(const ReadView &r, int p)      // _kmer.pxd:46-68
{
    return read_window(r, p) >> (64 - 2 * K);
}
// the 16 codes of aligned half word h (bases 16h .. 16h+15), first base on top
__device__ __forceinline__ uint32_t read_half(const ReadView &r, int h)
{
    return *reinterpret_cast<const uint32_t *>(r.base + (r.off + ((uint32_t)(h ^ 1) << 2)));
}
__device__ __forceinline__ uint32_t read_code(const ReadView &r, int p)
{
    return (uint32_t)(r.code_word(p >> 5) >> (62 - 2 * (p & 31))) & 3u;
}
// 16 bases of codes (32 bits) and 16 "is ACGT" bits starting at base p
__device__ __forceinline__ void read_window16(const ReadView &r, int p, uint32_t &codes, uint32_t &acgt)
{
    codes = (uint32_t)(read_window(r, p) >> 32);
    const int w = p >> 5, s = p & 31;
    uint64_t m = ((uint64_t)r.mask_word(w) << 32) | r.mask_word(w + 1);
    acgt = (uint32_t)((m << s) >> 48);
}

}  // namespace skm
