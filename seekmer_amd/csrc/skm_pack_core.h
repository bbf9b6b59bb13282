// FASTQ text -> the mapper's 2-bit read codes, on the host.
//
// The mapper's 64-byte read record (csrc/skm_map.hip: pack_reads_kernel) holds a read as 2-bit
// codes, 32 bases per u64 word with the first base in the top two bits (A0 C1 G2 T3, anything
// else 0: /root/reference/seekmer/_kmer.pxd:253-273), plus one bit per base that says "this
// character is an upper-case A, C, G or T" (the SIFT4 checks treat every other read character as a
// wildcard: /root/reference/seekmer/_mapper.pyx:500-501).  In a FASTQ file nearly every read has
// all of those bits set, so what crosses PCIe is the code words alone (32 bytes for a 100-base
// read instead of 100) and, for the few reads with an N or a lower-case letter, an exception
// entry that carries their bit plane.
//
// This header is the whole parser core.  It is compiled three times (skm_pack_scalar.cpp,
// skm_pack_ssse3.cpp, skm_pack_avx2.cpp: the same code over 1-, 16- and 32-byte blocks); the reader
// (skm_fastq_packed.cpp) picks one at run time.  The record walk is the reference's line rule
// (/root/reference/seekmer/common.py:126-197): from a line with i & 3 == 0, line i & 3 == 0 is the
// name (strip()[1:]), line i & 3 == 1 the bases (strip(), case kept), the other two are skipped
// without being looked at -- every newline is found, none is predicted.
#pragma once
#include "skm_fastq_shared.h"

namespace skmfq {

// the reads of one walk (one piece of one file)
struct PackedOut {
    Buf<uint64_t> codes;           // [n_reads][cw]
    Buf<uint32_t> lengths;         // [n_reads]
    Buf<uint32_t> exc_reads;       // reads with a character that is not upper-case ACGT, ascending
    Buf<uint32_t> exc_masks;       // [n_exc][cw]: bit 31 - i of word w = base 32 w + i is upper-case ACGT
    Buf<char> names;               // only when want_names
    Buf<int64_t> name_offsets;     // [n_reads + 1]
    int cw = 1;
    int64_t n_reads = 0;
    int64_t uniform_len = -2;      // -2 no read yet, >= 0 every read so far this long, -1 ragged
    bool want_names = false;

    // The code words -- nearly all of what crosses PCIe -- live in the caller's kind of memory
    // (page-locked: the copy is DMA straight out of the array).  Lengths (not sent at all when the
    // reads of a piece are equally long) and exception entries (a few per thousand reads) stay in
    // plain memory: page-locking costs about a millisecond per allocation whatever its size.
    void use(alloc_fn al, free_fn fr)
    {
        if (codes.al == al && codes.fr == fr) return;
        codes.release();
        codes.al = al; codes.fr = fr;
    }
    void start(int code_words, bool with_names)
    {
        codes.clear(); lengths.clear(); exc_reads.clear(); exc_masks.clear(); names.clear(); name_offsets.clear();
        cw = code_words < 1 ? 1 : code_words;
        n_reads = 0;
        uniform_len = -2;
        want_names = with_names;
        if (with_names) name_offsets.push(0);
    }
    bool failed() const
    {
        return codes.failed || lengths.failed || exc_reads.failed || exc_masks.failed || names.failed || name_offsets.failed;
    }
    void release()
    {
        codes.release(); lengths.release(); exc_reads.release(); exc_masks.release(); names.release(); name_offsets.release();
    }
    PackedOut() = default;
    PackedOut(const PackedOut &) = delete;
    PackedOut &operator=(const PackedOut &) = delete;
    ~PackedOut() { release(); }
    size_t bytes() const { return codes.cap * 8 + (lengths.cap + exc_reads.cap + exc_masks.cap) * 4; }
    // more words per read: the reads so far move apart, back to front
    void restride(int new_cw)
    {
        if (new_cw <= cw) return;
        codes.reserve((size_t)(n_reads + 1) * new_cw);
        if (codes.failed) return;
        for (int64_t r = n_reads - 1; r >= 0; --r) {
            uint64_t *from = codes.p + r * cw, *to = codes.p + r * new_cw;
            for (int w = new_cw - 1; w >= 0; --w) to[w] = w < cw ? from[w] : 0;
        }
        codes.n = (size_t)n_reads * new_cw;
        const int64_t n_exc = (int64_t)exc_reads.n;
        exc_masks.reserve((size_t)(n_exc + 1) * new_cw);
        if (exc_masks.failed) return;
        for (int64_t e = n_exc - 1; e >= 0; --e) {
            uint32_t *from = exc_masks.p + e * cw, *to = exc_masks.p + e * new_cw;
            for (int w = new_cw - 1; w >= 0; --w) to[w] = w < cw ? from[w] : 0;
        }
        exc_masks.n = (size_t)n_exc * new_cw;
        cw = new_cw;
    }
};

inline bool pack_is_space(char c)        // bytes.strip() with no argument
{
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f';
}

// _kmer._two_bit_encode (/root/reference/seekmer/_kmer.pxd:253-273): T/t 3, G/g 2, C/c 1, else 0
inline uint64_t pack_code(unsigned char c)
{
    switch (c) {
    case 'T': case 't': return 3;
    case 'G': case 'g': return 2;
    case 'C': case 'c': return 1;
    default: return 0;
    }
}

// One read the slow way (any characters): codes into words[cw] (zero-filled), the ACGT bit plane into
// mask[cw]; true when some character is not an upper-case A, C, G or T.
inline bool pack_any(const char *p, size_t len, uint64_t *words, uint32_t *mask, int cw)
{
    for (int w = 0; w < cw; ++w) { words[w] = 0; mask[w] = 0; }
    bool odd = false;
    for (size_t i = 0; i < len; ++i) {
        const unsigned char c = (unsigned char)p[i];
        words[i >> 5] |= pack_code(c) << (62 - 2 * (i & 31));
        const bool plain = c == 'A' || c == 'C' || c == 'G' || c == 'T';
        if (plain) mask[i >> 5] |= 1u << (31 - (i & 31));
        else odd = true;
    }
    return odd;
}

// The walkers.  walk_*: whole records from byte `start` (a line with i & 3 == 0) until a record
// would start at or after `stop` or the text ends; returns where it stopped (a record start, or n).
// A last record counts as soon as it has its bases line; a trailing name without one is dropped.
typedef size_t (*walk_fn)(const char *text, size_t n, size_t start, size_t stop, PackedOut &out);
size_t walk_scalar(const char *text, size_t n, size_t start, size_t stop, PackedOut &out);
size_t walk_ssse3(const char *text, size_t n, size_t start, size_t stop, PackedOut &out);
size_t walk_avx2(const char *text, size_t n, size_t start, size_t stop, PackedOut &out);
// span_*: one read given as (pointer, length), appended to `out` (skm_pack_reads)
typedef void (*span_fn)(const char *p, size_t len, PackedOut &out);
void span_scalar(const char *p, size_t len, PackedOut &out);
void span_ssse3(const char *p, size_t len, PackedOut &out);
void span_avx2(const char *p, size_t len, PackedOut &out);

}  // namespace skmfq

// ============================================================================================
#ifdef SKM_PACK_VARIANT
// The variant's block primitives come first (find_newline, scan_encode), then the shared walk.

#if SKM_PACK_VARIANT == 1 || SKM_PACK_VARIANT == 2
#include <immintrin.h>
#endif

namespace skmfq {
namespace {

#if SKM_PACK_VARIANT == 0
// ---- scalar
inline const char *find_newline(const char *p, const char *end)
{
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    return nl ? nl : end;
}

// the leading upper-case ACGT characters of p[0 .. avail), at most `cap` of them: their codes into
// words[0 .. cw) (the rest of the cw words zero); returns how many there are
inline size_t scan_encode(const char *p, size_t avail, size_t cap, uint64_t *words, int cw)
{
    const size_t limit = avail < cap ? avail : cap;
    for (int w = 0; w < cw; ++w) words[w] = 0;
    size_t i = 0;
    for (; i < limit; ++i) {
        const unsigned char c = (unsigned char)p[i];
        uint64_t code;
        if (c == 'A') code = 0; else if (c == 'C') code = 1; else if (c == 'G') code = 2; else if (c == 'T') code = 3;
        else break;
        words[i >> 5] |= code << (62 - 2 * (i & 31));
    }
    return i;
}
#endif

#if SKM_PACK_VARIANT == 1
// ---- 16-byte blocks (SSSE3)
inline const char *find_newline(const char *p, const char *end)
{
    const __m128i nl = _mm_set1_epi8('\n');
    while (p + 16 <= end) {
        const unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)p), nl));
        if (m) return p + __builtin_ctz(m);
        p += 16;
    }
    while (p < end && *p != '\n') ++p;
    return p;
}

// 16 characters: bit i of *ok = character i is an upper-case A, C, G or T; returns the 32 code bits
// (first character in the top two bits; characters that are not ACGT contribute garbage)
inline uint32_t encode16(const char *p, unsigned *ok)
{
    const __m128i v = _mm_loadu_si128((const __m128i *)p);
    const __m128i idx = _mm_and_si128(_mm_srli_epi16(v, 1), _mm_set1_epi8(3));          // A0 C1 T2 G3
    const __m128i letters = _mm_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    *ok = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, _mm_shuffle_epi8(letters, idx)));
    const __m128i code = _mm_xor_si128(idx, _mm_and_si128(_mm_srli_epi16(idx, 1), _mm_set1_epi8(1)));   // A0 C1 G2 T3
    const __m128i pairs = _mm_maddubs_epi16(code, _mm_set1_epi16(0x0104));             // c0 * 4 + c1
    const __m128i quads = _mm_madd_epi16(pairs, _mm_set1_epi32(0x00010010));           // (..) * 16 + (..)
    const __m128i bytes = _mm_shuffle_epi8(quads, _mm_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
    return (uint32_t)_mm_cvtsi128_si32(bytes);
}

inline size_t scan_encode(const char *p, size_t avail, size_t cap, uint64_t *words, int cw)
{
    const size_t limit = avail < cap ? avail : cap;
    size_t i = 0;
    int w = 0;
    uint64_t acc = 0;
    bool half = false;                 // acc holds the first 16 bases of word w
    for (;;) {
        if (i + 16 > avail) break;     // (the block must lie inside the text)
        if (i >= limit) break;
        unsigned ok;
        uint32_t val = encode16(p + i, &ok);
        const size_t room = limit - i;
        unsigned good = (unsigned)__builtin_ctz(~ok);                 // leading ACGT characters (ok has 16 bits)
        if (good > room) good = (unsigned)room;
        if (good < 16) val = good ? (val & ~(0xffffffffu >> (2 * good))) : 0;
        if (!half) acc = (uint64_t)val << 32; else { acc |= val; }
        if (good < 16) {
            i += good;
            if (good || half) { words[w++] = acc; }
            half = false;
            acc = 0;
            goto done;
        }
        i += 16;
        if (half) { words[w++] = acc; acc = 0; }
        half = !half;
    }
    // fewer than 16 characters of text left (or the cap reached): one at a time
    for (; i < limit; ++i) {
        const unsigned char c = (unsigned char)p[i];
        uint64_t code;
        if (c == 'A') code = 0; else if (c == 'C') code = 1; else if (c == 'G') code = 2; else if (c == 'T') code = 3;
        else break;
        acc |= code << (62 - 2 * (i & 31));
        if ((i & 31) == 31) { words[w++] = acc; acc = 0; }
    }
    if ((i & 31) != 0 && w < cw && w == (int)(i >> 5)) words[w++] = acc;
done:
    for (; w < cw; ++w) words[w] = 0;
    return i;
}
#endif

#if SKM_PACK_VARIANT == 2
// ---- 32-byte blocks (AVX2)
inline const char *find_newline(const char *p, const char *end)
{
    const __m256i nl = _mm256_set1_epi8('\n');
    while (p + 32 <= end) {
        const unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)p), nl));
        if (m) return p + __builtin_ctz(m);
        p += 32;
    }
    while (p < end && *p != '\n') ++p;
    return p;
}

inline uint64_t encode32(const char *p, unsigned *ok)
{
    const __m256i v = _mm256_loadu_si256((const __m256i *)p);
    const __m256i idx = _mm256_and_si256(_mm256_srli_epi16(v, 1), _mm256_set1_epi8(3));
    const __m256i letters = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                             'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    *ok = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_shuffle_epi8(letters, idx)));
    const __m256i code = _mm256_xor_si256(idx, _mm256_and_si256(_mm256_srli_epi16(idx, 1), _mm256_set1_epi8(1)));
    const __m256i pairs = _mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0104));
    const __m256i quads = _mm256_madd_epi16(pairs, _mm256_set1_epi32(0x00010010));
    const __m256i pick = _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m256i bytes = _mm256_shuffle_epi8(quads, pick);
    const uint32_t first = (uint32_t)_mm256_extract_epi32(bytes, 0), second = (uint32_t)_mm256_extract_epi32(bytes, 4);
    return ((uint64_t)first << 32) | second;
}

inline size_t scan_encode(const char *p, size_t avail, size_t cap, uint64_t *words, int cw)
{
    const size_t limit = avail < cap ? avail : cap;
    size_t i = 0;
    int w = 0;
    uint64_t acc = 0;
    while (i + 32 <= avail && i < limit) {
        unsigned ok;
        uint64_t val = encode32(p + i, &ok);
        const size_t room = limit - i;
        unsigned good = ok == 0xffffffffu ? 32u : (unsigned)__builtin_ctz(~ok);
        if (good > room) good = (unsigned)room;
        if (good < 32) {
            if (good) words[w++] = val & ~(~0ull >> (2 * good));
            i += good;
            goto done;
        }
        words[w++] = val;
        i += 32;
    }
    for (; i < limit; ++i) {
        const unsigned char c = (unsigned char)p[i];
        uint64_t code;
        if (c == 'A') code = 0; else if (c == 'C') code = 1; else if (c == 'G') code = 2; else if (c == 'T') code = 3;
        else break;
        acc |= code << (62 - 2 * (i & 31));
        if ((i & 31) == 31) { words[w++] = acc; acc = 0; }
    }
    if ((i & 31) != 0 && w < cw && w == (int)(i >> 5)) words[w++] = acc;
done:
    for (; w < cw; ++w) words[w] = 0;
    return i;
}
#endif

// ---- shared: one read into `out`
// the read is text[b .. e): the characters before `known` are upper-case ACGT and already encoded in the
// read's slot when `known` == e - b
inline void commit_read(PackedOut &out, int64_t len)
{
    out.lengths.push((uint32_t)len);
    out.uniform_len = out.uniform_len == -2 ? len : (out.uniform_len == len ? out.uniform_len : -1);
    out.codes.n += (size_t)out.cw;
    out.n_reads++;
}

// a read that needs the slow rule: [p, p + len) after strip(), any characters
inline void slow_read(PackedOut &out, const char *p, size_t len)
{
    while (len && pack_is_space(p[0])) { ++p; --len; }
    while (len && pack_is_space(p[len - 1])) --len;
    const int need = (int)((len + 31) / 32);
    if (need > out.cw) out.restride(need);
    out.codes.reserve(out.codes.n + (size_t)out.cw);
    out.exc_masks.reserve(out.exc_masks.n + (size_t)out.cw);
    if (out.codes.failed || out.exc_masks.failed) return;
    if (pack_any(p, len, out.codes.p + out.codes.n, out.exc_masks.p + out.exc_masks.n, out.cw)) {
        out.exc_reads.push((uint32_t)out.n_reads);
        out.exc_masks.n += (size_t)out.cw;
    }
    commit_read(out, (int64_t)len);
}

inline void add_name(PackedOut &out, const char *p, size_t len)
{
    while (len && pack_is_space(p[0])) { ++p; --len; }
    while (len && pack_is_space(p[len - 1])) --len;
    if (len) { ++p; --len; }                                      // strip()[1:]
    out.names.append(p, len);
    out.name_offsets.push((int64_t)out.names.n);
}

inline size_t walk_impl(const char *text, size_t n, size_t start, size_t stop, PackedOut &out)
{
    const char *const end = text + n;
    size_t at = start;
    if (out.codes.cap == 0 && start < n) {
        // a piece without arrays yet: size them for the whole range from the first record's extent
        // BEFORE anything is allocated (growing a page-locked array means locking a new one)
        const char *q = text + start;
        size_t bases_len = 0;
        for (int line = 0; line < 4 && q < end; ++line) {
            const char *line_end = find_newline(q, end);
            if (line == 1) bases_len = (size_t)(line_end - q);
            q = line_end < end ? line_end + 1 : end;
        }
        const size_t record = (size_t)(q - (text + start)), span = (stop < n ? stop : n) - start;
        if (out.n_reads == 0 && (int)((bases_len + 31) / 32) > out.cw && bases_len < (1u << 20))
            out.cw = (int)((bases_len + 31) / 32);             // (nothing stored yet: no re-striding)
        if (record >= 8) {
            const size_t expect = span / record + span / record / 16 + 64;
            out.codes.reserve(expect * (size_t)out.cw);
            out.lengths.reserve(expect);
        }
    }
    while (at < stop && at < n) {
        // line i & 3 == 0: the name
        const char *name = text + at;
        const char *name_end = find_newline(name, end);
        if (name_end == end) { at = n; break; }                   // a name and nothing after it: dropped
        const char *bases = name_end + 1;
        if (bases >= end) { at = n; break; }
        // line i & 3 == 1: the bases
        out.codes.reserve(out.codes.n + (size_t)out.cw);
        if (out.codes.failed) return n;
        const size_t avail = (size_t)(end - bases);
        const size_t got = scan_encode(bases, avail, (size_t)out.cw * 32, out.codes.p + out.codes.n, out.cw);
        const char *line_end;
        if (got < avail && bases[got] == '\n') {                  // the plain case
            line_end = bases + got;
            commit_read(out, (int64_t)got);
        } else if (got + 1 < avail && bases[got] == '\r' && bases[got + 1] == '\n') {
            line_end = bases + got + 1;
            commit_read(out, (int64_t)got);
        } else if (got == avail) {                                // the text ends inside the bases line
            line_end = end;
            commit_read(out, (int64_t)got);
        } else {                                                  // N, lower case, blanks, a longer read
            line_end = find_newline(bases + got, end);
            slow_read(out, bases, (size_t)(line_end - bases));
            if (out.failed()) return n;
        }
        if (out.want_names) add_name(out, name, (size_t)(name_end - name));
        if (line_end >= end) { at = n; break; }
        // lines i & 3 == 2 and 3: never looked at, only their ends are found
        const char *plus_end = find_newline(line_end + 1, end);
        if (plus_end >= end) { at = n; break; }
        const char *qual_end = find_newline(plus_end + 1, end);
        at = qual_end >= end ? n : (size_t)(qual_end - text) + 1;
        if (out.n_reads == 1 && at > start) {
            // size the arrays for the whole range from the first record (growing a page-locked array
            // means locking a new one): records of this size, and some slack
            const size_t span = (stop < n ? stop : n) - start, record = at - start;
            const size_t expect = span / record + span / record / 16 + 64;
            out.codes.reserve(expect * (size_t)out.cw);
            out.lengths.reserve(expect);
        }
    }
    return at;
}

inline void span_impl(const char *p, size_t len, PackedOut &out)
{
    const int need = (int)((len + 31) / 32);
    if (need > out.cw) out.restride(need);
    out.codes.reserve(out.codes.n + (size_t)out.cw);
    if (out.codes.failed) return;
    const size_t got = scan_encode(p, len, (size_t)out.cw * 32, out.codes.p + out.codes.n, out.cw);
    if (got == len) commit_read(out, (int64_t)len);
    else {                                            // (no strip here: the span IS the read)
        out.exc_masks.reserve(out.exc_masks.n + (size_t)out.cw);
        if (out.exc_masks.failed) return;
        if (pack_any(p, len, out.codes.p + out.codes.n, out.exc_masks.p + out.exc_masks.n, out.cw)) {
            out.exc_reads.push((uint32_t)out.n_reads);
            out.exc_masks.n += (size_t)out.cw;
        }
        commit_read(out, (int64_t)len);
    }
}

}  // namespace

#if SKM_PACK_VARIANT == 0
size_t walk_scalar(const char *text, size_t n, size_t start, size_t stop, PackedOut &out) { return walk_impl(text, n, start, stop, out); }
void span_scalar(const char *p, size_t len, PackedOut &out) { span_impl(p, len, out); }
#elif SKM_PACK_VARIANT == 1
size_t walk_ssse3(const char *text, size_t n, size_t start, size_t stop, PackedOut &out) { return walk_impl(text, n, start, stop, out); }
void span_ssse3(const char *p, size_t len, PackedOut &out) { span_impl(p, len, out); }
#else
size_t walk_avx2(const char *text, size_t n, size_t start, size_t stop, PackedOut &out) { return walk_impl(text, n, start, stop, out); }
void span_avx2(const char *p, size_t len, PackedOut &out) { span_impl(p, len, out); }
#endif

}  // namespace skmfq
#endif  // SKM_PACK_VARIANT
