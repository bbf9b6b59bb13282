// Native index builder (libseekmer_host.so): produces the four index arrays
// exactly as the reference's ContigAssembler does
// (/root/reference/seekmer/_index_builder.pyx:85-572), because contig
// numbering and orientation -- fixed by the slot-order walk of a linear-probing
// table whose size depends on the insertion history -- decide the order of the
// ids inside every equivalence-class tuple.
//
// The passes over the transcriptome are memory-latency bound (one random
// 16-byte slot per k-mer in a table of up to 2 GiB), so every pass runs a
// software pipeline: k-mers are produced in blocks, their home slots are
// prefetched, and only then is the strictly sequential table update applied
// in the original order.  Prefetching changes no result.
#include "../../include/seekmer_hip.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int K = 25;
constexpr uint64_t MASK = (1ULL << 50) - 1;
constexpr uint64_t EMPTY = ~0ULL;
constexpr int32_t NONE = 0x7FFFFFFF;          // _common.pxd:10
constexpr int BLOCK = 32;

struct Node { uint64_t kmer; int32_t last; int32_t next; };   // _index_builder.pyx:79-82
struct Contig { int64_t offset, length; uint64_t first_kmer, last_kmer; int64_t target_offset, target_count; };
struct Target { int32_t entry, offset; };

inline uint64_t code(char b)                                   // _kmer.pxd:253-273
{
    switch (b) {
    case 'T': case 't': return 3;
    case 'G': case 'g': return 2;
    case 'C': case 'c': return 1;
    default: return 0;
    }
}

inline uint64_t revcomp(uint64_t k)                            // _kmer.pxd:146-171
{
    k = ((k >> 2) & 0x3333333333333333ULL) | ((k & 0x3333333333333333ULL) << 2);
    k = ((k >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((k & 0x0f0f0f0f0f0f0f0fULL) << 4);
    k = __builtin_bswap64(k);
    return ~(k >> 14) & MASK;
}

inline uint64_t rotl(uint64_t x, int s) { return (x << s) | (x >> (64 - s)); }

inline uint32_t sip(uint64_t m)                                // _kmer.pxd:174-219
{
    uint64_t v0 = 5381ULL ^ 0x736f6d6570736575ULL, v1 = 42ULL ^ 0x646f72616e646f6dULL;
    uint64_t v2 = 5381ULL ^ 0x6c7967656e657261ULL, v3 = 42ULL ^ 0x7465646279746573ULL;
    auto round = [&]() {
        v0 += v1; v2 += v3; v1 = rotl(v1, 13) ^ v0; v3 = rotl(v3, 16) ^ v2; v0 = rotl(v0, 32);
        v2 += v1; v0 += v3; v1 = rotl(v1, 17) ^ v2; v3 = rotl(v3, 21) ^ v0; v2 = rotl(v2, 32);
    };
    v3 ^= m; round(); round(); v0 ^= m;
    v3 ^= 8ULL << 56; round(); round();
    v2 ^= 0xff; round(); round(); round(); round();
    return (uint32_t)(v0 ^ v1 ^ v2 ^ v3);
}

struct KmerRef { uint64_t kmer, rc; uint32_t hash; };

struct Assembler {
    Node *t = nullptr;
    int64_t size = 0;
    int64_t count = 0;
    bool last_new = true, last_fwd = true;
    int32_t last = NONE;

    ~Assembler() { free(t); }

    void wipe(int64_t from, int64_t to)
    {
        for (int64_t i = from; i < to; ++i) { t[i].kmer = EMPTY; t[i].last = NONE; t[i].next = NONE; }
    }

    // find_slot, _index_builder.pyx:313-342 (probe from the canonical k-mer's
    // home; stop at empty, the k-mer or its reverse complement)
    inline int32_t find(const KmerRef &k) const
    {
        const uint32_t mask = (uint32_t)(size - 1);
        uint32_t i = k.hash & mask;
        for (;;) {
            const uint64_t s = t[i].kmer;
            if (s == EMPTY || s == k.kmer || s == k.rc) return (int32_t)i;
            i = (i + 1) & mask;
        }
    }

    void expand()                                              // :204-224
    {
        const int64_t old = size;
        t = (Node *)realloc(t, sizeof(Node) * (size_t)(old << 1));
        size = old << 1;
        wipe(old, size);
        for (int64_t i = 0; i < old; ++i) {
            if (t[i].kmer == EMPTY) continue;
            KmerRef k;
            k.kmer = t[i].kmer;
            k.rc = revcomp(k.kmer);
            k.hash = sip(std::min(k.kmer, k.rc));
            const int32_t j = find(k);
            if (i != j) { t[j].kmer = k.kmer; t[i].kmer = EMPTY; }
        }
    }

    inline void add(const KmerRef &k)                          // _add_kmer, :181-198
    {
        int32_t i = find(k);
        if (t[i].kmer == EMPTY) {
            ++count;
            if ((double)count > 0.8 * (double)size) { expand(); i = find(k); }
            t[i].kmer = k.kmer;
        }
    }

    inline void link(int32_t i, int32_t j, bool fwd) { if (fwd) t[i].next = j; else t[i].last = j; }
    inline int32_t get(int32_t i, bool fwd) const { return fwd ? t[i].next : t[i].last; }
    inline void unlink(int32_t i, bool fwd)                    // :373-401
    {
        int32_t j;
        if (fwd) { j = t[i].next; t[i].next = NONE; } else { j = t[i].last; t[i].last = NONE; }
        if (j == NONE) return;
        if (t[j].next == i) t[j].next = NONE; else t[j].last = NONE;
    }

    inline void reg(const KmerRef &k)                          // _register_kmer, :256-307
    {
        const bool fwd = k.kmer <= k.rc;
        const uint64_t canon = fwd ? k.kmer : k.rc;
        const int32_t i = find(k);
        if (last == NONE) {
            if (t[i].kmer != EMPTY) { unlink(i, !fwd); last_new = false; }
            else { t[i].kmer = canon; last_new = true; }
            last_fwd = fwd; last = i;
            return;
        }
        if (t[i].kmer == EMPTY) {
            t[i].kmer = canon;
            if (last_new) { link(last, i, last_fwd); link(i, last, !fwd); }
            else unlink(last, last_fwd);
            last_new = true; last_fwd = fwd; last = i;
            return;
        }
        if (get(last, last_fwd) == i && get(i, !fwd) == last) {
            last_new = false; last_fwd = fwd; last = i;
            return;
        }
        unlink(last, last_fwd);
        unlink(i, !fwd);
        if (get(i, fwd) != NONE) { last_new = false; last_fwd = fwd; last = i; }
        else { last_new = true; last_fwd = true; last = NONE; }
    }
};

// Run `op(KmerRef, position)` over every k-mer of one transcript in order,
// a block at a time with the home slots prefetched first.
template <class Op>
inline void for_each_kmer(const Assembler &a, const char *seq, int64_t len, Op op)
{
    if (len < K) return;
    uint64_t kmer = 0;
    for (int i = 0; i < K - 1; ++i) kmer = (kmer << 2) | code(seq[i]);   // encode(seq, 0) >> 2
    KmerRef block[BLOCK];
    int64_t j = K - 1;
    while (j < len) {
        const int n = (int)std::min<int64_t>(BLOCK, len - j);
        const uint32_t mask = (uint32_t)(a.size - 1);
        for (int b = 0; b < n; ++b) {
            kmer = ((kmer << 2) | code(seq[j + b])) & MASK;
            block[b].kmer = kmer;
            block[b].rc = revcomp(kmer);
            block[b].hash = sip(std::min(block[b].kmer, block[b].rc));
            __builtin_prefetch(&a.t[block[b].hash & mask]);
        }
        for (int b = 0; b < n; ++b) op(block[b], j + b - K + 1);
        j += n;
    }
}

inline void decode(uint64_t kmer, char *out)                   // _kmer.pxd:113-143
{
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    for (int i = K - 1; i >= 0; --i) { out[i] = alphabet[kmer & 3]; kmer >>= 2; }
}

inline uint64_t encode(const char *s)
{
    uint64_t k = 0;
    for (int i = 0; i < K; ++i) k = (k << 2) | code(s[i]);
    return k;
}

thread_local std::string g_err;

}  // namespace

struct skm_built {
    std::vector<Node> kmers;          // final table, `last`/`next` now hold entry/offset
    std::vector<Contig> contigs;
    std::string sequences;
    std::vector<Target> targets;
};

extern "C" int skm_build_index(const char *pool, const int64_t *off, int64_t n_seqs, int n_threads,
                               skm_built **out)
{
    (void)n_threads;
    if (!pool || !off || !out || n_seqs <= 0) return SKM_ERR_ARG;
    Assembler a;
    a.size = 1024;                                              // _INITIAL_INDEX_SIZE
    a.t = (Node *)malloc(sizeof(Node) * (size_t)a.size);
    a.wipe(0, a.size);

    // pass 1 (_scan_kmers, :156-175) only decides the table size
    for (int64_t s = 0; s < n_seqs; ++s)
        for_each_kmer(a, pool + off[s], off[s + 1] - off[s],
                      [&](const KmerRef &k, int64_t) { a.add(k); });
    a.wipe(0, a.size);

    // pass 2 (_connect_kmers, :230-250): compacted de Bruijn graph
    for (int64_t s = 0; s < n_seqs; ++s) {
        const int64_t len = off[s + 1] - off[s];
        if (len < K) continue;
        a.last_new = true; a.last_fwd = true; a.last = NONE;
        for_each_kmer(a, pool + off[s], len, [&](const KmerRef &k, int64_t) { a.reg(k); });
        if (a.last != NONE) a.unlink(a.last, a.last_fwd);
    }

    // pass 3 (_assemble_contigs, :428-487): unitigs in slot order
    skm_built *b = new skm_built();
    std::vector<int64_t> lengths;
    char text[K];
    for (int64_t i = 0; i < a.size; ++i) {
        Node &n = a.t[i];
        if (n.kmer == EMPTY || n.last < 0) continue;
        if (n.last != NONE && n.next != NONE) continue;
        const int32_t contig = (int32_t)lengths.size();
        if (n.last == NONE && n.next == NONE) {
            n.last = ~contig; n.next = ~0;
            decode(n.kmer, text);
            b->sequences.append(text, K);
            lengths.push_back(K);
            continue;
        }
        const size_t start = b->sequences.size();
        int32_t prev = NONE, cur = (int32_t)i, offset = 0;
        while (cur != NONE) {
            if (cur < 0 || cur >= a.size) {                    // undefined behaviour in the reference
                delete b;
                g_err = "contig walk left the k-mer table";
                return SKM_ERR_UNDEFINED;
            }
            Node &c = a.t[cur];
            if (c.next == prev) { c.kmer = revcomp(c.kmer); std::swap(c.last, c.next); }
            prev = cur;
            cur = c.next;
            c.last = ~contig;
            c.next = ~offset;
            if (offset % K == 0) { decode(c.kmer, text); b->sequences.append(text, K); }
            ++offset;
        }
        decode(a.t[prev].kmer, text);
        const int tail = K - (offset - 1) % K;
        b->sequences.append(text + tail, K - tail);
        lengths.push_back((int64_t)(b->sequences.size() - start));
    }
    for (int64_t i = 0; i < a.size; ++i) { a.t[i].last = ~a.t[i].last; a.t[i].next = ~a.t[i].next; }

    // pass 4 (_map_contigs, :493-542): transcripts that touch a contig's first k-mer
    struct Rec { int32_t contig, entry, offset; };
    std::vector<Rec> recs;
    for (int64_t s = 0; s < n_seqs; ++s)
        for_each_kmer(a, pool + off[s], off[s + 1] - off[s], [&](const KmerRef &k, int64_t pos) {
            const Node &n = a.t[a.find(k)];
            if (n.next != 0) return;
            recs.push_back(Rec{n.last, k.kmer != n.kmer ? ~(int32_t)s : (int32_t)s, (int32_t)pos});
        });
    std::sort(recs.begin(), recs.end(), [](const Rec &x, const Rec &y) {
        if (x.contig != y.contig) return x.contig < y.contig;
        if (x.entry != y.entry) return x.entry < y.entry;
        return x.offset < y.offset;
    });

    // pass 5 (_compile_contigs, :544-572)
    const int64_t n_contigs = (int64_t)lengths.size();
    b->contigs.assign((size_t)n_contigs, Contig{0, 0, 0, 0, 0, 0});
    int64_t run = 0;
    for (int64_t c = 0; c < n_contigs; ++c) { b->contigs[c].offset = run; b->contigs[c].length = lengths[c]; run += lengths[c]; }
    for (const Rec &r : recs) {
        if (r.contig < 0 || r.contig >= n_contigs) { delete b; g_err = "target outside the contig table"; return SKM_ERR_UNDEFINED; }
        b->contigs[r.contig].target_count++;
    }
    run = 0;
    for (int64_t c = 0; c < n_contigs; ++c) {
        Contig &ct = b->contigs[c];
        ct.target_offset = run;
        run += ct.target_count;
        ct.first_kmer = encode(b->sequences.data() + ct.offset);
        ct.last_kmer = encode(b->sequences.data() + ct.offset + ct.length - K);
    }
    b->targets.resize(recs.size());
    for (size_t r = 0; r < recs.size(); ++r) b->targets[r] = Target{recs[r].entry, recs[r].offset};
    b->kmers.assign(a.t, a.t + a.size);
    *out = b;
    return SKM_OK;
}

extern "C" int skm_built_sizes(const skm_built *b, int64_t sizes[4])
{
    if (!b || !sizes) return SKM_ERR_ARG;
    sizes[0] = (int64_t)b->kmers.size();
    sizes[1] = (int64_t)b->contigs.size();
    sizes[2] = (int64_t)b->sequences.size();
    sizes[3] = (int64_t)b->targets.size();
    return SKM_OK;
}

extern "C" int skm_built_copy(const skm_built *b, void *kmers, void *contigs, char *sequences,
                              void *targets)
{
    if (!b) return SKM_ERR_ARG;
    if (kmers) memcpy(kmers, b->kmers.data(), b->kmers.size() * sizeof(Node));
    if (contigs) memcpy(contigs, b->contigs.data(), b->contigs.size() * sizeof(Contig));
    if (sequences) memcpy(sequences, b->sequences.data(), b->sequences.size());
    if (targets) memcpy(targets, b->targets.data(), b->targets.size() * sizeof(Target));
    return SKM_OK;
}

extern "C" int skm_built_free(skm_built *b)
{
    delete b;
    return SKM_OK;
}

extern "C" const char *skm_host_last_error(void) { return g_err.c_str(); }
