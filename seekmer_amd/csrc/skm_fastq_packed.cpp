// FASTQ files -> packed 2-bit pieces in one pass over the text (libseekmer_host.so):
// skm_fastq_packed_* and skm_pack_reads of include/seekmer_hip.h.
//
// The reference's feeders (/root/reference/seekmer/common.py:126-197) number the lines of a file
// and take line i & 3 == 0 as the name and line i & 3 == 1 as the bases; nothing else decides what
// a record is -- a quality line may begin with '@'.  A parallel reader therefore cannot know where
// records start in the middle of a file without having counted every newline before it.  The
// reader of skm_fastq.cpp counts them in a first pass over the whole text.  This one guesses and
// proves: every worker starts its range at the first place that looks like a record start and
// walks whole records from there (finding every newline, predicting none) until a record starts
// beyond its range; the consumer accepts a piece only if the piece before it -- proven by the
// same argument, the first one starting at byte 0 -- ended exactly where this one began, and walks
// the range again from the proven place otherwise.  One pass over the text instead of two, the
// first piece ready after one chunk's parsing time, and the same reads as the reference for any
// input.
#include "../../include/seekmer_hip.h"
#include "skm_pack_core.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace skmfq;

namespace {

std::atomic<int> g_forced_variant{-1};

int best_variant()
{
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx2")) return 2;
    if (__builtin_cpu_supports("ssse3")) return 1;
    return 0;
}

// -1 = the best this CPU has; otherwise the variant itself when the CPU has it, else -2
int resolve_variant(int want)
{
    const int best = best_variant();
    if (want < 0) {
        if (const char *v = getenv("SKM_PACK_VARIANT")) {
            const int asked = atoi(v);
            if (asked >= 0 && asked <= best) return asked;
        }
        return best;
    }
    return want <= best && want <= 2 ? want : -2;
}

walk_fn walk_of(int variant) { return variant == 2 ? walk_avx2 : variant == 1 ? walk_ssse3 : walk_scalar; }
span_fn span_of(int variant) { return variant == 2 ? span_avx2 : variant == 1 ? span_ssse3 : span_scalar; }

// Pieces outlive their reader (as the slabs of skm_fastq.cpp do): the next reader of the process
// starts with memory that is already touched -- and, with a page-locked allocator, already
// pinned.  The pool is bounded in bytes; what does not fit is freed.
std::mutex g_pool_lock;
std::vector<PackedOut *> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t POOL_MAX_BYTES = 1ull << 30;

PackedOut *take_piece(alloc_fn al, free_fn fr)
{
    PackedOut *s = nullptr;
    {
        std::lock_guard<std::mutex> hold(g_pool_lock);
        for (size_t i = g_pool.size(); i-- > 0;)
            if (g_pool[i]->codes.al == al) { s = g_pool[i]; g_pool.erase(g_pool.begin() + (long)i); break; }
        if (s) g_pool_bytes -= std::min(g_pool_bytes, s->bytes());
    }
    if (!s) s = new (std::nothrow) PackedOut();
    if (s) s->use(al, fr);
    return s;
}

void give_piece(PackedOut *s)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> hold(g_pool_lock);
        if (!s->failed() && g_pool_bytes + s->bytes() <= POOL_MAX_BYTES) {
            g_pool_bytes += s->bytes();
            g_pool.push_back(s);
            return;
        }
    }
    delete s;
}

inline size_t next_newline(const char *text, size_t n, size_t from)      // n when there is none
{
    if (from >= n) return n;
    const char *nl = (const char *)memchr(text + from, '\n', n - from);
    return nl ? (size_t)(nl - text) : n;
}

// does a record seem to start at `t` (a line start)?  '@' line, bases line, '+' line, a quality line as
// long as the bases line, then '@' again or the end.  Only a guess: the consumer proves or refutes it.
bool looks_like_record(const char *text, size_t n, size_t t)
{
    if (t >= n || text[t] != '@') return false;
    const size_t e0 = next_newline(text, n, t);
    if (e0 >= n) return false;
    const size_t l1 = e0 + 1, e1 = next_newline(text, n, l1);
    if (e1 >= n) return true;
    const size_t l2 = e1 + 1;
    if (l2 >= n) return true;
    if (text[l2] != '+') return false;
    const size_t e2 = next_newline(text, n, l2);
    if (e2 >= n) return true;
    const size_t l3 = e2 + 1, e3 = next_newline(text, n, l3);
    if (e3 - l3 != e1 - l1) return false;
    return e3 + 1 >= n || text[e3 + 1] == '@';
}

// the first line start in [a, b) that looks like a record start (a > 0); false when none is found
bool guess_start(const char *text, size_t n, size_t a, size_t b, size_t *start)
{
    size_t from = a - 1;
    for (int tries = 0; tries < 64; ++tries) {
        const size_t nl = next_newline(text, n, from);
        const size_t line = nl + 1;
        if (nl >= n || line >= b || line >= n) return false;
        if (looks_like_record(text, n, line)) { *start = line; return true; }
        from = line;
    }
    return false;
}

}  // namespace

struct skm_fastq_packed {
    std::vector<std::string> paths;
    bool paired = false, want_names = false;
    int n_threads = 0;
    size_t chunk_bytes = 8u << 20;
    alloc_fn al = malloc;
    free_fn fr = free;
    int variant = 0;
    walk_fn walk = walk_scalar;
    std::vector<Mapped> files;
    std::vector<size_t> file_begin, file_end;  // the bytes of every file this reader covers (a rank's share; else all)
    struct Item { int file, stream; size_t a, b; };          // file < 0: the end of a pair of files
    std::vector<Item> items;
    struct Result { PackedOut *out = nullptr; size_t start = 0, end = 0; bool has_start = false, broken = false; };
    std::mutex pm;
    std::condition_variable pcv;
    size_t next_claim = 0, next_deliver = 0;
    std::map<size_t, Result> ready;
    bool stop = false, started = false;
    std::vector<std::thread> workers;
    std::atomic<int> cw_hint{1};
    // consumer
    std::vector<size_t> file_pos;            // proven record boundary of every file
    std::vector<int64_t> file_reads;
    int64_t stream_next[2] = {0, 0};
    int64_t pair_base = 0;
    PackedOut *cur = nullptr, *prev = nullptr;      // the piece handed out last, and the one before it
    int64_t accepted = 0, reparsed = 0, n_reads = 0, n_exceptions = 0;
    bool failed = false;
    const bool trace = getenv("SKM_FASTQ_TRACE") != nullptr;      // tuning aid: per-piece timings on stderr

    Result parse(const Item &it)
    {
        Result r;
        const Mapped &f = files[(size_t)it.file];
        const auto t0 = std::chrono::steady_clock::now();
        r.out = take_piece(al, fr);
        if (!r.out) return r;
        const size_t before = r.out->bytes();
        r.out->start(cw_hint.load(std::memory_order_relaxed), want_names && it.stream == 0);
        if (it.a == file_begin[(size_t)it.file]) { r.has_start = true; r.start = it.a; }   // (proven: byte 0, or a share's first record)
        else r.has_start = guess_start(f.p, f.n, it.a, it.b, &r.start);
        if (r.has_start) r.end = walk(f.p, f.n, r.start, it.b, *r.out);
        if (trace)
            fprintf(stderr, "[skm_fastq_packed] file %d [%zu, %zu): %lld reads in %.2f ms (arrays %zu -> %zu bytes)\n", it.file,
                    it.a, it.b, (long long)r.out->n_reads,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), before, r.out->bytes());
        return r;
    }

    void worker_main()
    {
        for (;;) {
            size_t k;
            {
                std::unique_lock<std::mutex> hold(pm);
                pcv.wait(hold, [&] {
                    return stop || next_claim >= items.size() || next_claim < next_deliver + (size_t)n_threads + 4;
                });
                if (stop || next_claim >= items.size()) return;
                k = next_claim++;
            }
            Result r;
            try {                              // (an allocation failure fails the reader, not the process)
                if (items[k].file >= 0) r = parse(items[k]);
            } catch (const std::bad_alloc &) {
                give_piece(r.out);
                r = Result();
                r.broken = true;
            }
            {
                std::lock_guard<std::mutex> hold(pm);
                ready[k] = r;
            }
            pcv.notify_all();
        }
    }

    ~skm_fastq_packed()
    {
        {
            std::lock_guard<std::mutex> hold(pm);
            stop = true;
        }
        pcv.notify_all();
        for (auto &t : workers) t.join();
        for (auto &kv : ready) give_piece(kv.second.out);
        give_piece(cur);
        give_piece(prev);
        for (auto &f : files) f.unmap();
    }
};

extern "C" int skm_fastq_cache_bytes(int64_t bytes)
{
    if (bytes < 0) return SKM_ERR_ARG;
    set_map_keep_bytes((size_t)bytes);
    return SKM_OK;
}

extern "C" int skm_pack_set_variant(int variant)
{
    if (variant >= 0 && resolve_variant(variant) < 0) return SKM_ERR_STATE;
    g_forced_variant.store(variant < 0 ? -1 : variant);
    return SKM_OK;
}

namespace {
int open_reader(const char *const *paths, int n_paths, int paired, int n_threads, int64_t chunk_bytes, int want_names,
                const int64_t *begin, const int64_t *end, int64_t first_unit, skm_fastq_packed **out);
}

extern "C" int skm_fastq_packed_open(const char *const *paths, int n_paths, int paired, int n_threads,
                                     int64_t chunk_bytes, int want_names, skm_fastq_packed **out)
{
    return open_reader(paths, n_paths, paired, n_threads, chunk_bytes, want_names, nullptr, nullptr, 0, out);
}

extern "C" int skm_fastq_packed_open_ranges(const char *const *paths, int n_paths, int paired, int n_threads,
                                            int64_t chunk_bytes, int want_names, const int64_t *begin,
                                            const int64_t *end, int64_t first_unit, skm_fastq_packed **out)
{
    if (!begin || !end || first_unit < 0) return SKM_ERR_ARG;
    return open_reader(paths, n_paths, paired, n_threads, chunk_bytes, want_names, begin, end, first_unit, out);
}

namespace {
int open_reader(const char *const *paths, int n_paths, int paired, int n_threads, int64_t chunk_bytes, int want_names,
                const int64_t *begin, const int64_t *end, int64_t first_unit, skm_fastq_packed **out)
{
    if (!paths || n_paths <= 0 || !out || n_threads < 0 || chunk_bytes < 0) return SKM_ERR_ARG;
    if (paired && (n_paths % 2) != 0) return SKM_ERR_ARG;      // common.py:178-179 raises ValueError
    skm_fastq_packed *q = new (std::nothrow) skm_fastq_packed();
    if (!q) return SKM_ERR_STATE;
    for (int i = 0; i < n_paths; ++i) q->paths.emplace_back(paths[i]);
    q->paired = paired != 0;
    q->want_names = want_names != 0;
    q->n_threads = n_threads;
    if (chunk_bytes) q->chunk_bytes = std::max<size_t>((size_t)chunk_bytes, 64);
    q->variant = resolve_variant(g_forced_variant.load());
    if (q->variant < 0) q->variant = best_variant();
    q->walk = walk_of(q->variant);
    q->files.resize(q->paths.size());
    for (size_t i = 0; i < q->paths.size(); ++i)
        if (!q->files[i].map(q->paths[i].c_str())) { delete q; return SKM_ERR_IO; }
    // the bytes to read: whole files, or the share the caller has located (record boundaries both,
    // skm_fastq_locate_line) -- a walk that starts at a share's first byte is proven as one from byte 0 is
    q->file_begin.assign(q->files.size(), 0);
    q->file_end.resize(q->files.size());
    for (size_t i = 0; i < q->files.size(); ++i) {
        q->file_end[i] = q->files[i].n;
        if (begin) {
            if (begin[i] < 0 || end[i] < begin[i] || (size_t)end[i] > q->files[i].n) { delete q; return SKM_ERR_ARG; }
            q->file_begin[i] = (size_t)begin[i];
            q->file_end[i] = (size_t)end[i];
        }
    }
    q->file_pos = q->file_begin;
    q->file_reads.assign(q->files.size(), 0);
    q->pair_base = first_unit;
    q->stream_next[0] = q->stream_next[1] = first_unit;
    const size_t step = q->paired ? 2 : 1;
    for (size_t f = 0; f + step <= q->files.size(); f += step) {
        size_t chunks[2] = {0, 0};
        for (size_t s = 0; s < step; ++s)
            chunks[s] = (q->file_end[f + s] - q->file_begin[f + s] + q->chunk_bytes - 1) / q->chunk_bytes;
        for (size_t k = 0; k < std::max(chunks[0], chunks[1]); ++k)
            for (size_t s = 0; s < step; ++s)
                if (k < chunks[s]) {
                    const size_t a = q->file_begin[f + s], b = q->file_end[f + s];
                    q->items.push_back({(int)(f + s), (int)s, a + k * q->chunk_bytes, std::min(b, a + (k + 1) * q->chunk_bytes)});
                }
        q->items.push_back({-1, (int)f, 0, 0});
    }
    *out = q;
    return SKM_OK;
}
}  // namespace

// ---- a sample shared out over ranks ------------------------------------------------------------
// The reference numbers the LINES of a file (seekmer/common.py:126-197): record u is lines 4u .. 4u + 3
// whatever they hold.  A rank that reads units [lo, hi) of a sample must therefore know where line
// 4 lo starts, which takes the number of newlines before it: every rank counts the newlines of its
// share of fixed-size chunks (skm_fastq_count_newlines; the counts are summed over the ranks by the
// caller), then finds the chunk that holds the line it wants and walks that one chunk
// (skm_fastq_locate_line).  memchr over 1/N of the text and over one chunk: the text is parsed once,
// by the rank that owns it.
extern "C" int skm_fastq_count_newlines(const char *path, int64_t chunk_bytes, int64_t first_chunk, int64_t chunk_step,
                                        int n_threads, int64_t *counts, int64_t n_chunks, int *last_line_open)
{
    if (!path || chunk_bytes <= 0 || first_chunk < 0 || chunk_step <= 0 || !counts || n_chunks < 0) return SKM_ERR_ARG;
    Mapped f;
    if (!f.map(path)) return SKM_ERR_IO;
    const int64_t have = (int64_t)((f.n + (size_t)chunk_bytes - 1) / (size_t)chunk_bytes);
    if (have != n_chunks) { f.unmap(); return SKM_ERR_ARG; }
    if (last_line_open) *last_line_open = f.n > 0 && f.p[f.n - 1] != '\n';
    std::atomic<int64_t> next{first_chunk};
    auto work = [&]() {
        for (;;) {
            const int64_t k = next.fetch_add(chunk_step);
            if (k >= n_chunks) return;
            const size_t a = (size_t)k * (size_t)chunk_bytes, b = std::min(f.n, a + (size_t)chunk_bytes);
            int64_t c = 0;
            for (size_t at = a; at < b;) {
                const char *nl = (const char *)memchr(f.p + at, '\n', b - at);
                if (!nl) break;
                ++c;
                at = (size_t)(nl - f.p) + 1;
            }
            counts[k] = c;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < std::max(1, n_threads); ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    f.unmap();
    return SKM_OK;
}

// byte at which line `line` (numbered from 0) starts, from the newline counts of ALL chunks; the file's
// size when the file has fewer lines
extern "C" int skm_fastq_locate_line(const char *path, int64_t chunk_bytes, const int64_t *counts, int64_t n_chunks,
                                     int64_t line, int64_t *byte_offset)
{
    if (!path || chunk_bytes <= 0 || !counts || n_chunks < 0 || line < 0 || !byte_offset) return SKM_ERR_ARG;
    if (line == 0) { *byte_offset = 0; return SKM_OK; }
    Mapped f;
    if (!f.map(path)) return SKM_ERR_IO;
    auto done = [&](int rc) { f.unmap(); return rc; };
    if ((int64_t)((f.n + (size_t)chunk_bytes - 1) / (size_t)chunk_bytes) != n_chunks) return done(SKM_ERR_ARG);
    int64_t before = 0, k = 0;                     // line `line` starts behind newline number `line` (counted from 1)
    while (k < n_chunks && before + counts[k] < line) before += counts[k++];
    if (k >= n_chunks) { *byte_offset = (int64_t)f.n; return done(SKM_OK); }
    size_t at = (size_t)k * (size_t)chunk_bytes;
    const size_t b = std::min(f.n, at + (size_t)chunk_bytes);
    for (int64_t seen = before; at < b;) {
        const char *nl = (const char *)memchr(f.p + at, '\n', b - at);
        if (!nl) break;
        at = (size_t)(nl - f.p) + 1;
        if (++seen == line) { *byte_offset = (int64_t)at; return done(SKM_OK); }
    }
    return done(SKM_ERR_STATE);                    // the counts are not this file's
}

extern "C" int skm_fastq_packed_set_allocator(skm_fastq_packed *q, void *(*alloc)(size_t), void (*release)(void *))
{
    if (!q || !alloc || !release) return SKM_ERR_ARG;
    if (q->started) return SKM_ERR_STATE;
    q->al = alloc;
    q->fr = release;
    return SKM_OK;
}

extern "C" int skm_fastq_packed_next(void *reader, skm_packed_reads *piece)
{
    skm_fastq_packed *q = (skm_fastq_packed *)reader;
    if (!q || !piece) return SKM_ERR_ARG;
    memset(piece, 0, sizeof(*piece));
    piece->uniform_len = -1;
    if (q->failed) return SKM_ERR_IO;
    if (!q->started) {
        q->started = true;
        for (int t = 0; t < q->n_threads; ++t) q->workers.emplace_back([q]() { q->worker_main(); });
    }
    // a piece's arrays stay valid through ONE more call: whoever copies them asynchronously (the
    // mapper's drain) asks for the next piece while that copy runs
    give_piece(q->prev);
    q->prev = q->cur;
    q->cur = nullptr;
    for (;;) {
        if (q->next_deliver >= q->items.size()) return SKM_OK;       // n_reads == 0: the end
        const skm_fastq_packed::Item it = q->items[q->next_deliver];
        if (it.file < 0) {
            // zip(file1, file2) stops at the shorter file: the pair of files counts min(records) units,
            // the next pair (or, single-ended, the next file) continues from there
            const size_t f = (size_t)it.stream;
            int64_t units = q->file_reads[f];
            if (q->paired) units = std::min(units, q->file_reads[f + 1]);
            // a stream that got more reads than that (the longer file) is cut back before anything
            // of the next pair of files is handed out: its leftover reads must never meet the next
            // file's reads of the other stream as their mates
            int cut = -1;
            for (int s = 0; q->paired && s < 2; ++s)
                if (q->file_reads[f + (size_t)s] > units) cut = s;
            q->pair_base += units;
            q->stream_next[0] = q->stream_next[1] = q->pair_base;
            if (q->n_threads > 0) {
                std::unique_lock<std::mutex> hold(q->pm);
                q->pcv.wait(hold, [&] { return q->ready.count(q->next_deliver) != 0; });
                q->ready.erase(q->next_deliver);
                q->next_deliver++;
                hold.unlock();
                q->pcv.notify_all();
            } else q->next_deliver++;
            if (cut >= 0) {
                piece->stream = cut;
                piece->code_words = SKM_PACKED_CUT;
                piece->first_read = q->pair_base;
                return SKM_OK;
            }
            continue;
        }
        skm_fastq_packed::Result r;
        if (q->n_threads > 0) {
            std::unique_lock<std::mutex> hold(q->pm);
            q->pcv.wait(hold, [&] { return q->ready.count(q->next_deliver) != 0; });
            r = q->ready[q->next_deliver];
            q->ready.erase(q->next_deliver);
            q->next_deliver++;
            hold.unlock();
            q->pcv.notify_all();
        } else {
            const size_t pos = q->file_pos[(size_t)it.file];
            if (pos >= it.b && it.b > 0) { q->next_deliver++; continue; }    // (covered: nothing to parse)
            r = q->parse(it);
            q->next_deliver++;
        }
        if (!r.out || r.broken) { q->failed = true; return SKM_ERR_STATE; }
        const Mapped &f = q->files[(size_t)it.file];
        size_t &pos = q->file_pos[(size_t)it.file];
        if (pos >= it.b) {                       // the walk before this range ran past it: nothing starts here
            give_piece(r.out);
            continue;
        }
        if (r.has_start && r.start == pos) q->accepted++;
        else {                                   // the guess is refuted (or there was none): walk from the proven place
            r.out->start(q->cw_hint.load(std::memory_order_relaxed), q->want_names && it.stream == 0);
            r.end = q->walk(f.p, f.n, pos, it.b, *r.out);
            q->reparsed++;
        }
        if (r.out->failed()) { give_piece(r.out); q->failed = true; return SKM_ERR_STATE; }
        pos = r.end;
        if (r.out->cw > q->cw_hint.load(std::memory_order_relaxed)) q->cw_hint.store(r.out->cw, std::memory_order_relaxed);
        if (r.out->n_reads == 0) { give_piece(r.out); continue; }
        PackedOut *o = r.out;
        q->cur = o;
        q->file_reads[(size_t)it.file] += o->n_reads;
        q->n_reads += o->n_reads;
        q->n_exceptions += (int64_t)o->exc_reads.n;
        piece->stream = it.stream;
        piece->code_words = o->cw;
        piece->first_read = q->stream_next[it.stream];
        q->stream_next[it.stream] += o->n_reads;
        piece->n_reads = o->n_reads;
        piece->read_stride = o->cw;
        piece->uniform_len = o->uniform_len >= 0 ? o->uniform_len : -1;
        piece->codes = o->codes.p;
        piece->lengths = o->lengths.p;
        piece->n_exceptions = (int64_t)o->exc_reads.n;
        piece->exception_reads = o->exc_reads.p;
        piece->exception_masks = o->exc_masks.p;
        if (o->want_names) {
            if (o->names.n == 0) o->names.reserve(1);
            piece->names = o->names.p;
            piece->name_offsets = o->name_offsets.p;
        }
        return SKM_OK;
    }
}

extern "C" int skm_fastq_packed_stats(const skm_fastq_packed *q, int64_t stats[8])
{
    if (!q || !stats) return SKM_ERR_ARG;
    for (int i = 0; i < 8; ++i) stats[i] = 0;
    stats[0] = q->accepted;
    stats[1] = q->reparsed;
    stats[2] = q->n_reads;
    stats[3] = q->n_exceptions;
    stats[4] = q->variant;
    // units so far: whole pairs of files, plus what the current pair has in both streams
    stats[5] = q->paired ? std::min(q->stream_next[0], q->stream_next[1]) : q->stream_next[0];
    return SKM_OK;
}

// About how many units the files hold: each mate-1 (or single-end) file's size over the extent of its
// first record.  A hint for whoever sizes tables before the reads arrive (skm_mapper_expect_units).
extern "C" int skm_fastq_packed_estimate(const skm_fastq_packed *q, int64_t *units)
{
    if (!q || !units) return SKM_ERR_ARG;
    int64_t total = 0;
    const size_t step = q->paired ? 2 : 1;
    for (size_t f = 0; f < q->files.size(); f += step) {
        const Mapped &m = q->files[f];
        const size_t from = q->file_begin[f], to = q->file_end[f];
        size_t at = from;
        for (int line = 0; line < 4 && at < to; ++line) at = next_newline(m.p, m.n, at) + 1;
        if (at > to) at = to;
        if (at > from) total += (int64_t)((to - from + (at - from) - 1) / (at - from));
    }
    *units = total;
    return SKM_OK;
}

extern "C" int skm_fastq_packed_close(skm_fastq_packed *q)
{
    delete q;
    return SKM_OK;
}

extern "C" int skm_pack_reads(const char *bases, const int64_t *offsets, int64_t n_reads, int32_t code_words,
                              uint64_t *codes, uint32_t *lengths, uint32_t *exception_reads,
                              uint32_t *exception_masks, int64_t cap_exceptions, int64_t *n_exceptions, int variant)
{
    if (!offsets || n_reads < 0 || code_words < 1 || !codes || !lengths || cap_exceptions < 0 || !n_exceptions)
        return SKM_ERR_ARG;
    if (n_reads > 0 && !bases) return SKM_ERR_ARG;
    if (cap_exceptions > 0 && (!exception_reads || !exception_masks)) return SKM_ERR_ARG;
    const int v = resolve_variant(variant);
    if (v < 0) return SKM_ERR_STATE;
    const span_fn span = span_of(v);
    *n_exceptions = 0;
    // in blocks of reads through a scratch piece (the exception arrays are bounded by the caller)
    PackedOut out;
    constexpr int64_t BLOCK_READS = 4096;
    for (int64_t first = 0; first < n_reads; first += BLOCK_READS) {
        const int64_t last = std::min(n_reads, first + BLOCK_READS);
        out.start(code_words, false);
        for (int64_t r = first; r < last; ++r) {
            const int64_t len = offsets[r + 1] - offsets[r];
            if (len < 0 || len > (int64_t)code_words * 32) return SKM_ERR_ARG;
            span(bases + offsets[r], (size_t)len, out);
        }
        if (out.failed() || out.cw != code_words) return SKM_ERR_STATE;
        memcpy(codes + first * code_words, out.codes.p, (size_t)(last - first) * code_words * 8);
        memcpy(lengths + first, out.lengths.p, (size_t)(last - first) * 4);
        for (size_t e = 0; e < out.exc_reads.n; ++e) {
            if (*n_exceptions < cap_exceptions) {
                exception_reads[*n_exceptions] = (uint32_t)(first + out.exc_reads.p[e]);
                memcpy(exception_masks + *n_exceptions * code_words, out.exc_masks.p + e * code_words, (size_t)code_words * 4);
            }
            ++*n_exceptions;
        }
    }
    return *n_exceptions > cap_exceptions ? SKM_ERR_STATE : SKM_OK;
}
