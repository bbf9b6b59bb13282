// Native FASTQ batch packer (libseekmer_host.so): the reference's Python
// feeders (/root/reference/seekmer/common.py:126-197) hold the GIL and top out
// near 0.7 M pairs/s; this reader yields the same batches -- line i&3==0 ->
// name = strip()[1:], line i&3==1 -> bases = strip(), case preserved, the '+'
// and quality lines never validated, one buffer carried across files, flush
// when batch_units names are held or at the very end -- as flat arrays
// (bases back to back + offsets) ready for skm_mapper_map_batch.
//
// Two engines behind one interface:
//   * sequential -- any input (pipes from zcat/bzcat/xzcat included), one thread;
//   * parallel   -- plain files whose line counts are multiples of four: the files are
//     memory-mapped, a first pass counts the newlines of every 1 MiB block (all threads), and
//     from then on batch k = units [k * batch_units, ...) is a known range of lines of each
//     file: worker threads parse whole batches side by side into slabs of their own and the
//     reader hands them out in file order.  The reference's rule is purely line-number based
//     (`i & 3`), which is what makes the split exact: a '@' at the start of a quality line
//     cannot shift anything.
// Slabs come from an allocator the caller may replace (skm_fastq_set_allocator): with
// page-locked memory (skm_pinned_alloc) a batch goes over PCIe by DMA straight from the slab.
#include "../../include/seekmer_hip.h"
#include "skm_fastq_shared.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace skmfq;

namespace {

struct LineReader {
    FILE *f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;

    bool open(const char *path)
    {
        f = fopen(path, "rb");
        buf.resize(1 << 22);
        pos = end = 0;
        eof = false;
        return f != nullptr;
    }
    void close() { if (f) fclose(f); f = nullptr; }

    // next line without its terminator; false at end of file
    bool next(const char **line, size_t *len, std::string &spill)
    {
        spill.clear();
        for (;;) {
            if (pos < end) {
                char *nl = (char *)memchr(buf.data() + pos, '\n', end - pos);
                if (nl) {
                    const size_t n = (size_t)(nl - (buf.data() + pos));
                    if (spill.empty()) { *line = buf.data() + pos; *len = n; }
                    else { spill.append(buf.data() + pos, n); *line = spill.data(); *len = spill.size(); }
                    pos += n + 1;
                    return true;
                }
                spill.append(buf.data() + pos, end - pos);
                pos = end;
            }
            if (eof) {
                if (spill.empty()) return false;
                *line = spill.data();       // last line without a terminator
                *len = spill.size();
                return true;
            }
            end = fread(buf.data(), 1, buf.size(), f);
            pos = 0;
            if (end == 0) eof = true;
        }
    }
};

inline bool is_space(char c)        // bytes.strip() with no argument
{
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f';
}

inline void strip(const char *&p, size_t &n)
{
    while (n && is_space(p[0])) { ++p; --n; }
    while (n && is_space(p[n - 1])) --n;
}

}  // namespace

// The four arrays of one batch.  Slabs are recycled: touching fresh pages is
// what a large batch costs most (measured here: 0.27 GB/s of first-touch
// against 3 GB/s of parsing), so a reader keeps the slabs it has grown.
struct skm_fastq_slab {
    Buf<char> bases, names;
    Buf<int64_t> offsets, name_offsets;
    int64_t units = 0;
    int64_t read_len = -2;                     // -2 no read yet, >= 0 every read so far this long, -1 ragged
    void add_read(const char *p, size_t len)
    {
        bases.append(p, len);
        offsets.push((int64_t)bases.n);
        read_len = read_len == -2 ? (int64_t)len : (read_len == (int64_t)len ? read_len : -1);
    }
    void use(alloc_fn al, free_fn fr)
    {
        if (bases.al == al && bases.fr == fr) return;
        bases.release(); names.release(); offsets.release(); name_offsets.release();
        bases.al = al; names.al = al; bases.fr = fr; names.fr = fr;
        offsets.al = al; offsets.fr = fr; name_offsets.al = al; name_offsets.fr = fr;
    }
    void start()
    {
        bases.clear(); names.clear(); offsets.clear(); name_offsets.clear();
        offsets.push(0); name_offsets.push(0);
        units = 0;
        read_len = -2;
    }
    bool failed() const { return bases.failed || names.failed || offsets.failed || name_offsets.failed; }
    ~skm_fastq_slab() { bases.release(); names.release(); offsets.release(); name_offsets.release(); }
};

namespace {

// Slabs outlive their reader: the next reader of the process (the next sample, the next
// benchmark pass) starts with memory that is already touched -- and, with a page-locked
// allocator, already pinned (pinning costs far more than parsing).
std::mutex g_spare_lock;
std::vector<skm_fastq_slab *> g_spare;
constexpr size_t MAX_SPARE = 32;

}  // namespace

// ---- the process-wide cache of file mappings (skm_fastq_shared.h) ------------------------
// Setting up and tearing down the page tables of a multi-gigabyte text file costs more than
// parsing it (4.4 GB: ~50 ms of faults, ~100 ms of munmap with the process's memory-map lock
// held -- the quantification that follows a mapping run stalled on exactly that), so a mapping
// is shared by the readers that have the same file open at the same time and torn down in the
// background and in pieces.  A caller that reads the same files again and again (a benchmark's
// passes, a service) may let unused mappings stay cached up to a budget
// (skm_fastq_cache_bytes); the default keeps nothing: a file replaced or truncated in place
// must not be read through a stale mapping by the next sample.
namespace skmfq {
namespace {
struct MapEntry {
    std::string path;
    dev_t dev; ino_t ino; off_t size; long mtime_s, mtime_ns;
    const char *p;
    int users;
    uint64_t stamp;
};
std::mutex g_map_lock;
std::vector<MapEntry> g_maps;
uint64_t g_map_stamp = 0;
size_t g_map_keep_bytes = 0;

void unmap_in_pieces(const char *p, size_t n)
{
    // The memory-map lock is released between pieces -- and left alone for a moment: a thread that
    // needs it (hipHostMalloc, hipMalloc, hipStreamCreate of the quantification that follows a mapping
    // run) got it only after the whole teardown when the pieces were 32 MB with a yield in between
    // (measured: the first such call of the next phase took 115-137 ms behind 4.4 GB of text).
    constexpr size_t PIECE = 4u << 20;
    for (size_t at = 0; at < n; at += PIECE) {
        munmap((void *)(p + at), std::min(PIECE, n - at));
        std::this_thread::sleep_for(std::chrono::microseconds(40));
    }
}

// (g_map_lock held) unused mappings beyond the budget, oldest first
void collect_droppable(std::vector<std::pair<const char *, size_t>> &drop)
{
    size_t kept = 0;
    for (auto &e : g_maps) if (e.users == 0) kept += (size_t)e.size;
    while (kept > g_map_keep_bytes) {
        int oldest = -1;
        for (size_t i = 0; i < g_maps.size(); ++i)
            if (g_maps[i].users == 0 && (oldest < 0 || g_maps[i].stamp < g_maps[(size_t)oldest].stamp)) oldest = (int)i;
        if (oldest < 0) break;
        drop.emplace_back(g_maps[(size_t)oldest].p, (size_t)g_maps[(size_t)oldest].size);
        kept -= (size_t)g_maps[(size_t)oldest].size;
        g_maps.erase(g_maps.begin() + oldest);
    }
}

std::atomic<int> g_unmapping{0};      // background teardowns under way (skm_fastq_wait_unmapped)

void drop_in_background(std::vector<std::pair<const char *, size_t>> drop)
{
    if (drop.empty()) return;
    g_unmapping.fetch_add(1);
    std::thread([drop]() {
        for (auto &d : drop) unmap_in_pieces(d.first, d.second);
        g_unmapping.fetch_sub(1);
    }).detach();
}
}  // namespace

// Mappings that nobody uses any more are torn down by a background thread, a few MB at a time
// (above).  Whoever is about to time something that faults pages or allocates (both queue behind
// the teardown's hold on the process's memory-map lock) waits here first.
extern "C" int skm_fastq_wait_unmapped(void)
{
    while (g_unmapping.load() > 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    return SKM_OK;
}

void set_map_keep_bytes(size_t bytes)
{
    std::vector<std::pair<const char *, size_t>> drop;
    {
        std::lock_guard<std::mutex> hold(g_map_lock);
        g_map_keep_bytes = bytes;
        collect_droppable(drop);
    }
    drop_in_background(drop);
}

bool Mapped::map(const char *path)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return false; }
    n = (size_t)st.st_size;
    if (n) {
        std::lock_guard<std::mutex> hold(g_map_lock);
        for (auto &e : g_maps)
            if (e.dev == st.st_dev && e.ino == st.st_ino && e.size == st.st_size
                    && e.mtime_s == (long)st.st_mtim.tv_sec && e.mtime_ns == (long)st.st_mtim.tv_nsec) {
                e.users++;
                e.stamp = ++g_map_stamp;
                p = e.p;
                close(fd);
                return true;
            }
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); return false; }
        (void)madvise(m, n, MADV_WILLNEED);
        p = (const char *)m;
        g_maps.push_back(MapEntry{path, st.st_dev, st.st_ino, st.st_size, (long)st.st_mtim.tv_sec,
                                  (long)st.st_mtim.tv_nsec, p, 1, ++g_map_stamp});
    }
    close(fd);
    return true;
}

void Mapped::unmap()
{
    if (!p) return;
    std::vector<std::pair<const char *, size_t>> drop;
    {
        std::lock_guard<std::mutex> hold(g_map_lock);
        for (auto &e : g_maps)
            if (e.p == p && e.users > 0) { e.users--; break; }
        collect_droppable(drop);
    }
    drop_in_background(drop);
    p = nullptr;
    n = 0;
}

}  // namespace skmfq

// ---- page tables of the input ahead of its reader ------------------------------------------
// A run knows its FASTQ files long before it can parse them (the index takes a second or more to
// load and upload).  Mapping the files then and touching their pages from a few helper threads
// takes the page-table set-up of the text -- as much work as parsing it: 4.4 GB, 65 ms cold against
// 29 ms warm on 14 threads -- out of the reader's way; the reader finds the mappings in the cache.
struct skm_fastq_prefault {
    std::vector<skmfq::Mapped> files;
    std::vector<std::thread> workers;
    std::atomic<bool> stop{false};
};

extern "C" int skm_fastq_prefault_start(const char *const *paths, int n_paths, int n_threads, skm_fastq_prefault **out)
{
    if (!paths || n_paths <= 0 || n_threads < 1 || !out) return SKM_ERR_ARG;
    skm_fastq_prefault *h = new (std::nothrow) skm_fastq_prefault();
    if (!h) return SKM_ERR_STATE;
    h->files.resize((size_t)n_paths);
    for (int i = 0; i < n_paths; ++i)
        if (!h->files[(size_t)i].map(paths[i])) {
            for (auto &f : h->files) f.unmap();
            delete h;
            return SKM_ERR_IO;
        }
    // ranges of 8 MiB dealt round-robin over the threads, all files in the order a reader takes them
    constexpr size_t RANGE = 8u << 20;
    for (int t = 0; t < n_threads; ++t)
        h->workers.emplace_back([h, t, n_threads]() {
            size_t k = 0;
            unsigned sink = 0;
            for (const auto &f : h->files)
                for (size_t at = 0; at < f.n; at += RANGE, ++k) {
                    if ((int)(k % (size_t)n_threads) != t) continue;
                    const size_t end = std::min(f.n, at + RANGE);
                    for (size_t b = at; b < end && !h->stop.load(std::memory_order_relaxed); b += 4096) sink += (unsigned char)f.p[b];
                }
            if (sink == 0xffffffffu) h->stop.store(true);     // (keeps the loads)
        });
    *out = h;
    return SKM_OK;
}

// stops the helpers and gives the mappings back to the cache (a reader that has the files open keeps them)
extern "C" int skm_fastq_prefault_finish(skm_fastq_prefault *h)
{
    if (!h) return SKM_OK;
    h->stop.store(true);
    for (auto &t : h->workers) t.join();
    for (auto &f : h->files) f.unmap();
    delete h;
    return SKM_OK;
}

namespace skmfq {

size_t Mapped::line_start(int64_t line) const
{
    if (line <= 0) return 0;
    if (line >= lines) return n;
    // the last block with fewer than `line` newlines before it
    size_t b = (size_t)(std::lower_bound(before.begin(), before.end(), line) - before.begin()) - 1;
    int64_t seen = before[b];
    size_t at = b * BLOCK;
    while (seen < line) {
        const char *nl = (const char *)memchr(p + at, '\n', n - at);
        if (!nl) return n;
        at = (size_t)(nl - p) + 1;
        ++seen;
    }
    return at;
}

}  // namespace skmfq

namespace {

// (a byte-compare-and-add loop: the compiler turns it into 16/32-byte vector compares, an
// order of magnitude faster than one memchr call per 100-byte line)
inline int64_t count_newlines(const char *p, size_t n)
{
    int64_t total = 0;
    size_t i = 0;
    while (i < n) {
        const size_t stop = std::min(n, i + 4096);
        unsigned int c = 0;
        for (; i < stop; ++i) c += p[i] == '\n';
        total += c;
    }
    return total;
}

}  // namespace

extern "C" int skm_fastq_recycle(skm_fastq *q, skm_fastq_slab *slab);

struct skm_fastq {
    std::vector<std::string> paths;
    bool paired = false;
    int64_t batch_units = 65536;
    alloc_fn al = malloc;
    free_fn fr = free;
    int shard_rank = 0, shard_world = 1;      // this reader hands out batches k with k % world == rank
    int64_t seq_batch = 0;                    // sequential engine: index of the next batch it parses
    int64_t last_index = -1;                  // index (in the whole sample) of the batch handed out last
    int64_t last_read_len = -1;               // its reads' common length, or -1
    // ---- sequential engine
    size_t next_path = 0;
    LineReader r1, r2;
    bool open = false;
    int64_t line_no = 0;
    bool finished = false;
    std::string s1, s2;
    // current batch
    skm_fastq_slab *cur = nullptr;
    int64_t held = 0;
    // ---- parallel engine
    int n_threads = 0;                        // > 0: parallel
    std::vector<Mapped> files;                // one per path
    std::vector<int64_t> pair_first_unit;     // first global unit of every file (pair), then the total
    int64_t total_units = 0, n_batches = 0;
    std::vector<std::thread> workers;
    std::mutex pm;
    std::condition_variable pcv;
    int64_t next_claim = 0, next_deliver = 0;
    std::map<int64_t, skm_fastq_slab *> ready;
    bool stop = false, failed = false;

    skm_fastq_slab *take_slab()
    {
        skm_fastq_slab *s = nullptr;
        {
            std::lock_guard<std::mutex> hold(g_spare_lock);
            // (prefer one that already lives in this reader's kind of memory)
            for (size_t i = g_spare.size(); i-- > 0;)
                if (g_spare[i]->bases.al == al) { s = g_spare[i]; g_spare.erase(g_spare.begin() + (long)i); break; }
            if (!s && !g_spare.empty()) { s = g_spare.back(); g_spare.pop_back(); }
        }
        if (!s) s = new skm_fastq_slab();
        s->use(al, fr);
        s->start();
        return s;
    }
    void reset_batch()
    {
        if (cur) { cur->use(al, fr); cur->start(); }
        else cur = take_slab();
        held = 0;
    }
    void stop_workers()
    {
        {
            std::lock_guard<std::mutex> hold(pm);
            stop = true;
        }
        pcv.notify_all();
        for (auto &t : workers) t.join();
        workers.clear();
    }
    ~skm_fastq()
    {
        stop_workers();
        for (auto &kv : ready) delete kv.second;
        for (auto &f : files) f.unmap();               // (back to the process-wide cache of mappings)
        if (cur) skm_fastq_recycle(nullptr, cur);      // (grown, touched, maybe page-locked: worth keeping)
        cur = nullptr;
    }

    // ---- parallel engine --------------------------------------------------------------
    // units [u0, u1) of the file (pair) starting at path index `first`, appended to `out`
    void parse_units(size_t first, int64_t u0, int64_t u1, skm_fastq_slab *out)
    {
        const Mapped &a = files[first];
        const Mapped *b = paired ? &files[first + 1] : nullptr;
        size_t pa = a.line_start(4 * u0), pb = b ? b->line_start(4 * u0) : 0;
        const size_t ea = a.line_start(4 * u1), eb = b ? b->line_start(4 * u1) : 0;
        // a sequence line is never longer than its quality line: bases <= half the text
        out->bases.reserve(out->bases.n + (ea - pa) / 2 + (b ? (eb - pb) / 2 : 0) + 4096);
        out->offsets.reserve(out->offsets.n + (size_t)(u1 - u0) * (paired ? 2 : 1) + 1);
        out->name_offsets.reserve(out->name_offsets.n + (size_t)(u1 - u0) + 1);
        auto line = [](const Mapped &f, size_t &at, const char *&p, size_t &len) {
            const char *nl = (const char *)memchr(f.p + at, '\n', f.n - at);
            p = f.p + at;
            len = nl ? (size_t)(nl - p) : f.n - at;
            at = nl ? (size_t)(nl - f.p) + 1 : f.n;
        };
        const char *p;
        size_t len;
        for (int64_t u = u0; u < u1; ++u) {
            line(a, pa, p, len);                                  // i & 3 == 0: the name
            strip(p, len);
            if (len) { ++p; --len; }                              // strip()[1:]
            out->names.append(p, len);
            out->name_offsets.push((int64_t)out->names.n);
            line(a, pa, p, len);                                  // i & 3 == 1: the bases
            strip(p, len);
            out->add_read(p, len);
            line(a, pa, p, len);
            line(a, pa, p, len);
            if (b) {
                line(*b, pb, p, len);
                line(*b, pb, p, len);
                strip(p, len);
                out->add_read(p, len);
                line(*b, pb, p, len);
                line(*b, pb, p, len);
            }
        }
        out->units += u1 - u0;
    }

    int64_t own_batches() const
    {
        return n_batches > shard_rank ? (n_batches - shard_rank + shard_world - 1) / shard_world : 0;
    }

    void build_batch(int64_t k, skm_fastq_slab *out)
    {
        const int64_t g0 = k * batch_units, g1 = std::min(total_units, g0 + batch_units);
        const size_t step = paired ? 2 : 1;
        for (size_t f = 0; f + step <= files.size(); f += step) {
            const int64_t first = pair_first_unit[f / step], end = pair_first_unit[f / step + 1];
            const int64_t lo = std::max(g0, first), hi = std::min(g1, end);
            if (lo < hi) parse_units(f, lo - first, hi - first, out);
        }
        out->bases.push(0);
        if (out->names.n == 0) out->names.reserve(1);
    }

    void worker_main()
    {
        for (;;) {
            int64_t k;
            {
                std::unique_lock<std::mutex> hold(pm);
                // (next_claim / next_deliver count THIS reader's batches: number j is batch
                // shard_rank + j * shard_world of the sample)
                pcv.wait(hold, [&] {
                    return stop || next_claim >= own_batches() || next_claim < next_deliver + n_threads + 1;
                });
                if (stop || next_claim >= own_batches()) return;
                k = next_claim++;
            }
            const bool trace = getenv("SKM_FASTQ_TRACE") != nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            skm_fastq_slab *slab = nullptr;
            size_t cap_before = 0;
            auto t1 = t0;
            try {                              // (an allocation failure fails the reader, not the process)
                slab = take_slab();
                cap_before = slab->bases.cap;
                t1 = std::chrono::steady_clock::now();
                build_batch(shard_rank + k * shard_world, slab);
            } catch (const std::bad_alloc &) {
                if (!slab) slab = new (std::nothrow) skm_fastq_slab();
                if (slab) slab->bases.failed = true;
            }
            if (!slab) {
                std::lock_guard<std::mutex> hold(pm);
                failed = true;
                ready[k] = nullptr;
                pcv.notify_all();
                continue;
            }
            if (trace)
                fprintf(stderr, "[skm_fastq] batch %lld: slab %.1f ms (capacity %zu -> %zu), parse %.1f ms, %lld units\n", (long long)k,
                        std::chrono::duration<double, std::milli>(t1 - t0).count(), cap_before, slab->bases.cap,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(),
                        (long long)slab->units);
            {
                std::lock_guard<std::mutex> hold(pm);
                if (slab->failed()) failed = true;
                ready[k] = slab;
            }
            pcv.notify_all();
        }
    }

    // count the lines of every file with all threads; false = not eligible
    bool index_files(int threads)
    {
        const bool trace = getenv("SKM_FASTQ_TRACE") != nullptr;      // tuning aid: phases on stderr
        const auto t0 = std::chrono::steady_clock::now();
        files.resize(paths.size());
        for (size_t i = 0; i < paths.size(); ++i)
            if (!files[i].map(paths[i].c_str())) return false;
        for (auto &f : files) {
            const size_t blocks = (f.n + BLOCK - 1) / BLOCK;
            std::vector<int64_t> per(blocks, 0);
            std::vector<std::thread> pool;
            std::mutex next_mu;
            size_t next = 0;
            for (int t = 0; t < threads; ++t)
                pool.emplace_back([&]() {
                    for (;;) {
                        size_t first;
                        {
                            std::lock_guard<std::mutex> hold(next_mu);
                            first = next;
                            next += 16;
                        }
                        if (first >= blocks) return;
                        for (size_t b = first; b < std::min(blocks, first + 16); ++b)
                            per[b] = count_newlines(f.p + b * BLOCK, std::min(BLOCK, f.n - b * BLOCK));
                    }
                });
            for (auto &t : pool) t.join();
            f.before.assign(blocks + 1, 0);
            for (size_t b = 0; b < blocks; ++b) f.before[b + 1] = f.before[b] + per[b];
            f.lines = f.before[blocks] + ((f.n && f.p[f.n - 1] != '\n') ? 1 : 0);
        }
        // zip(file1, file2) stops at the shorter file; only whole records split exactly
        const size_t step = paired ? 2 : 1;
        pair_first_unit.assign(1, 0);
        for (size_t f = 0; f + step <= files.size(); f += step) {
            int64_t lines = files[f].lines;
            if (paired) lines = std::min(lines, files[f + 1].lines);
            if (lines % 4 != 0) return false;
            pair_first_unit.push_back(pair_first_unit.back() + lines / 4);
        }
        total_units = pair_first_unit.back();
        n_batches = (total_units + batch_units - 1) / batch_units;
        if (trace)
            fprintf(stderr, "[skm_fastq] %zu files mapped and %lld units indexed in %.1f ms (%d threads)\n", files.size(),
                    (long long)total_units, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), threads);
        return true;
    }
};

extern "C" int skm_fastq_open(const char *const *paths, int n_paths, int paired, int64_t batch_units,
                              skm_fastq **out)
{
    if (!paths || n_paths <= 0 || !out || batch_units <= 0) return SKM_ERR_ARG;
    if (paired && (n_paths % 2) != 0) return SKM_ERR_ARG;      // common.py:178-179 raises ValueError
    skm_fastq *q = new skm_fastq();
    for (int i = 0; i < n_paths; ++i) q->paths.emplace_back(paths[i]);
    q->paired = paired != 0;
    q->batch_units = batch_units;
    *out = q;
    return SKM_OK;
}

extern "C" int skm_fastq_set_allocator(skm_fastq *q, void *(*alloc)(size_t), void (*release)(void *))
{
    if (!q || !alloc || !release) return SKM_ERR_ARG;
    if (!q->workers.empty()) return SKM_ERR_STATE;
    q->al = alloc;
    q->fr = release;
    return SKM_OK;
}

extern "C" int skm_fastq_set_parallel(skm_fastq *q, int n_threads, int *enabled)
{
    if (!q || n_threads < 0) return SKM_ERR_ARG;
    if (enabled) *enabled = 0;
    if (q->open || q->finished || q->next_path || !q->workers.empty()) return SKM_ERR_STATE;   // before the first batch
    if (n_threads == 0) return SKM_OK;
    if (!q->index_files(n_threads)) {           // pipes, ragged files: the sequential engine reads them
        for (auto &f : q->files) f.unmap();
        q->files.clear();
        return SKM_OK;
    }
    q->n_threads = n_threads;
    if (enabled) *enabled = 1;
    return SKM_OK;
}

extern "C" int skm_fastq_recycle(skm_fastq *q, skm_fastq_slab *slab);

static int next_parallel(skm_fastq *q, int64_t *n_units)
{
    if (q->workers.empty() && q->next_deliver < q->own_batches())
        for (int t = 0; t < q->n_threads; ++t) q->workers.emplace_back([q]() { q->worker_main(); });
    if (q->cur) skm_fastq_recycle(q, q->cur);   // (a batch that was not detached)
    q->cur = nullptr;
    if (q->next_deliver >= q->own_batches()) {
        q->cur = new skm_fastq_slab();          // (an empty batch of its own: the pooled slabs stay pooled)
        q->cur->start();
        q->cur->bases.push(0);
        q->cur->names.reserve(1);
        *n_units = 0;
        return SKM_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> hold(q->pm);
    q->pcv.wait(hold, [&] { return q->ready.count(q->next_deliver) != 0; });
    if (getenv("SKM_FASTQ_TRACE"))
        fprintf(stderr, "[skm_fastq] next: waited %.1f ms for batch %lld\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                (long long)q->next_deliver);
    q->cur = q->ready[q->next_deliver];
    q->ready.erase(q->next_deliver);
    if (!q->cur) { q->failed = true; return SKM_ERR_IO; }
    q->last_index = q->shard_rank + q->next_deliver * q->shard_world;
    q->last_read_len = q->cur->read_len;
    q->next_deliver++;
    const bool failed = q->failed;
    hold.unlock();
    q->pcv.notify_all();
    if (failed) return SKM_ERR_IO;
    *n_units = q->cur->units;
    return SKM_OK;
}

extern "C" int skm_fastq_next(skm_fastq *q, int64_t *n_units, const char **bases,
                              const int64_t **offsets, const char **names,
                              const int64_t **name_offsets)
{
    if (!q || !n_units) return SKM_ERR_ARG;
    if (q->n_threads > 0) {
        const int rc = next_parallel(q, n_units);
        if (rc != SKM_OK) return rc;
        skm_fastq_slab *const b = q->cur;
        if (bases) *bases = b->bases.p;
        if (offsets) *offsets = b->offsets.p;
        if (names) *names = b->names.p;
        if (name_offsets) *name_offsets = b->name_offsets.p;
        return SKM_OK;
    }
    for (;;) {
    q->reset_batch();
    skm_fastq_slab *const b = q->cur;
    bool full = false;
    while (!full && !q->finished) {
        if (!q->open) {
            if (q->next_path >= q->paths.size()) { q->finished = true; break; }
            if (!q->r1.open(q->paths[q->next_path].c_str())) return SKM_ERR_IO;
            if (q->paired && !q->r2.open(q->paths[q->next_path + 1].c_str())) { q->r1.close(); return SKM_ERR_IO; }
            q->next_path += q->paired ? 2 : 1;
            q->open = true;
            q->line_no = 0;
        }
        const char *l1, *l2 = nullptr;
        size_t n1, n2 = 0;
        bool ok = q->r1.next(&l1, &n1, q->s1);
        if (ok && q->paired) ok = q->r2.next(&l2, &n2, q->s2);   // zip(file1, file2)
        if (!ok) {
            q->r1.close();
            if (q->paired) q->r2.close();
            q->open = false;
            continue;
        }
        const int phase = (int)(q->line_no & 3);
        q->line_no++;
        if (phase == 0) {
            strip(l1, n1);
            if (n1) { ++l1; --n1; }                              // strip()[1:]
            b->names.append(l1, n1);
            b->name_offsets.push((int64_t)b->names.n);
            q->held++;
        } else if (phase == 1) {
            strip(l1, n1);
            b->add_read(l1, n1);
            if (q->paired) {
                strip(l2, n2);
                b->add_read(l2, n2);
            }
            if (q->held >= q->batch_units) full = true;          // len(read_names) >= BUFFER_SIZE
        }
    }
    // `if reads:` -- a trailing name without bases is dropped, as in the reference
    const int64_t reads = (int64_t)b->offsets.n - 1;
    if (reads == 0) { *n_units = 0; q->held = 0; }
    else *n_units = q->held;
    b->bases.push(0);
    if (b->names.n == 0) b->names.reserve(1);
    if (b->failed()) return SKM_ERR_IO;
    // a sharded reader parses everything (a pipe cannot be skipped through) and hands out its own
    const int64_t index = q->seq_batch++;
    if (*n_units != 0 && index % q->shard_world != q->shard_rank) continue;
    q->last_index = index;
    q->last_read_len = b->read_len;
    if (bases) *bases = b->bases.p;
    if (offsets) *offsets = b->offsets.p;
    if (names) *names = b->names.p;
    if (name_offsets) *name_offsets = b->name_offsets.p;
    return SKM_OK;
    }
}

extern "C" int skm_fastq_set_shard(skm_fastq *q, int rank, int world)
{
    if (!q || world < 1 || rank < 0 || rank >= world) return SKM_ERR_ARG;
    if (q->open || q->finished || q->next_path || !q->workers.empty() || q->next_deliver) return SKM_ERR_STATE;
    q->shard_rank = rank;
    q->shard_world = world;
    return SKM_OK;
}

extern "C" int skm_fastq_batch_read_length(const skm_fastq *q, int64_t *read_len)
{
    if (!q || !read_len) return SKM_ERR_ARG;
    *read_len = q->last_read_len >= 0 ? q->last_read_len : -1;   // (survives skm_fastq_detach)
    return SKM_OK;
}

extern "C" int skm_fastq_batch_index(const skm_fastq *q, int64_t *index)
{
    if (!q || !index) return SKM_ERR_ARG;
    *index = q->last_index;
    return SKM_OK;
}

extern "C" int skm_fastq_detach(skm_fastq *q, skm_fastq_slab **slab)
{
    if (!q || !slab || !q->cur) return SKM_ERR_ARG;
    *slab = q->cur;              // the pointers of the last skm_fastq_next stay valid
    q->cur = nullptr;
    return SKM_OK;
}

extern "C" int skm_fastq_recycle(skm_fastq *q, skm_fastq_slab *slab)
{
    (void)q;                     // (slabs go back to the process-wide pool: see g_spare)
    if (!slab) return SKM_OK;
    {
        std::lock_guard<std::mutex> hold(g_spare_lock);
        if (g_spare.size() < MAX_SPARE) { g_spare.push_back(slab); return SKM_OK; }
    }
    delete slab;
    return SKM_OK;
}

extern "C" int skm_fastq_close(skm_fastq *q)
{
    if (!q) return SKM_OK;
    const auto t0 = std::chrono::steady_clock::now();
    q->r1.close();
    q->r2.close();
    delete q;
    if (getenv("SKM_FASTQ_TRACE"))
        fprintf(stderr, "[skm_fastq] closed in %.1f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return SKM_OK;
}
