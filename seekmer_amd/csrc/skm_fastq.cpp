// Native FASTQ batch packer (libseekmer_host.so): the reference's Python
// feeders (/root/reference/seekmer/common.py:126-197) hold the GIL and top out
// near 0.7 M pairs/s; this reader yields the same batches -- line i&3==0 ->
// name = strip()[1:], line i&3==1 -> bases = strip(), case preserved, the '+'
// and quality lines never validated, one buffer carried across files, flush
// when batch_units names are held or at the very end -- as flat arrays
// (bases back to back + offsets) ready for skm_mapper_map_batch.
#include "../../include/seekmer_hip.h"

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

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
    std::vector<char> bases, names;
    std::vector<int64_t> offsets, name_offsets;
};

struct skm_fastq {
    std::vector<std::string> paths;
    bool paired = false;
    int64_t batch_units = 65536;
    size_t next_path = 0;
    LineReader r1, r2;
    bool open = false;
    int64_t line_no = 0;
    bool finished = false;
    // current batch
    skm_fastq_slab *cur = nullptr;
    std::vector<skm_fastq_slab *> spare;      // under spare_lock: recycled from other threads
    std::mutex spare_lock;
    int64_t held = 0;
    std::string s1, s2;

    void reset_batch()
    {
        if (!cur) {
            std::lock_guard<std::mutex> hold(spare_lock);
            if (!spare.empty()) { cur = spare.back(); spare.pop_back(); }
        }
        if (!cur) cur = new skm_fastq_slab();
        cur->bases.clear(); cur->names.clear();
        cur->offsets.assign(1, 0); cur->name_offsets.assign(1, 0);
        held = 0;
    }
    ~skm_fastq()
    {
        delete cur;
        for (skm_fastq_slab *s : spare) delete s;
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
    q->reset_batch();
    *out = q;
    return SKM_OK;
}

extern "C" int skm_fastq_next(skm_fastq *q, int64_t *n_units, const char **bases,
                              const int64_t **offsets, const char **names,
                              const int64_t **name_offsets)
{
    if (!q || !n_units) return SKM_ERR_ARG;
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
            b->names.insert(b->names.end(), l1, l1 + n1);
            b->name_offsets.push_back((int64_t)b->names.size());
            q->held++;
        } else if (phase == 1) {
            strip(l1, n1);
            b->bases.insert(b->bases.end(), l1, l1 + n1);
            b->offsets.push_back((int64_t)b->bases.size());
            if (q->paired) {
                strip(l2, n2);
                b->bases.insert(b->bases.end(), l2, l2 + n2);
                b->offsets.push_back((int64_t)b->bases.size());
            }
            if (q->held >= q->batch_units) full = true;          // len(read_names) >= BUFFER_SIZE
        }
    }
    // `if reads:` -- a trailing name without bases is dropped, as in the reference
    const int64_t reads = (int64_t)b->offsets.size() - 1;
    if (reads == 0) { *n_units = 0; q->held = 0; }
    else *n_units = q->held;
    b->bases.push_back(0);
    if (bases) *bases = b->bases.data();
    if (offsets) *offsets = b->offsets.data();
    if (names) *names = b->names.data();
    if (name_offsets) *name_offsets = b->name_offsets.data();
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
    if (!slab) return SKM_OK;
    if (q) {
        std::lock_guard<std::mutex> hold(q->spare_lock);
        if (q->spare.size() < 64) { q->spare.push_back(slab); return SKM_OK; }
    }
    delete slab;
    return SKM_OK;
}

extern "C" int skm_fastq_close(skm_fastq *q)
{
    if (!q) return SKM_OK;
    q->r1.close();
    q->r2.close();
    delete q;
    return SKM_OK;
}
