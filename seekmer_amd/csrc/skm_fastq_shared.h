// Shared by the two FASTQ readers of libseekmer_host.so (skm_fastq.cpp: ASCII batches,
// skm_fastq_packed.cpp: 2-bit pieces): growable arrays over a pluggable allocator and the
// process-wide cache of file mappings.
#pragma once
#include <sys/types.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace skmfq {

typedef void *(*alloc_fn)(size_t);
typedef void (*free_fn)(void *);

// growable array in memory of the slab's allocator
template <class T>
struct Buf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    alloc_fn al = malloc;
    free_fn fr = free;
    bool failed = false;

    void reserve(size_t want)
    {
        if (want <= cap) return;
        size_t grown = std::max(want, cap + cap / 2 + 1024);
        T *q = (T *)al(grown * sizeof(T));
        if (!q) { failed = true; return; }
        if (n) memcpy(q, p, n * sizeof(T));
        if (p) fr(p);
        p = q;
        cap = grown;
    }
    void append(const T *src, size_t count)
    {
        if (n + count > cap) { reserve(n + count); if (failed) return; }
        memcpy(p + n, src, count * sizeof(T));
        n += count;
    }
    void push(T v)
    {
        if (n + 1 > cap) { reserve(n + 1); if (failed) return; }
        p[n++] = v;
    }
    void clear() { n = 0; }
    void release() { if (p) fr(p); p = nullptr; n = cap = 0; }
};


constexpr size_t BLOCK = 1 << 20;          // granularity of the newline index

// A memory-mapped text file out of the process-wide cache of mappings (skm_fastq.cpp).
struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int64_t lines = 0;                     // lines of the file (a last line without '\n' counts)
    std::vector<int64_t> before;           // before[b] = newlines in [0, b * BLOCK)

    bool map(const char *path);
    void unmap();                          // back to the cache; what exceeds the budget goes in the background
    size_t line_start(int64_t line) const; // byte offset where line `line` starts (needs `before`)
};

// bytes of unused mappings the cache may keep (default 0: a mapping goes when its last reader closes)
void set_map_keep_bytes(size_t bytes);

}  // namespace skmfq
