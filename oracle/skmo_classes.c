/* CPU ORACLE (test infrastructure only) -- equivalence-class counting.
 * Restates MapResult.update / summarize (seekmer/mapper.py:60-104) and
 * _get_ids (seekmer/_mapper.pyx:528-537): the key is the tuple of target ids
 * with the strand sign stripped, IN LIST ORDER (duplicates kept); classes are
 * enumerated in first-seen order (collections.Counter insertion order); the
 * empty tuple counts unaligned units. */
#include "skmo.h"
#include <stdlib.h>
#include <string.h>

struct skmo_classes {
    /* open-addressing table of class indices keyed by the id tuple */
    int64_t *slots; int64_t n_slots;
    /* classes in first-seen order */
    int64_t n_classes, cap_classes;
    int64_t *offsets;     /* [n_classes+1] into ids */
    int64_t *counts;
    uint64_t *hashes;
    int32_t *ids; int64_t n_ids, cap_ids;
    int64_t unaligned;
};

static uint64_t tuple_hash(const int32_t *ids, int n)
{
    uint64_t h = 0xcbf29ce484222325ULL ^ (uint64_t)n;
    for (int i = 0; i < n; ++i) {
        h ^= (uint32_t)ids[i];
        h *= 0x100000001b3ULL;
        h ^= h >> 29;
    }
    return h;
}

skmo_classes *skmo_classes_new(void)
{
    skmo_classes *c = (skmo_classes *)calloc(1, sizeof(*c));
    c->n_slots = 1024;
    c->slots = (int64_t *)malloc(sizeof(int64_t) * c->n_slots);
    for (int64_t i = 0; i < c->n_slots; ++i) c->slots[i] = -1;
    c->cap_classes = 256;
    c->offsets = (int64_t *)malloc(sizeof(int64_t) * (c->cap_classes + 1));
    c->counts = (int64_t *)malloc(sizeof(int64_t) * c->cap_classes);
    c->hashes = (uint64_t *)malloc(sizeof(uint64_t) * c->cap_classes);
    c->offsets[0] = 0;
    c->cap_ids = 1024;
    c->ids = (int32_t *)malloc(sizeof(int32_t) * c->cap_ids);
    return c;
}

void skmo_classes_free(skmo_classes *c)
{
    if (!c) return;
    free(c->slots); free(c->offsets); free(c->counts); free(c->hashes); free(c->ids);
    free(c);
}

static void rehash(skmo_classes *c)
{
    int64_t n = c->n_slots * 2;
    int64_t *s = (int64_t *)malloc(sizeof(int64_t) * n);
    for (int64_t i = 0; i < n; ++i) s[i] = -1;
    for (int64_t k = 0; k < c->n_classes; ++k) {
        int64_t p = (int64_t)(c->hashes[k] & (uint64_t)(n - 1));
        while (s[p] >= 0) p = (p + 1) & (n - 1);
        s[p] = k;
    }
    free(c->slots);
    c->slots = s; c->n_slots = n;
}

static void add_tuple(skmo_classes *c, const int32_t *ids, int n)
{
    if (n == 0) { c->unaligned++; return; }
    uint64_t h = tuple_hash(ids, n);
    int64_t p = (int64_t)(h & (uint64_t)(c->n_slots - 1));
    while (c->slots[p] >= 0) {
        int64_t k = c->slots[p];
        if (c->hashes[k] == h && c->offsets[k + 1] - c->offsets[k] == n
                && memcmp(c->ids + c->offsets[k], ids, sizeof(int32_t) * (size_t)n) == 0) {
            c->counts[k]++;
            return;
        }
        p = (p + 1) & (c->n_slots - 1);
    }
    if (c->n_classes == c->cap_classes) {
        c->cap_classes *= 2;
        c->offsets = (int64_t *)realloc(c->offsets, sizeof(int64_t) * (c->cap_classes + 1));
        c->counts = (int64_t *)realloc(c->counts, sizeof(int64_t) * c->cap_classes);
        c->hashes = (uint64_t *)realloc(c->hashes, sizeof(uint64_t) * c->cap_classes);
    }
    while (c->n_ids + n > c->cap_ids) {
        c->cap_ids *= 2;
        c->ids = (int32_t *)realloc(c->ids, sizeof(int32_t) * c->cap_ids);
    }
    int64_t k = c->n_classes++;
    memcpy(c->ids + c->n_ids, ids, sizeof(int32_t) * (size_t)n);
    c->n_ids += n;
    c->offsets[k + 1] = c->n_ids;
    c->counts[k] = 1;
    c->hashes[k] = h;
    c->slots[p] = k;
    if (c->n_classes * 2 > c->n_slots) rehash(c);
}

int skmo_classes_update(skmo_classes *c, int64_t n_units,
                        const int32_t *counts, const int32_t *entries)
{
    if (!c || n_units < 0) return -1;
    int32_t stackbuf[256];
    int64_t pos = 0;
    for (int64_t u = 0; u < n_units; ++u) {
        int n = counts[u];
        int32_t *buf = n <= 256 ? stackbuf : (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
        for (int i = 0; i < n; ++i) {
            int32_t e = entries[pos + i];
            buf[i] = e < 0 ? ~e : e;            /* seekmer/_mapper.pyx:533-536 */
        }
        add_tuple(c, buf, n);
        if (buf != stackbuf) free(buf);
        pos += n;
    }
    return 0;
}

int64_t skmo_classes_count(const skmo_classes *c) { return c->n_classes; }
int64_t skmo_classes_map_size(const skmo_classes *c) { return c->n_ids; }
int64_t skmo_classes_unaligned(const skmo_classes *c) { return c->unaligned; }

void skmo_classes_export(const skmo_classes *c, int64_t *class_offsets,
                         int32_t *class_targets, int64_t *class_counts)
{
    memcpy(class_offsets, c->offsets, sizeof(int64_t) * (size_t)(c->n_classes + 1));
    memcpy(class_targets, c->ids, sizeof(int32_t) * (size_t)c->n_ids);
    memcpy(class_counts, c->counts, sizeof(int64_t) * (size_t)c->n_classes);
}
