/* CPU ORACLE (test infrastructure only) -- index builder.
 * Restates ContigAssembler (seekmer/_index_builder.pyx:85-572): the k-mer
 * scan that only sizes the table, the de Bruijn linking pass, the slot-order
 * contig walk (whose `position` fields alias `last`/`next`), the contig ->
 * transcript target map and the table compilation.  Where the reference would
 * index outside the table (undefined behaviour) this returns a negative code. */
#include "skmo.h"
#include <stdlib.h>
#include <string.h>

/* seekmer/_index_builder.pyx:79-82; same 16 bytes as skmo_index_entry */
typedef struct { uint64_t kmer; int32_t last; int32_t next; } node_t;

typedef struct {
    node_t *t;
    int64_t size;
    int64_t kmer_count;
    /* LastKMerInfo, seekmer/_index_builder.pyx:73-76 */
    int last_is_new, last_forward, last_index;
} assembler;

/* seekmer/_index_builder.pyx:313-342 */
static int find_slot(const assembler *a, uint64_t kmer)
{
    uint64_t rc_kmer = skmo_kmer_reverse_complement(kmer);
    uint64_t index_kmer = kmer < rc_kmer ? kmer : rc_kmer;
    int size = (int)a->size;
    int mask = size - 1;
    int offset = skmo_kmer_hash(index_kmer) & mask;
    for (int i = offset; i < size; ++i)
        if (a->t[i].kmer == SKMO_INVALID_KMER || a->t[i].kmer == kmer || a->t[i].kmer == rc_kmer)
            return i;
    for (int i = 0; i < offset; ++i)
        if (a->t[i].kmer == SKMO_INVALID_KMER || a->t[i].kmer == kmer || a->t[i].kmer == rc_kmer)
            return i;
    return -1;
}

static void wipe(node_t *t, int64_t from, int64_t to)
{
    for (int64_t i = from; i < to; ++i) {
        t[i].kmer = SKMO_INVALID_KMER;
        t[i].last = SKMO_INVALID_INDEX;
        t[i].next = SKMO_INVALID_INDEX;
    }
}

/* seekmer/_index_builder.pyx:204-224 -- in-place rehash in old-slot order */
static void expand(assembler *a)
{
    int64_t old_size = a->size;
    a->t = (node_t *)realloc(a->t, sizeof(node_t) * (size_t)(old_size << 1));
    a->size = old_size << 1;
    wipe(a->t, old_size, a->size);
    for (int64_t i = 0; i < old_size; ++i) {
        if (a->t[i].kmer == SKMO_INVALID_KMER) continue;
        uint64_t kmer = a->t[i].kmer;
        int j = find_slot(a, kmer);
        if (i != j) {
            a->t[i].kmer = SKMO_INVALID_KMER;
            a->t[j].kmer = kmer;
        }
    }
}

/* seekmer/_index_builder.pyx:181-198 -- stores the k-mer as seen (not canonical) */
static void add_kmer(assembler *a, uint64_t kmer)
{
    int i = find_slot(a, kmer);
    if (a->t[i].kmer == SKMO_INVALID_KMER) {
        a->kmer_count += 1;
        if ((double)a->kmer_count > 0.8 * (double)a->size) {
            expand(a);
            i = find_slot(a, kmer);
        }
        a->t[i].kmer = kmer;
    }
}

/* seekmer/_index_builder.pyx:348-367 */
static void link_(assembler *a, int i, int j, int is_forward)
{
    if (is_forward) a->t[i].next = j; else a->t[i].last = j;
}

/* seekmer/_index_builder.pyx:373-401 */
static void unlink_(assembler *a, int i, int is_forward)
{
    int j;
    if (is_forward) { j = a->t[i].next; a->t[i].next = SKMO_INVALID_INDEX; }
    else { j = a->t[i].last; a->t[i].last = SKMO_INVALID_INDEX; }
    if (j == SKMO_INVALID_INDEX) return;
    if (a->t[j].next == i) a->t[j].next = SKMO_INVALID_INDEX;
    else a->t[j].last = SKMO_INVALID_INDEX;
}

/* seekmer/_index_builder.pyx:407-422 */
static int get_link(const assembler *a, int i, int is_forward)
{
    return is_forward ? a->t[i].next : a->t[i].last;
}

/* seekmer/_index_builder.pyx:256-307 */
static void register_kmer(assembler *a, uint64_t kmer)
{
    uint64_t rc_kmer = skmo_kmer_reverse_complement(kmer);
    uint64_t index_kmer = kmer < rc_kmer ? kmer : rc_kmer;
    int i = find_slot(a, index_kmer);
    int forward = index_kmer == kmer;
    if (a->last_index == SKMO_INVALID_INDEX) {
        if (a->t[i].kmer != SKMO_INVALID_KMER) {
            unlink_(a, i, !forward);
            a->last_is_new = 0;
        } else {
            a->t[i].kmer = index_kmer;
            a->last_is_new = 1;
        }
        a->last_forward = forward;
        a->last_index = i;
        return;
    }
    if (a->t[i].kmer == SKMO_INVALID_KMER) {
        a->t[i].kmer = index_kmer;
        if (a->last_is_new) {
            link_(a, a->last_index, i, a->last_forward);
            link_(a, i, a->last_index, !forward);
        } else {
            unlink_(a, a->last_index, a->last_forward);
        }
        a->last_is_new = 1;
        a->last_forward = forward;
        a->last_index = i;
        return;
    }
    if (get_link(a, a->last_index, a->last_forward) == i
            && get_link(a, i, !forward) == a->last_index) {
        a->last_is_new = 0;
        a->last_forward = forward;
        a->last_index = i;
        return;
    }
    unlink_(a, a->last_index, a->last_forward);
    unlink_(a, i, !forward);
    if (get_link(a, i, forward) != SKMO_INVALID_INDEX) {
        a->last_is_new = 0;
        a->last_forward = forward;
        a->last_index = i;
    } else {
        a->last_is_new = 1;
        a->last_forward = 1;
        a->last_index = SKMO_INVALID_INDEX;
    }
}

typedef struct { char *p; int64_t n, cap; } bytebuf;
static void bb_append(bytebuf *b, const char *s, int64_t n)
{
    if (b->n + n > b->cap) {
        while (b->n + n > b->cap) b->cap = b->cap ? b->cap * 2 : 4096;
        b->p = (char *)realloc(b->p, (size_t)b->cap);
    }
    memcpy(b->p + b->n, s, (size_t)n);
    b->n += n;
}

typedef struct { int32_t contig, entry, offset; } target_rec;

/* numpy structured sort = lexicographic on (contig, entry, offset), signed */
static int target_cmp(const void *pa, const void *pb)
{
    const target_rec *a = (const target_rec *)pa, *b = (const target_rec *)pb;
    if (a->contig != b->contig) return a->contig < b->contig ? -1 : 1;
    if (a->entry != b->entry) return a->entry < b->entry ? -1 : 1;
    if (a->offset != b->offset) return a->offset < b->offset ? -1 : 1;
    return 0;
}

void skmo_built_free(skmo_built *b)
{
    if (!b) return;
    free(b->kmers); free(b->contigs); free(b->sequences); free(b->targets);
    memset(b, 0, sizeof(*b));
}

int skmo_build(const char *pool, const int64_t *seq_offsets, int64_t n_seqs,
               skmo_built *out)
{
    memset(out, 0, sizeof(*out));
    assembler a;
    a.size = 1024;                                   /* _INITIAL_INDEX_SIZE, :18 */
    a.t = (node_t *)malloc(sizeof(node_t) * (size_t)a.size);
    wipe(a.t, 0, a.size);
    a.kmer_count = 0;
    a.last_is_new = 1; a.last_forward = 1; a.last_index = SKMO_INVALID_INDEX;

    /* _scan_kmers, seekmer/_index_builder.pyx:156-175: only the final table
     * size survives -- the table is wiped afterwards. */
    for (int64_t s = 0; s < n_seqs; ++s) {
        const char *seq = pool + seq_offsets[s];
        int64_t len = seq_offsets[s + 1] - seq_offsets[s];
        if (len < SKMO_K) continue;
        uint64_t kmer = skmo_kmer_encode(seq, 0) >> 2;
        for (int64_t j = SKMO_K - 1; j < len; ++j) {
            kmer = skmo_kmer_append(kmer, seq[j]);
            add_kmer(&a, kmer);
        }
    }
    out->scan_kmer_count = a.kmer_count;
    wipe(a.t, 0, a.size);

    /* _connect_kmers, :230-250 */
    for (int64_t s = 0; s < n_seqs; ++s) {
        const char *seq = pool + seq_offsets[s];
        int64_t len = seq_offsets[s + 1] - seq_offsets[s];
        if (len < SKMO_K) continue;
        a.last_is_new = 1; a.last_forward = 1; a.last_index = SKMO_INVALID_INDEX;
        uint64_t kmer = skmo_kmer_encode(seq, 0) >> 2;
        for (int64_t j = SKMO_K - 1; j < len; ++j) {
            kmer = skmo_kmer_append(kmer, seq[j]);
            register_kmer(&a, kmer);
        }
        if (a.last_index != SKMO_INVALID_INDEX)
            unlink_(&a, a.last_index, a.last_forward);
    }

    /* _assemble_contigs, :428-487.  entry aliases `last`, offset aliases `next`. */
    bytebuf pooled = { 0, 0, 0 };
    int64_t n_contigs = 0, cap_contigs = 1024;
    int64_t *contig_len = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap_contigs);
    int rc = 0;
    char text[SKMO_K];
    for (int64_t i = 0; i < a.size && rc == 0; ++i) {
        if (a.t[i].kmer == SKMO_INVALID_KMER) continue;
        if (a.t[i].last < 0) continue;
        if (a.t[i].last != SKMO_INVALID_INDEX && a.t[i].next != SKMO_INVALID_INDEX) continue;
        if (n_contigs == cap_contigs) {
            cap_contigs *= 2;
            contig_len = (int64_t *)realloc(contig_len, sizeof(int64_t) * (size_t)cap_contigs);
        }
        int contig_index = (int)n_contigs;
        if (a.t[i].last == SKMO_INVALID_INDEX && a.t[i].next == SKMO_INVALID_INDEX) {
            a.t[i].last = ~contig_index;
            a.t[i].next = ~0;
            skmo_kmer_decode(a.t[i].kmer, text);
            bb_append(&pooled, text, SKMO_K);
            contig_len[n_contigs++] = SKMO_K;
            continue;
        }
        int64_t start_n = pooled.n;
        int last_index = SKMO_INVALID_INDEX;
        int index = (int)i;
        int offset = 0;
        while (index != SKMO_INVALID_INDEX) {
            if (index < 0 || index >= a.size) { rc = -2; break; }  /* UB in the reference */
            if (a.t[index].next == last_index) {
                a.t[index].kmer = skmo_kmer_reverse_complement(a.t[index].kmer);
                int32_t tmp = a.t[index].last;
                a.t[index].last = a.t[index].next;
                a.t[index].next = tmp;
            }
            last_index = index;
            index = a.t[index].next;
            a.t[last_index].last = ~contig_index;
            a.t[last_index].next = ~offset;
            if (offset % SKMO_K == 0) {
                skmo_kmer_decode(a.t[last_index].kmer, text);
                bb_append(&pooled, text, SKMO_K);
            }
            offset += 1;
        }
        if (rc) break;
        skmo_kmer_decode(a.t[last_index].kmer, text);
        offset = SKMO_K - (offset - 1) % SKMO_K;
        bb_append(&pooled, text + offset, SKMO_K - offset);
        contig_len[n_contigs++] = pooled.n - start_n;
    }
    if (rc) { free(a.t); free(pooled.p); free(contig_len); return rc; }
    for (int64_t i = 0; i < a.size; ++i) {
        a.t[i].last = ~a.t[i].last;
        a.t[i].next = ~a.t[i].next;
    }

    /* _map_contigs, :493-542 */
    int64_t n_t = 0, cap_t = 4096;
    target_rec *recs = (target_rec *)malloc(sizeof(target_rec) * (size_t)cap_t);
    for (int64_t s = 0; s < n_seqs; ++s) {
        const char *seq = pool + seq_offsets[s];
        int64_t len = seq_offsets[s + 1] - seq_offsets[s];
        if (len < SKMO_K) continue;
        uint64_t kmer = skmo_kmer_encode(seq, 0) >> 2;
        for (int64_t j = SKMO_K - 1; j < len; ++j) {
            kmer = skmo_kmer_append(kmer, seq[j]);
            int k = find_slot(&a, kmer);
            if (a.t[k].next != 0) continue;            /* position.offset != 0 */
            if (n_t == cap_t) {
                cap_t *= 2;
                recs = (target_rec *)realloc(recs, sizeof(target_rec) * (size_t)cap_t);
            }
            int entry = (int)s;
            if (kmer != a.t[k].kmer) entry = ~entry;
            recs[n_t].contig = a.t[k].last;            /* position.entry */
            recs[n_t].entry = entry;
            recs[n_t].offset = (int32_t)(j - SKMO_K + 1);
            n_t++;
        }
    }
    qsort(recs, (size_t)n_t, sizeof(target_rec), target_cmp);

    /* _compile_contigs, :544-572 */
    skmo_contig *contigs = (skmo_contig *)calloc((size_t)(n_contigs > 0 ? n_contigs : 1), sizeof(skmo_contig));
    int64_t off = 0;
    for (int64_t c = 0; c < n_contigs; ++c) {
        contigs[c].length = contig_len[c];
        contigs[c].offset = off;
        off += contig_len[c];
    }
    for (int64_t r = 0; r < n_t; ++r) {
        if (recs[r].contig < 0 || recs[r].contig >= n_contigs) { rc = -3; break; }
        contigs[recs[r].contig].target_length += 1;   /* numpy field name: target_count */
    }
    if (rc) { free(a.t); free(pooled.p); free(contig_len); free(recs); free(contigs); return rc; }
    off = 0;
    for (int64_t c = 0; c < n_contigs; ++c) {
        contigs[c].target_offset = off;
        off += contigs[c].target_length;
        contigs[c].first_kmer = skmo_kmer_encode(pooled.p + contigs[c].offset, 0);
        contigs[c].last_kmer = skmo_kmer_encode(pooled.p + contigs[c].offset,
                                                (int)(contigs[c].length - SKMO_K));
    }
    skmo_coord *targets = (skmo_coord *)malloc(sizeof(skmo_coord) * (size_t)(n_t > 0 ? n_t : 1));
    for (int64_t r = 0; r < n_t; ++r) {
        targets[r].entry = recs[r].entry;
        targets[r].offset = recs[r].offset;
    }
    free(recs); free(contig_len);

    out->kmers = (skmo_index_entry *)a.t;
    out->n_kmers = a.size;
    out->contigs = contigs;
    out->n_contigs = n_contigs;
    out->sequences = pooled.p;
    out->n_sequences = pooled.n;
    out->targets = targets;
    out->n_targets = n_t;
    return 0;
}
