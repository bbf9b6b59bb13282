/*
 * skmo.h -- CPU ORACLE for the `seekmer infer` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (GuanLab/seekmer,
 * /root/reference/seekmer).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (seekmer_amd/) never
 * links, imports or falls back to it.
 *
 * Parity status (see DESIGN.md "Oracle pinning"):
 *   - k-mer / coordinate / sequence primitives: PINNED against the reference's
 *     own header-inline Cython primitives compiled from /root/reference by
 *     oracle/build_ref.py (oracle/_ref/), fixtures in tests/golden/.
 *   - index builder, mapper, class counting, EM: the reference modules import
 *     `logbook`/`tables`, which are absent from this image, so they cannot be
 *     imported without stand-ins; these stages are pinned only by the
 *     reference's own test data and assertions (seekmer/test) and by the
 *     observations recorded in SURVEY.md.  Anything else: "parity unpinned".
 *
 * Each function cites the reference file:line it restates.
 */
#ifndef SKMO_H
#define SKMO_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKMO_K 25                         /* seekmer/_kmer.pxd:9-17 */
#define SKMO_INVALID_KMER 0xFFFFFFFFFFFFFFFFULL /* seekmer/_kmer.pxd:20-28 */
#define SKMO_INVALID_INDEX 0x7FFFFFFF     /* seekmer/_common.pxd:10 */
#define SKMO_MAX_FRAGMENT_LENGTH 2000     /* seekmer/_mapper.pyx:18 */
#define SKMO_ALIGN_LENGTH 8               /* seekmer/_mapper.pyx:22 */
#define SKMO_MAX_OFFSET 2                 /* seekmer/_mapper.pyx:24 */
#define SKMO_MAX_DISTANCE 4               /* seekmer/_mapper.pyx:26 */
#define SKMO_INVALID_SHIFT 0x7FFF         /* seekmer/_mapper.pyx:28 */

/* seekmer/_coordinate.pxd:8-10 */
typedef struct { int32_t entry; int32_t offset; } skmo_coord;
/* seekmer/_common.pxd:15-17 */
typedef struct { uint64_t kmer; skmo_coord position; } skmo_index_entry;
/* seekmer/_common.pxd:21-27 */
typedef struct {
    int64_t offset, length;
    uint64_t first_kmer, last_kmer;
    int64_t target_offset, target_length;
} skmo_contig;

typedef struct {
    const skmo_index_entry *kmers; int64_t n_kmers;     /* power of two */
    const skmo_contig *contigs;    int64_t n_contigs;
    const char *sequences;         int64_t n_sequences;
    const skmo_coord *targets;     int64_t n_targets;
} skmo_index;

/* access counters that define the algorithmic bytes of the mapping phase
 * (DESIGN.md "Algorithmic bytes") */
typedef struct {
    int64_t reads;          /* reads (mates) mapped                                  */
    int64_t read_bases;     /* L: bases of every mapped read                          */
    int64_t lookups;        /* map_kmer calls                                         */
    int64_t slots;          /* P: index slots examined by map_kmer                    */
    int64_t contig_reads;   /* K: ContigEntry reads                                   */
    int64_t targets_copied; /* Tc: target entries copied by map_contig                */
    int64_t targets_merged; /* Tm: index-side target entries consumed by merges       */
    int64_t seq_fetches;    /* S: 8-base contig fetches                               */
    int64_t merges;         /* _filter_on_contig calls                                */
    int64_t tuple_ids;      /* sum of |tuple| over units                              */
} skmo_stats;

/* ---- primitives (skmo_kmer.c) ---- */
uint64_t skmo_kmer_mask(void);
uint64_t skmo_two_bit_encode(char base);
uint64_t skmo_kmer_encode(const char *sequence, int offset);
uint64_t skmo_kmer_append(uint64_t kmer, char base);
uint64_t skmo_kmer_prepend(uint64_t kmer, char base);
void     skmo_kmer_decode(uint64_t kmer, char *out25);
uint64_t skmo_kmer_reverse_complement(uint64_t kmer);
int32_t  skmo_kmer_hash(uint64_t kmer);
void     skmo_sequence_reverse_complement(char *bases, int length);

/* ---- index queries (skmo_index.c) ---- */
skmo_coord skmo_map_kmer(const skmo_index *ix, uint64_t kmer, skmo_stats *st);
void skmo_get_contig_sequence(const skmo_index *ix, skmo_coord c, int length,
                              char *out, skmo_stats *st);
uint64_t skmo_get_tail_kmer(const skmo_index *ix, skmo_coord c, skmo_stats *st);

/* ---- mapper (skmo_mapper.c) ---- */
typedef struct {
    int32_t begin, end;
    skmo_coord anchor;
    int32_t n;            /* targets.size  */
    skmo_coord *items;    /* targets.items */
} skmo_span;

skmo_span skmo_map_read(const skmo_index *ix, const char *bases, int length,
                        skmo_stats *st);
skmo_span skmo_map_read_pair(const skmo_index *ix, const char *b1, int l1,
                             const char *b2, int l2, skmo_stats *st);
int skmo_sift4_align_left(const char *ref, int ref_len, const char *query,
                          int query_len, int offset);
int skmo_sift4_align_right(const char *ref, int ref_len, const char *query,
                           int query_len, int offset);

/* Map a batch.  `bases` holds all reads back to back, read r occupying
 * [offsets[r], offsets[r+1]).  Paired: reads 2u and 2u+1 are the mates of
 * unit u.  Per-unit outputs (any may be NULL):
 *   out_begin/out_end/out_anchor_entry/out_anchor_offset [n_units]
 *   out_count[n_units]              number of targets
 *   out_entries[cap_entries]        signed target entries, units back to back
 * fld[2000] is accumulated (+=).  Returns total number of target entries
 * (which may exceed cap_entries: then only the first cap_entries are stored),
 * or -1 on bad arguments. */
int64_t skmo_map_batch(const skmo_index *ix, const char *bases,
                       const int64_t *offsets, int64_t n_units, int paired,
                       int32_t *out_begin, int32_t *out_end,
                       int32_t *out_anchor_entry, int32_t *out_anchor_offset,
                       int32_t *out_count, int32_t *out_entries,
                       int64_t cap_entries, int64_t *fld, skmo_stats *st);

/* ---- equivalence classes (skmo_classes.c) ---- */
typedef struct skmo_classes skmo_classes;
skmo_classes *skmo_classes_new(void);
void skmo_classes_free(skmo_classes *c);
/* add n_units tuples (CSR of signed entries as produced by skmo_map_batch) */
int skmo_classes_update(skmo_classes *c, int64_t n_units,
                        const int32_t *counts, const int32_t *entries);
int64_t skmo_classes_count(const skmo_classes *c);      /* C (without the empty tuple) */
int64_t skmo_classes_map_size(const skmo_classes *c);   /* M */
int64_t skmo_classes_unaligned(const skmo_classes *c);
/* first-seen order: class_offsets[C+1], class_targets[M] (unsigned ids), class_counts[C] */
void skmo_classes_export(const skmo_classes *c, int64_t *class_offsets,
                         int32_t *class_targets, int64_t *class_counts);

/* ---- quantification (skmo_quant.c) ---- */
double skmo_pairwise_sum(const double *a, int64_t n);
void skmo_effective_lengths(const int64_t *fld, const double *lengths,
                            int64_t n_tx, double *out);
double skmo_harmonic_mean_fragment_length(const int64_t *fld);
/* x inout [n_tx].  max_iters<=0: reference criterion only.  fixed_iters>0: run
 * exactly that many steps.  trace (optional) receives x after each of
 * trace_iters[0..n_trace) (1-based iteration numbers), n_tx doubles each.
 * Returns the number of EM steps executed, or -1 when the reference would
 * raise (no x > 1e-8). */
int64_t skmo_em(double *x, const double *l, int64_t n_tx,
                const int64_t *class_of_pair, const int64_t *tx_of_pair,
                int64_t n_pairs, const double *class_count, int64_t n_classes,
                int64_t max_iters, int64_t fixed_iters,
                const int64_t *trace_iters, int64_t n_trace, double *trace);
void skmo_tpm(double *x, int64_t n_tx);
void skmo_est_counts(const double *tpm, const double *lengths, int64_t n_tx,
                     double aligned, double *out);

/* ---- index builder (skmo_builder.c) ---- */
typedef struct {
    skmo_index_entry *kmers; int64_t n_kmers;
    skmo_contig *contigs;    int64_t n_contigs;
    char *sequences;         int64_t n_sequences;
    skmo_coord *targets;     int64_t n_targets;
    int64_t scan_kmer_count; /* kmer_count after _scan_kmers */
} skmo_built;
/* sequences: pooled transcript bases, transcript i = [seq_offsets[i], seq_offsets[i+1]).
 * returns 0, or a negative code where the reference would hit undefined behaviour */
int skmo_build(const char *pool, const int64_t *seq_offsets, int64_t n_seqs,
               skmo_built *out);
void skmo_built_free(skmo_built *b);

#ifdef __cplusplus
}
#endif
#endif
