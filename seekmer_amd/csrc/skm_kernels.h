// Host-visible declarations of the kernel launchers (one translation unit per
// kernel family) and the plain structs they take.
#pragma once
#include "skm_device.h"

namespace skm {

// One batch on its way through the mapper (all pointers are device memory).
struct MapBatch {
    const uint64_t *codes;        // [n_reads][words_per_read] packed 2-bit codes
    const uint32_t *acgt;         // [n_reads][words_per_read] "is upper-case ACGT" bits
    const int64_t *offsets;       // [n_reads + 1] byte offsets of the raw reads
    int64_t n_units;
    int32_t words_per_read;
    int32_t paired;
    int32_t *workspace;           // lane-interleaved target lists
    // per-unit results
    int32_t *unit_begin, *unit_end;
    Coord *unit_anchor;
    int32_t *unit_count;
    int64_t *unit_offset;         // into unit_entries
    uint64_t *unit_key;           // 64-bit class key, 0 = empty tuple
    int32_t *unit_entries;        // signed target entries, units in arena order
    int64_t ids_capacity;
    unsigned long long *ids_cursor;
    unsigned long long *fld;      // [2000] batch-local histogram
    unsigned long long *stats;    // [16] access counters (STATS build only)
};

void launch_pack_reads(const uint8_t *bases, const int64_t *offsets, int64_t n_reads,
                       int words_per_read, uint64_t *codes, uint32_t *acgt, hipStream_t stream);
void launch_map_units(const DevIndex &ix, const MapBatch &b, int grid_blocks, bool stats,
                      hipStream_t stream);
void launch_pack_sequences(const char *bases, int64_t n_bases, uint64_t *seq2, int64_t n_words,
                           hipStream_t stream);

// ---- equivalence-class table (skm_classes.hip)
struct ClassSlot {                // 32 B
    unsigned long long key;       // 0 = empty
    unsigned long long count;
    unsigned long long first_seen;  // global unit index of the first unit of the class
    long long arena_offset;       // -1 until the tuple has been committed to the arena
};
struct ClassTable {
    ClassSlot *slots;
    uint64_t slot_mask;
    int32_t *arena;               // committed tuples (unsigned ids)
    int32_t *arena_len;           // per slot: tuple length (valid once committed)
    int64_t arena_capacity;
    unsigned long long *arena_cursor;
    unsigned long long *n_classes;
    unsigned long long *n_unaligned;
    unsigned long long *n_units;
    unsigned long long *global_fld;   // [2000]
    int *error;                   // SKM_ERR_* raised by a kernel
};
void launch_class_insert(const ClassTable &t, const MapBatch &b, int64_t unit_base,
                         int64_t *unit_slot, hipStream_t stream);
void launch_class_verify_commit(const ClassTable &t, const MapBatch &b, int64_t unit_base,
                                const int64_t *unit_slot, hipStream_t stream);
void launch_class_rehash(const ClassTable &from, const ClassTable &to, hipStream_t stream);
void launch_class_init(const ClassTable &t, hipStream_t stream);
void launch_class_compact(const ClassTable &t, int64_t *cls_offset, int32_t *cls_len,
                          double *cls_count, unsigned long long *cls_first_seen,
                          unsigned long long *cursor, hipStream_t stream);
void launch_class_merge(const ClassTable &t, int64_t n_classes, const int64_t *class_offsets,
                        const int32_t *class_targets, const int64_t *class_counts,
                        const int64_t *first_seen, hipStream_t stream);

// ---- quantification (skm_em.hip)
struct EmProblem {
    int64_t n_tx, n_classes;
    const int64_t *cls_offset;    // [C] start of each class in `ids`
    const int32_t *cls_len;       // [C]
    const int32_t *ids;           // transcript ids
    const double *cls_count;      // [C]
    const double *eff_len;        // [T]
    double *x[2];                 // ping-pong abundance vectors
    double *acc;                  // [T] scatter-add target (numerators)
    double n_total;               // sum of class counts
    double rel_tol, x_floor;
    // control block: [0]=done [1]=iters [2]=ticket [3]=max bits [4]=any [5]=nan [6]=undefined
    unsigned long long *ctl;
    int64_t max_iters, fixed_iters;
};
void launch_em_scatter(const EmProblem &p, int parity, hipStream_t stream);
void launch_em_finalize(const EmProblem &p, int parity, hipStream_t stream);
void launch_effective_lengths(const unsigned long long *fld, const double *lengths, int64_t n_tx,
                              double *out, hipStream_t stream);
void launch_multinomial(const unsigned long long *cum, int64_t n_classes, int64_t n_draws,
                        uint64_t seed, uint64_t stream_id, unsigned long long *counts,
                        hipStream_t stream);
void launch_u64_to_double(const unsigned long long *in, int64_t n, double *out, hipStream_t stream);
void launch_double_to_u64(const double *in, int64_t n, unsigned long long *out, hipStream_t stream);

}  // namespace skm
