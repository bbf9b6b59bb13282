// Host-visible declarations of the kernel launchers (one translation unit per
// kernel family) and the plain structs they take.
#pragma once
#include "skm_device.h"

namespace skm {

// One batch on its way through the mapper (all pointers are device memory).
struct MapBatch {
    const uint32_t *records;      // [n_reads][record_words]: codes (u64 x W), ACGT bits (u32 x W), length
    int64_t n_units;
    int32_t words_per_read;       // W
    int32_t record_words;         // u32 words per record, a multiple of 16 (64 bytes)
    int32_t paired;
    int32_t *workspace;           // per-context mask extension words (slices > 64 targets)
    // Results.  Units finish out of order, so a wave writes what it finishes as RECORDS: 64
    // finished units take 64 consecutive places of their block's own range (block b owns
    // records [b * per_block, ...) like it owns those units), and the three stores of a wave
    // are full sectors.  Record r = {unit index, class key, arena offset | tuple length << 40}.
    int32_t *rec_unit;
    uint64_t *rec_key;            // 64-bit class key, 0 = empty tuple
    unsigned long long *rec_tuple;
    // per-unit spans (begin, end, anchor of MappedSpan, _common.pxd:31-35), by unit index;
    // only written when keep_spans is set (parity tests, diagnostics): nothing on the infer
    // path reads them
    int32_t keep_spans;
    int32_t *unit_begin, *unit_end;
    Coord *unit_anchor;
    int32_t *unit_entries;        // signed target entries, units in arena order
    int64_t ids_capacity;
    unsigned long long *ids_cursor;
    unsigned long long *fld;      // [2000] batch-local histogram
    unsigned long long *stats;    // [16] access counters (STATS build only)
    int32_t vote[8];              // quorum per action: start, lookup, merge, left, right, emit, scan
};

// map kernel geometry: lanes per persistent block and the occupancy the register
// allocator is asked to fit (waves per SIMD); 4 blocks per CU either way (LDS)
#ifndef SKM_MAP_THREADS
#define SKM_MAP_THREADS 256
#endif
#ifndef SKM_MAP_WAVES_PER_EU
#define SKM_MAP_WAVES_PER_EU 4
#endif
constexpr int MAP_THREADS = SKM_MAP_THREADS;
constexpr int MAP_BLOCKS_PER_CU = 4;
#ifndef SKM_MAP_CONTEXTS
#define SKM_MAP_CONTEXTS 480
#endif
constexpr int MAP_CONTEXTS = SKM_MAP_CONTEXTS;   // unit contexts per block: 16 words + 7 ring entries each, 4 blocks in 160 KB of LDS

void launch_pack_reads(const uint8_t *bases, const int64_t *offsets, int64_t n_reads,
                       int words_per_read, int record_words, uint32_t *records, hipStream_t stream);
// host-packed reads (skm_packed_reads) -> records: `n_reads` reads whose code words lie `stride`
// u64 words apart go to dst, dst + dst_stride, ... (u32 words); *error = SKM_ERR_ARG when a length
// exceeds 32 * code_words.  Then the bit planes of the exception reads (indices relative to
// `first_read` of the same piece).
void launch_unpack_reads(const uint64_t *codes, int64_t stride, int code_words, const uint32_t *lengths,
                         uint32_t uniform_len, int64_t n_reads, int words_per_read, uint32_t *dst,
                         int64_t dst_stride, int *error, hipStream_t stream);
void launch_unpack_exceptions(const uint32_t *exc_reads, const uint32_t *exc_masks, int64_t n_exceptions,
                              int code_words, int64_t first_read, int words_per_read, uint32_t *dst,
                              int64_t dst_stride, hipStream_t stream);
// out[0] = longest read, out[1] = places where the offsets step backwards; then offsets -= base
void launch_offsets_scan(int64_t *offsets, int64_t n_reads, int64_t base, unsigned long long *out,
                         hipStream_t stream);
void launch_offsets_uniform(int64_t *offsets, int64_t n_reads, int64_t read_len, hipStream_t stream);
// build the bucket table from the reference table and check the reference probe (see DevBucket);
// report: [0] placed [1] placed outside the home bucket [2] k-mers met twice [3] slots the
// reference probe does not reach
void launch_bucket_build(const DevIndex &ix, uint64_t n_slots, DevBucket *buckets, uint32_t bucket_mask,
                         uint32_t bucket_shift, unsigned long long *report, hipStream_t stream);
// DevContig::succ of every record, by lookups over ix's bucket table (skm_index_create, once)
void launch_signature_build(const DevBucket *buckets, uint64_t n_buckets, uint64_t *signatures, uint32_t shift,
                            hipStream_t stream);
void launch_successor_build(const DevIndex &ix, DevContig *records, int64_t n_contigs, int force_lookup,
                            hipStream_t stream);
// stats: 0 production, 1 counting build, 2 census build (skm_map.hip)
void launch_map_units(const DevIndex &ix, const MapBatch &b, int grid_blocks, int stats,
                      hipStream_t stream);
void launch_pack_sequences(const char *bases, int64_t n_bases, uint64_t *seq2, int64_t n_words,
                           hipStream_t stream);

void launch_gather_probe(const void *table, uint64_t n_slots, int blocks, int per_lane, int chain,
                         unsigned long long *sink, hipStream_t stream);

// ---- equivalence-class table (skm_classes.hip)
struct alignas(32) ClassSlot {    // 32 B; key and first_seen side by side: one 16-byte load per probe
    unsigned long long key;       // 0 = empty
    unsigned long long first_seen;  // global unit index of the first unit of the class
    unsigned long long count;
    long long tuple;              // -1 until the tuple has been committed to the arena, then
                                  // arena offset (bits 0-39) | tuple length (bits 40-62)
};
__host__ __device__ inline long long tuple_pack(long long offset, int n) { return offset | ((long long)n << 40); }
__host__ __device__ inline long long tuple_offset(long long t) { return t & ((1LL << 40) - 1); }
__host__ __device__ inline int tuple_len(long long t) { return (int)(t >> 40); }
struct ClassTable {
    ClassSlot *slots;
    uint64_t slot_mask;
    int32_t *arena;               // committed tuples (unsigned ids)
    int64_t arena_capacity;
    unsigned long long *arena_cursor;
    unsigned long long *n_classes;
    unsigned long long *n_unaligned;
    unsigned long long *n_units;
    unsigned long long *global_fld;   // [2000]
    int64_t *class_list;          // dense registry: slot of every committed class
    int64_t class_list_capacity;
    unsigned long long *n_listed;
    unsigned long long *n_deferred;   // units whose probe ran past PROBE_LIMIT (table too full)
    unsigned long long *arena_committed;   // arena cursor as of the last finished launch
    int *error;                   // SKM_ERR_* raised by a kernel
};
constexpr int CLASS_PROBE_LIMIT = 128;
// (the class kernels walk the batch record by record; unit_slot is indexed by record.)  insert:
// find-or-create + count + commit of the new classes + on-the-spot compare with classes of earlier
// launches, then the totals step (publishes the arena cursor; merge_fld: the batch histogram once)
void launch_class_insert(const ClassTable &t, const MapBatch &b, int64_t unit_base,
                         int64_t *unit_slot, bool retry_deferred, bool merge_fld, hipStream_t stream);
void launch_class_verify(const ClassTable &t, const MapBatch &b, const int64_t *unit_slot,
                         hipStream_t stream);
void launch_class_rehash(const ClassTable &from, const ClassTable &to, int64_t *forward,
                         hipStream_t stream);
void launch_slot_remap(int64_t *slots, int64_t n, const int64_t *forward, hipStream_t stream);
void launch_class_init(const ClassTable &t, hipStream_t stream);
void launch_class_compact(const ClassTable &t, int64_t n_classes, int64_t *cls_offset,
                          int64_t *cls_len, double *cls_count, unsigned long long *cls_first_seen,
                          hipStream_t stream);
void launch_class_merge(const ClassTable &t, int64_t n_classes, const int64_t *class_offsets,
                        const int32_t *class_targets, const int64_t *class_counts,
                        const int64_t *first_seen, hipStream_t stream);

// the same with the foreign table in HBM as class_compact leaves it (registry order, counts as
// doubles) + its unit totals and histogram: nothing crosses the host
void launch_class_merge_device(const ClassTable &t, int64_t n_classes, const int64_t *class_start,
                               const int64_t *class_len, const int32_t *ids, const double *class_counts,
                               const unsigned long long *first_seen, unsigned long long unaligned,
                               unsigned long long units, const unsigned long long *fld, hipStream_t stream);

// ---- quantification (skm_em.hip, skm_quant_setup.hip)
constexpr int EM_ROW_CAP = 512;   // longest run of one transcript's classes summed by one lane group

struct EmProblem {
    int64_t n_tx, n_classes, n_rows;
    // class-major side: class c owns ids[cls_offset[c] .. cls_offset[c+1])
    const int64_t *cls_offset;    // [C+1]
    const int32_t *ids;           // [M] transcript ids, tuple order inside a class
    const double *cls_count;      // [C]
    double *inner;                // [C] S_c / count_c of the current step
    // transcript-major side: rows = runs of <= EM_ROW_CAP entries of one transcript
    const int64_t *row_start;     // [R+1] into tx_cls
    const int32_t *row_tx;        // [R]
    const int32_t *tx_cls;        // [M] class index of every (transcript, class) pair, by transcript
    const int64_t *tx_row;        // [T+1] rows of each transcript
    double *row_sum;              // [R]
    const double *eff_len;        // [T]
    double *x[2];                 // ping-pong abundance vectors
    double *acc;                  // [T] numerators (multi-GPU all-reduce buffer)
    double n_total;               // sum of class counts over all ranks
    double rel_tol, x_floor;
    // control block: [0]=done [1]=iters [2]=ticket [3]=undefined; partials follow
    unsigned long long *ctl;
    double *part_max;             // [EM_FINAL_BLOCKS]
    unsigned int *part_flags;     // [EM_FINAL_BLOCKS] bit0 = any, bit1 = nan
    int64_t max_iters, fixed_iters;
    // one rank: rows and finalize are ONE launch (em_rows_finalize_kernel); `arrivals` [T] counts the
    // rows of a many-row transcript that have been summed in the current step (zero between steps)
    int fused;
    unsigned int *arrivals;
};
#ifndef SKM_EM_FINAL_BLOCKS
#define SKM_EM_FINAL_BLOCKS 2048
#endif
constexpr int EM_FINAL_BLOCKS = SKM_EM_FINAL_BLOCKS;
// judge_previous: apply the stopping rule to finalize pass `steps_done` first (see em_evaluate)
void launch_em_inner(const EmProblem &p, int parity, bool judge_previous, int64_t steps_done, hipStream_t stream);
void launch_em_decide(const EmProblem &p, int64_t steps_done, hipStream_t stream);
void launch_em_rows(const EmProblem &p, int parity, hipStream_t stream);
void launch_em_rows_finalize(const EmProblem &p, int parity, hipStream_t stream);
void launch_em_rows_acc(const EmProblem &p, int parity, hipStream_t stream);   // several ranks: em_rows + em_rows_to_acc
void launch_em_rows_to_acc(const EmProblem &p, hipStream_t stream);
void launch_em_finalize(const EmProblem &p, int parity, bool from_acc, hipStream_t stream);
// out = the result of an EM that latched after ctl[1] steps (x0 if even, x1 if odd)
void launch_em_result(const unsigned long long *ctl, const double *x0, const double *x1, int64_t n, double *out,
                      hipStream_t stream);

// ---- the EM for EM_BATCH problems of one class structure side by side (skm_em_batch.hip):
// the bootstrap replicates.  Arrays with a replicate dimension are [item][EM_BATCH].
constexpr int EM_BATCH = 8;
struct EmBatchProblem {
    int64_t n_tx, n_classes, n_rows;
    const int64_t *cls_offset;    // shared structure: as EmProblem
    const int32_t *ids;
    const int64_t *row_start;
    const int32_t *row_tx;
    const int32_t *tx_cls;
    const int64_t *tx_row;
    const double *eff_len;
    const double *cls_count;      // [C][EM_BATCH]
    double *inner;                // [C][EM_BATCH]
    double *row_sum;              // [R][EM_BATCH]
    double *x[2];                 // [T][EM_BATCH] ping-pong
    double n_total;               // the same for every replicate (a resample keeps the total)
    double rel_tol, x_floor;
    // control block (32 words): [0] all stopped [1] step at which the last one stopped,
    // [8 + r] replicate r stopped, [16 + r] its step count, [24 + r] undefined (no x above x_floor)
    unsigned long long *ctl;
    double *part_max;             // [EM_FINAL_BLOCKS][EM_BATCH]
    unsigned int *part_flags;     // [EM_FINAL_BLOCKS][EM_BATCH]
    int managed;                  // the device refills the places (launch_em_batch_manage): see skm_em_batch.hip
    unsigned long long *mgr;      // managed: the manager's words, planned for inside em_inner_batch (or nullptr)
    int64_t *iters_out;           // managed: step count of every replicate of the group
    int fused;                    // rows and finalize are one launch (em_rows_finalize_batch_kernel)
    unsigned int *arrivals;       // [T] rows of a many-row transcript summed so far in this step (zero between steps)
};
// The working set kept full by the device: `mgr` is 64 words of HBM; counts_all[i][C] the pre-drawn
// class counts of replicate i of the group, out_all[i][T] its result, iters_out[i] its step count
// (device memory).  _init fills the first places; _manage after EVERY step takes what has stopped and
// puts the next replicates in.  ctl[0] is set once every replicate of the group has finished; mgr[2]
// counts the finished ones, mgr[3] != 0: a replicate had no abundance above x_floor.
void launch_em_batch_manage_init(const EmBatchProblem &p, unsigned long long *mgr, unsigned long long *host_pinned64,
                                 int64_t n_reps, const double *counts_all, const double *x_start, double *out_all,
                                 hipStream_t stream);
void launch_em_batch_manage(const EmBatchProblem &p, unsigned long long *mgr, const double *counts_all, const double *x_start,
                            double *out_all, int64_t *iters_out, int64_t step, bool planned, hipStream_t stream);
// one step (inner, rows, finalize); step > 0 first judges the step before it
void launch_em_batch_step(const EmBatchProblem &p, int64_t step, hipStream_t stream);
void launch_em_batch_decide(const EmBatchProblem &p, int64_t steps_done, hipStream_t stream);
// out[t] = x[t][r]
void launch_em_batch_take(const double *x, int64_t n_tx, int r, double *out, hipStream_t stream);
// fresh control block; bit r of `idle`: place r holds no replicate and counts as stopped
void launch_em_batch_ctl(unsigned long long *ctl, unsigned int idle, hipStream_t stream);
// device-side construction of the two CSR views (skm_quant_setup.hip)
struct QuantBuild {
    int64_t n_tx, n_classes, n_ids;
    int64_t *cls_offset;          // [C+1] out
    int32_t *ids;                 // [M]   out (class-major)
    double *cls_count;            // [C]   out
    int32_t *tx_cls;              // [M]   out
    int64_t *tx_row;              // [T+1] out
    int64_t *row_start;           // [R+1] out (capacity n_rows_cap + 1)
    int32_t *row_tx;              // [R]   out
    int64_t n_rows_cap;
    int64_t first_seen_bound;     // every first-seen value of the table is below this (0: unknown)
};
// One asynchronous pipeline: classes (from a mapper's table when `table` is given, the caller's
// order being first-seen order) in (smallest transcript id, caller's index) order for gather
// locality, perm[k] = caller's index of internal class k -> transcript-major rows.  Returns the
// number of rows, or < 0.
int64_t quant_setup(const ClassTable *table, QuantBuild &q, int32_t *perm, hipStream_t stream);
// y[k] = x[perm[k]] (gather) or y[perm[k]] = x[k] (scatter), n doubles
void launch_permute_f64(const double *x, const int32_t *perm, int64_t n, double *y, bool scatter,
                        hipStream_t stream);
const char *quant_setup_failure();       // what made the last quant_setup of this thread return < 0
int64_t quant_rows_upper_bound(int64_t n_tx, int64_t n_ids);

// numpy.sum(a) bit for bit -> out[0], out[1] = out[0] / divisor; block_sums: ceil(n/8192) doubles
void launch_np_sum(const double *a, int64_t n, double divisor, double *block_sums, double *out,
                   hipStream_t stream);
// the same for `count` vectors `stride` elements apart: block_sums: count * ceil(n / 8192) doubles,
// out[2 v] = sum of vector v, out[2 v + 1] = sum / divisor
void launch_np_sum_many(const double *a, int64_t n, int64_t count, int64_t stride, double divisor,
                        double *block_sums, double *out, hipStream_t stream);
void launch_reciprocal(const double *l, int64_t n, double *x, hipStream_t stream);
void launch_divide(double *x, int64_t n, const double *s, bool threshold, double floor, hipStream_t stream);
// vector v of `count` (n elements, `stride` apart) divided by s[2 v]
void launch_divide_many(double *x, int64_t n, int64_t count, int64_t stride, const double *s, bool threshold,
                        double floor, hipStream_t stream);
void launch_effective_lengths(const unsigned long long *fld, const double *lengths, int64_t n_tx,
                              double *out, hipStream_t stream);
// multinomial(n_draws, counts / n_draws) over the classes whose inclusive cumulative counts are
// `cum`: counts[c * stride] = draws of class c (f8).  tile_total: 4096 unsigned ints of scratch.
// false = table too large for the tiled draw.
bool launch_multinomial(const unsigned long long *cum, int64_t n_classes, int64_t n_draws,
                        uint64_t seed, uint64_t stream_id, unsigned int *tile_total, double *counts,
                        int stride, hipStream_t stream);
void launch_u64_to_double(const unsigned long long *in, int64_t n, double *out, hipStream_t stream);
void launch_double_to_u64(const double *in, int64_t n, unsigned long long *out, hipStream_t stream);

// one hipFuncGetAttributes per translation unit: its code object is loaded now, not by a sample's first launch
void warm_code_map();
void warm_code_classes();
void warm_code_em();
void warm_code_em_batch();
void warm_code_quant_setup();

}  // namespace skm
