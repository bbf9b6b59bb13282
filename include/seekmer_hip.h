/*
 * seekmer_hip.h -- C ABI of the MI355X (gfx950) engine behind `seekmer infer`.
 *
 * The reference (GuanLab/seekmer) has no FFI seam on this path: its Python
 * layer calls Cython extension classes directly.  Every entry point below
 * names the reference interface it replaces (file:line under
 * /root/reference); INTEGRATION.md shows the ctypes binding a maintainer
 * would add to the reference.  Plain pointers and sizes only; no exceptions
 * cross the boundary: every function returns an int status (0 = SKM_OK) and
 * skm_last_error() gives the message for the calling thread.
 *
 * Two shared libraries implement it:
 *   libseekmer_hip.so   -- everything that touches the GPU (skm_index_*,
 *                          skm_mapper_*, skm_quant_*, skm_comm_*, skm_device_*)
 *   libseekmer_host.so  -- host-side native code with no GPU dependency
 *                          (skm_build_*, skm_fastq_*, skm_fastq_packed_*, skm_pack_*, skm_synth_*)
 */
#ifndef SEEKMER_HIP_H
#define SEEKMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKM_OK 0
#define SKM_ERR_ARG 1          /* bad argument */
#define SKM_ERR_HIP 2          /* HIP runtime error (message has the call) */
#define SKM_ERR_NO_DEVICE 3    /* no usable GPU */
#define SKM_ERR_COLLISION 4    /* two different class tuples share a 64-bit key */
#define SKM_ERR_STATE 5        /* call order / capacity */
#define SKM_ERR_IO 6
#define SKM_ERR_UNDEFINED 7    /* the reference's behaviour is undefined for this input */
#define SKM_ERR_COMM 8         /* RCCL error */

#define SKM_KMER_SIZE 25            /* seekmer/_kmer.pxd:9-17 */
#define SKM_MAX_FRAGMENT_LENGTH 2000 /* seekmer/_mapper.pyx:18-20 */

/* Array element layouts are the reference's numpy dtypes:
 *   kmers    {u64 kmer; i32 entry; i32 offset}                     16 B  seekmer/_common.pxd:15-17
 *   contigs  {i64 offset,length; u64 first_kmer,last_kmer;
 *             i64 target_offset,target_count}                      48 B  seekmer/_common.pxd:21-27
 *   sequences  char 'ACGT'                                               seekmer/_index_builder.pyx:568
 *   targets  {i32 entry; i32 offset}                                8 B  seekmer/_coordinate.pxd:8-10 */

const char *skm_last_error(void);
/* number of visible GPUs; SKM_ERR_NO_DEVICE when there is none */
int skm_device_count(int *count);

/* Raw HBM buffers for callers that keep batches resident (bench, pipelines). */
int skm_device_malloc(int device, int64_t bytes, void **out);
int skm_device_free(int device, void *ptr);
int skm_device_upload(int device, void *dst, const void *src, int64_t bytes);
int skm_device_download(int device, void *dst, const void *src, int64_t bytes);
int skm_device_synchronize(int device);
/* Page-locked host memory (hipHostMalloc).  A host batch that lives in it crosses PCIe by DMA
 * at the full link rate; from pageable memory the runtime stages the copy.  Plain allocator
 * signatures: skm_fastq_set_allocator takes this pair so that FASTQ slabs are page-locked. */
void *skm_pinned_alloc(size_t bytes);
void skm_pinned_free(void *ptr);
/* The GPU whose context page-locks the memory (default 0; one process per GPU sets its own before
 * the FASTQ reader's threads allocate).  The memory is portable: any device of the process may
 * copy from it. */
int skm_pinned_set_device(int device);
/* Diagnostic: rate of random 16-byte gathers over a table of `table_bytes`
 * (power of two); chain=0 independent (throughput), chain=1 dependent
 * (latency under load).  The ceiling the index probes are priced against. */
int skm_device_gather_ceiling(int device, int64_t table_bytes, int blocks, int per_lane,
                              int chain, double *gathers_per_second);

/* ------------------------------------------------------------------ index
 * Replaces the memoryview binding of KMerIndex.__init__
 * (seekmer/_common.pyx:21-48).  Host arrays are borrowed for the call and
 * copied to HBM (the pooled sequences are re-packed to 2 bits per base); the
 * handle owns device memory only.  Limits (SKM_ERR_ARG beyond them): n_slots a
 * power of two <= 2^31, n_contigs < 2^25, n_targets + 32 n_contigs < 2^30 (the
 * contig records -- two 64-byte sides, one per end of the contig -- carry the first
 * eight targets and the junction successors of that end: one int32 address space),
 * n_bases < 2^31, at most 2^22 - 1 targets per contig.
 * skm_index_destroy gives up the caller's handle; the device copy is released once
 * every mapper created from it has been destroyed too (in either order). */
typedef struct skm_index skm_index;
int skm_index_create(const void *kmers, int64_t n_slots,
                     const void *contigs, int64_t n_contigs,
                     const char *sequences, int64_t n_bases,
                     const void *targets, int64_t n_targets,
                     int device, skm_index **out);
int skm_index_destroy(skm_index *index);
/* info[0]=n_slots [1]=n_contigs [2]=n_bases [3]=n_targets [4]=max target_count
 * [5]=device bytes held [6]=1 when every contig's first_kmer/last_kmer spell
 * its first/last k pooled bases (the mapper then takes 8-base windows at contig
 * ends from the contig row instead of the pool) [7]=1 when every contig's target
 * slice ascends by signed entry (short list merges then run on registers) */
int skm_index_info(const skm_index *index, int64_t info[8]);
/* How the device copy of the k-mer table is probed.  The reference's table is a set
 * (KMerIndex.map_kmer, seekmer/_common.pyx:54-97, returns the position stored with a k-mer
 * or none); the device keeps the same set a second time in 64-byte buckets of four entries
 * under a cheap hash, one sector per lookup.  layout[0]=1 when that copy is in use, 0 when
 * the reference's own layout is probed (a table the reference's probe does not reach
 * everywhere, or one holding a k-mer twice or with bits above 2k); [1]=buckets [2]=k-mers placed
 * [3]=placed outside their home bucket [4]=k-mers met twice or with such bits [5]=slots the
 * reference's probe does not reach
 * [6]=1 when every contig record carries its junction successors: the map_kmer results of the
 * eight k-mers a hop of _filter_targets_to_left/right (seekmer/_mapper.pyx:246-248, 308-310) can
 * ask for when it leaves the contig, computed once at upload, so that a hop reads its answer
 * from the record instead of visiting the k-mer table (results identical by construction;
 * off for a table probed in the reference's layout).  A junction k-mer that is neither the
 * first nor the last k-mer of its contig -- none in a built index -- is marked "look it up".
 * [7]=slots of the signature table (0: none): per minimizer of the table's k-mers a 64-bit word
 * that says which k-mers it has; the roll past a sequencing error (_find_first_kmer,
 * seekmer/_mapper.pyx:207-216) asks it before the table -- a k-mer whose bit is clear is not in
 * the table, and a run of the read's k-mers shares a minimizer.  Results identical by
 * construction (no false negatives); SKM_NO_SIGNATURES=1 leaves it out. */
int skm_index_layout(const skm_index *index, int64_t layout[8]);

/* ------------------------------------------------------------------ mapper
 * One handle = ReadMapper + the MapResult it feeds
 * (seekmer/_mapper.pyx:31-105; seekmer/mapper.py:40-115): it maps batches
 * and accumulates the equivalence-class counter and the fragment-length
 * histogram in HBM. */
typedef struct skm_mapper skm_mapper;
int skm_mapper_create(skm_index *index, skm_mapper **out);
int skm_mapper_destroy(skm_mapper *mapper);

/* ReadMapper.__call__ for one batch (seekmer/_mapper.pyx:73-101): `bases`
 * holds the reads back to back, read r = [offsets[r], offsets[r+1]);
 * n_reads = n_units (single-ended) or 2*n_units (paired: reads 2u, 2u+1 are
 * the mates of unit u, the layout feed_pair_ended_reads yields,
 * seekmer/common.py:161-197).  Host buffers; copied to the GPU. */
int skm_mapper_map_batch(skm_mapper *mapper, const char *bases,
                         const int64_t *offsets, int64_t n_units, int paired);
/* The same without waiting for the kernels: the batch is copied to HBM on a staging stream
 * by the calling thread (returns once the host arrays are free again) and queued; one worker
 * per mapper maps the queued batches in submission order, so the copy of batch i+1 runs under
 * the kernels of batch i -- the overlap the reference gets from parsing in one thread and
 * mapping in others (seekmer/mapper.py:174-189).  Any thread may submit; integer results do
 * not depend on the interleaving.  first_unit >= 0 gives the batch's place in the sample (the
 * global index of its first unit): first-seen class order is then the -j1 order whatever the
 * submission order; -1 = after the units mapped so far.  skm_mapper_sync waits for everything
 * queued and returns the first failure; every call that reads or changes the table waits too.
 * After a failed batch the handle holds a partial table until skm_mapper_reset / _clear. */
int skm_mapper_map_batch_async(skm_mapper *mapper, const char *bases,
                               const int64_t *offsets, int64_t n_units, int paired,
                               int64_t first_unit);
/* A batch whose reads all have the same length (raw Illumina reads): `bases` holds them back to
 * back, read r = [r * read_len, (r + 1) * read_len); no offsets cross PCIe (8 bytes per read, 7 %
 * of a 2x100 bp batch), the device makes them.  The fixed-stride form of SURVEY.md 8(b).2. */
int skm_mapper_map_batch_uniform_async(skm_mapper *mapper, const char *bases, int32_t read_len,
                                       int64_t n_units, int paired, int64_t first_unit);
int skm_mapper_sync(skm_mapper *mapper);
/* An optional hint before a sample's reads arrive: about this many units will be mapped (a reader
 * knows its files' sizes).  The class table and the batch buffers are then sized once instead of
 * growing under the sample's first launches; results do not depend on it. */
int skm_mapper_expect_units(skm_mapper *mapper, int64_t n_units);
/* Reads packed on the host.
 * The mapper works on 2-bit codes and one "is an upper-case ACGT" bit per base
 * (seekmer/_kmer.pxd:253-273, seekmer/_mapper.pyx:500-501); in FASTQ text nearly every read has all
 * of those bits set.  A piece carries its reads as code words only (32 bases per u64 word, first
 * base in the top two bits, zero beyond the read's end: 32 bytes for a 100-base read) plus an
 * exception entry -- the bit plane -- for each read that holds any other character. */
#define SKM_PACKED_CUT (-1)
typedef struct skm_packed_reads {
    int32_t stream;                  /* 0 = single-end reads / mate 1 files, 1 = mate 2 files */
    int32_t code_words;              /* u64 words per read in `codes`; SKM_PACKED_CUT with n_reads == 0: a cut */
    int64_t first_read;              /* place of reads[0] in its stream = the unit it belongs to */
    int64_t n_reads;                 /* 0 = end of the sample -- or, with code_words == SKM_PACKED_CUT, a cut:
                                        the stream's reads from first_read on are dropped (the reader sends
                                        one when the mate-2 file of a pair of files was the longer one, before
                                        any read of the next pair of files) and more pieces follow */
    int64_t read_stride;             /* u64 words from one read's codes to the next (>= code_words) */
    int64_t uniform_len;             /* >= 0: every read is this long and `lengths` may be NULL */
    const uint64_t *codes;           /* read r = codes[r * read_stride .. + code_words) */
    const uint32_t *lengths;         /* [n_reads] */
    int64_t n_exceptions;
    const uint32_t *exception_reads; /* [n_exceptions] indices into this piece, ascending */
    const uint32_t *exception_masks; /* [n_exceptions][code_words]: bit 31 - i of word w set = base
                                        32 w + i is an upper-case A, C, G or T */
    const char *names;               /* stream 0 only, and only when asked for: names back to back */
    const int64_t *name_offsets;     /* [n_reads + 1] */
} skm_packed_reads;
/* ReadMapper.__call__ (seekmer/_mapper.pyx:73-101) for reads that arrive packed.  A piece is copied
 * to HBM by the calling thread (the call returns when its arrays are free again) and joins its
 * stream; the mapper's worker maps, in the background and in launches as large as what has arrived,
 * every run of units that its streams cover (single-ended: stream 0 alone; paired: unit u = read u
 * of stream 0 + read u of stream 1).  A piece whose first_read lies below the end of what its stream
 * holds replaces the reads from there on (a cut -- see skm_packed_reads -- replaces them by
 * nothing; either waits for a launch that is mapping from the replaced piece).  skm_mapper_sync -- and every call that reads the table
 * -- first maps every unit whose reads have all arrived; reads still without a mate are not part
 * of any result (zip(file1, file2): seekmer/common.py:180-197) and wait in HBM until their mates
 * come or the handle is reset, cleared or destroyed.  Class order does not depend on how the
 * pieces were cut or when they arrived: first-seen values are unit numbers. */
int skm_mapper_push_packed(skm_mapper *mapper, const skm_packed_reads *piece, int paired);
/* Drain a source of pieces (skm_fastq_packed_next with its reader as context) into the mapper
 * without leaving native code: next() until a piece with n_reads == 0 that is not a cut, every piece
 * pushed.  A source must keep the arrays of a piece valid until it has been asked for the next
 * piece BUT ONE: the copy of a piece to the GPU runs while the next piece is being fetched.
 * *n_pieces (optional) = pieces pushed. */
typedef int (*skm_packed_source)(void *context, skm_packed_reads *piece);
int skm_mapper_map_packed_source(skm_mapper *mapper, skm_packed_source next, void *context,
                                 int paired, int64_t *n_pieces);
/* Same with the batch already resident in HBM (device pointers; max_read_len
 * must bound every read length). */
int skm_mapper_map_batch_device(skm_mapper *mapper, const void *d_bases,
                                const void *d_offsets, int64_t n_units,
                                int paired, int32_t max_read_len);
/* Per-unit results of the LAST batch (what ReadMapper keeps in `results` and
 * `span`, seekmer/_mapper.pyx:83-99): any pointer may be NULL.  counts[u] =
 * number of targets, entries = signed target entries of all units back to
 * back in unit order (cap_entries bounds it; *n_entries = needed size).
 * begin / end / anchor are the MappedSpan fields (seekmer/_common.pxd:31-35);
 * nothing on the infer path reads them, so they are only written for batches
 * mapped after skm_mapper_keep_spans(mapper, 1) (SKM_ERR_STATE otherwise). */
int skm_mapper_last_batch(skm_mapper *mapper, int32_t *begin, int32_t *end,
                          int32_t *anchor_entry, int32_t *anchor_offset,
                          int32_t *counts, int32_t *entries,
                          int64_t cap_entries, int64_t *n_entries);
int skm_mapper_keep_spans(skm_mapper *mapper, int enable);
/* Counter sizes (MapResult.summarize, seekmer/mapper.py:77-104):
 * summary[0]=C classes [1]=M (class,target) rows [2]=unaligned [3]=total units */
int skm_mapper_summary(skm_mapper *mapper, int64_t summary[4]);
/* Classes in first-seen order (collections.Counter insertion order under -j1):
 * class_offsets[C+1], class_targets[M] unsigned ids in tuple order,
 * class_counts[C], first_seen[C] (global unit index), fld[2000]. */
int skm_mapper_export(skm_mapper *mapper, int64_t *class_offsets,
                      int32_t *class_targets, int64_t *class_counts,
                      int64_t *first_seen, int64_t *fld);
/* MapResult.merge_fragment_lengths / Counter.update with foreign data: merge
 * an exported table (e.g. another GPU's) into this one. */
int skm_mapper_merge(skm_mapper *mapper, int64_t n_classes,
                     const int64_t *class_offsets, const int32_t *class_targets,
                     const int64_t *class_counts, const int64_t *first_seen,
                     int64_t unaligned, const int64_t *fld);
/* The same hand-over without the host: a mapper's table where it lies in HBM --
 * classes in registry order (any order: the merge is by key, first-seen values travel with the
 * classes), class c = ids[class_start[c] .. + class_len[c]) (unsigned ids, tuple order), counts as
 * doubles, plus the unit totals and the histogram -- and its merge into a mapper ON THE SAME GPU.
 * Between GPUs the arrays are first copied over xGMI (ncclSend / ncclRecv or a peer copy: n_classes
 * elements of class_start / class_len / class_count / first_seen, n_ids of ids, 2000 of fld) and the
 * struct re-pointed at the copies.  The pointers of skm_mapper_device_table stay valid until the
 * mapper maps, merges, resets or is destroyed. */
typedef struct skm_device_table {
    int32_t device;
    int64_t n_classes, n_ids;
    const int64_t *class_start;      /* [n_classes] into ids */
    const int64_t *class_len;        /* [n_classes] */
    const double *class_count;       /* [n_classes] */
    const uint64_t *first_seen;      /* [n_classes] global unit index of the class's first unit */
    const int32_t *ids;              /* [n_ids] */
    int64_t unaligned, units, first_seen_bound;
    const uint64_t *fld;             /* [2000] */
} skm_device_table;
int skm_mapper_device_table(skm_mapper *mapper, skm_device_table *out);
int skm_mapper_merge_device(skm_mapper *mapper, const skm_device_table *table);
int skm_mapper_clear(skm_mapper *mapper);           /* MapResult.clear, mapper.py:143-145 */
/* Back to the state of a fresh MapResult (counter, unit count AND histogram
 * zeroed) while keeping every HBM buffer allocated. */
int skm_mapper_reset(skm_mapper *mapper);
/* stats[0]=pack kernel ns [1]=map kernel ns [2]=class kernels ns [3]=batches
 * [4]=units (HIP-event times accumulated over batches on the mapper stream)
 * [5]=EM ns [6]=EM steps of the skm_quant_infer calls made on this mapper */
int skm_mapper_timing(skm_mapper *mapper, double stats[8]);
/* Access counters of the instrumented build of the map kernel (they define the
 * algorithmic bytes): enable, map, then read.  out[0]=reads
 * [1]=read bases [2]=lookups [3]=slots probed [4]=ContigEntry reads [5]=target
 * entries copied [6]=target entries merged [7]=8-base fetches [8]=merges
 * [9]=tuple ids; out[16..30] = scheduler census of the map kernel: rounds, then
 * (chunk executions, lanes) of start, lookup, merge, left, right, emit, scan;
 * out[32..46] = wave-cycle sums: schedule, (unused), then per action, then six
 * phases of the emission.  enable = 2 selects the census build instead: the production code
 * paths with the census and cycle sums (a profiling aid; its access counters undercount). */
int skm_mapper_set_stats(skm_mapper *mapper, int enable);
int skm_mapper_access_stats(skm_mapper *mapper, int64_t out[48]);

/* ------------------------------------------------------------ quantification
 * MapResult.effective_lengths (seekmer/mapper.py:134-141). */
int skm_effective_lengths(int device, const int64_t *fld, const double *lengths,
                          int64_t n_tx, double *out);

/* Device-resident class table for infer.em / infer.quantify
 * (seekmer/infer.py:88-168).  class_counts are f8 as in
 * SummarizedResult.class_count. */
typedef struct skm_quant skm_quant;
int skm_quant_create(int device, int64_t n_tx, int64_t n_classes,
                     const int64_t *class_offsets, const int32_t *class_targets,
                     const double *class_counts, skm_quant **out);
/* Same, taking the table straight from a mapper without leaving HBM. */
int skm_quant_create_from_mapper(skm_mapper *mapper, int64_t n_tx, skm_quant **out);
int skm_quant_destroy(skm_quant *quant);
/* infer.em (seekmer/infer.py:133-168): x inout [n_tx], l = effective lengths.
 * Stops on the reference criterion max_{x'>x_floor} |x'-x|/x' <= rel_tol
 * (0.01, 1e-8 in the reference); fixed_iters>0 runs exactly that many steps
 * instead; max_iters>0 caps.  *iters = steps done.  Returns SKM_ERR_UNDEFINED
 * where numpy would raise (no x' > x_floor). */
int skm_quant_em(skm_quant *quant, double *x, const double *l, double rel_tol,
                 double x_floor, int64_t max_iters, int64_t fixed_iters,
                 int64_t *iters);
/* One bootstrap replicate set (seekmer/infer.py:79-82,108-111): n_boot
 * multinomial resamples of the class counts (counter-based RNG, `seed`), EM
 * from x0 each; out[n_boot][n_tx] raw EM results (before TPM scaling);
 * counts_out (optional) [n_boot][C] the resampled counts. */
int skm_quant_bootstrap(skm_quant *quant, int64_t n_boot, uint64_t seed,
                        const double *x0, const double *l, double rel_tol,
                        double x_floor, int64_t max_iters, double *out,
                        int64_t *counts_out, int64_t *iters_out);
/* The same with every replicate returned as the TPM vector quantify() makes of it
 * (seekmer/infer.py:127-129: x /= x.sum() / 1e6; x[x < 0.001] = 0; again), numpy's summation
 * order restated on the device -- what run() appends to its bootstrap list (infer.py:79-82). */
int skm_quant_bootstrap_tpm(skm_quant *quant, int64_t n_boot, uint64_t seed,
                            const double *x0, const double *l, double rel_tol,
                            double x_floor, int64_t max_iters, double *out,
                            int64_t *iters_out);
/* One rank's share of the `-b N` loop: the n_boot replicates numbered first,
 * first + step, first + 2 step, ... of the run; out[n_boot][n_tx] as skm_quant_bootstrap_tpm.  A
 * replicate's draw depends on (seed, its number) alone, so ranks r = 0 .. G-1 calling with
 * (first = r, step = G) on the same merged table produce, together, exactly the replicates of
 * skm_quant_bootstrap_tpm(N) -- no collective, one gather of the results. */
int skm_quant_bootstrap_share_tpm(skm_quant *quant, int64_t n_boot, int64_t first, int64_t step,
                                  uint64_t seed, const double *x0, const double *l, double rel_tol,
                                  double x_floor, int64_t max_iters, double *out,
                                  int64_t *iters_out);
/* EM with externally supplied class counts (parity of the bootstrap EM leg). */
int skm_quant_set_counts(skm_quant *quant, const double *class_counts);
/* timing[0]=EM kernel ns total [1]=iterations [2]=launches */
int skm_quant_timing(skm_quant *quant, double timing[4]);

/* ---------------------------------------------------------------- multi-GPU
 * Not in the reference (single process).  Reads shard across ranks; the only
 * data-path collective is one all-reduce(sum) of f64[n_tx] per EM step (RCCL
 * over xGMI), issued on the EM stream between the class pass and the
 * finalise pass.  A communicator is created once per process and attached to
 * every quant handle that should take part. */
typedef struct skm_comm skm_comm;
int skm_comm_unique_id(void *id128);                 /* rank 0: 128-byte id, single use */
int skm_comm_create(int device, const void *id128, int rank, int world, skm_comm **out);
int skm_comm_count(skm_comm *comm, int *count);     /* ncclCommCount: ranks RCCL itself sees */
int skm_comm_destroy(skm_comm *comm);
/* The table hand-over of SURVEY.md 8(e).1 over xGMI: `send`'s table goes to rank send_to as it lies
 * in HBM (ncclSend of skm_mapper_device_table's arrays), the table rank recv_from sends is received
 * into HBM and merged into `recv` by key (skm_mapper_merge_device) -- no host copy, no host sort.
 * Either side may be NULL.  Matching calls: rank r: (mapper, 0, NULL, -1), rank 0: (NULL, -1,
 * mapper, r) for every r > 0 in turn.  Experimental: no multi-GPU node has run it; a rank sending
 * to itself is what the one-GPU test exercises. */
int skm_mapper_exchange_tables(skm_mapper *send, int send_to, skm_mapper *recv, int recv_from, skm_comm *comm);
int skm_quant_set_comm(skm_quant *quant, skm_comm *comm);   /* NULL detaches */

/* One sample, mapper table -> TPM, without leaving the device: MapResult.effective_lengths
 * (seekmer/mapper.py:134-141, the histogram all-reduced over `comm`'s ranks when comm is
 * not NULL) + quantify() (seekmer/infer.py:88-130: start vector 1/l normalised with
 * numpy's sum, em(), TPM scaling).  lengths: host f8[n_tx] transcript lengths.  Host
 * outputs (each may be NULL): tpm[n_tx], effective_lengths[n_tx], *iters. */
int skm_quant_infer(skm_mapper *mapper, skm_comm *comm, const double *lengths, int64_t n_tx,
                    double rel_tol, double x_floor, int64_t max_iters,
                    double *tpm, double *effective_lengths, int64_t *iters);

/* ============================ libseekmer_host.so ========================== */

/* ContigAssembler.assemble (seekmer/_index_builder.pyx:105-150): pooled
 * transcript bases, transcript i = [seq_offsets[i], seq_offsets[i+1]). */
typedef struct skm_built skm_built;
int skm_build_index(const char *pool, const int64_t *seq_offsets, int64_t n_seqs,
                    int n_threads, skm_built **out);
/* sizes[0]=n_slots [1]=n_contigs [2]=n_bases [3]=n_targets */
int skm_built_sizes(const skm_built *b, int64_t sizes[4]);
int skm_built_copy(const skm_built *b, void *kmers, void *contigs,
                   char *sequences, void *targets);
int skm_built_free(skm_built *b);

/* feed_single_ended_reads / feed_pair_ended_reads (seekmer/common.py:126-197)
 * as a native batch packer over already-decompressed FASTQ text. */
typedef struct skm_fastq skm_fastq;
int skm_fastq_open(const char *const *paths, int n_paths, int paired,
                   int64_t batch_units, skm_fastq **out);
/* Before the first skm_fastq_next.  set_allocator: where slabs live (default malloc/free; with
 * skm_pinned_alloc/skm_pinned_free a batch crosses PCIe by DMA straight from its slab).
 * set_parallel: n_threads > 0 asks for the parallel engine -- the files are memory-mapped, their
 * newlines counted once, and whole batches are parsed side by side by n_threads workers and
 * handed out in file order; same batches as the sequential engine.  It needs plain files whose
 * line counts are multiples of four; *enabled = 0 (and the sequential engine stays) otherwise. */
int skm_fastq_set_allocator(skm_fastq *reader, void *(*alloc)(size_t), void (*release)(void *));
int skm_fastq_set_parallel(skm_fastq *reader, int n_threads, int *enabled);
/* Several readers, one sample (one rank per GPU): reader `rank` of `world` hands out the
 * batches k with k % world == rank (the parallel engine parses only those; the sequential
 * engine has to read past the others).  skm_fastq_batch_index: k of the batch handed out last
 * -- its first unit is k * batch_units of the whole sample. */
int skm_fastq_set_shard(skm_fastq *reader, int rank, int world);
int skm_fastq_batch_index(const skm_fastq *reader, int64_t *index);
/* the common length of the reads of the batch handed out last, or -1 when they differ
 * (skm_mapper_map_batch_uniform_async takes a batch of equal-length reads without offsets) */
int skm_fastq_batch_read_length(const skm_fastq *reader, int64_t *read_len);
/* next batch: *n_units = 0 at end.  Buffers are owned by the reader and stay
 * valid until the next call.  names: '\n'-separated. */
int skm_fastq_next(skm_fastq *reader, int64_t *n_units, const char **bases,
                   const int64_t **offsets, const char **names,
                   const int64_t **name_offsets);
/* Keep the arrays of the last skm_fastq_next alive past the next call: the
 * reader hands their storage over as *slab and fills another one from now on.
 * skm_fastq_recycle gives it back for re-use (reader NULL: just frees it) --
 * a large batch costs more in first-touch page faults than in parsing. */
typedef struct skm_fastq_slab skm_fastq_slab;
int skm_fastq_detach(skm_fastq *reader, skm_fastq_slab **slab);
int skm_fastq_recycle(skm_fastq *reader, skm_fastq_slab *slab);
int skm_fastq_close(skm_fastq *reader);

/* Bytes of file mappings nobody has open that the readers may keep for the next reader of the same
 * file (default 0: a mapping goes when its last reader closes; a caller that reads the same files
 * again and again -- a benchmark's passes -- saves the page-table set-up of the later passes). */
int skm_fastq_cache_bytes(int64_t bytes);
/* The page tables of the input ahead of its reader: map the files now and touch their pages from
 * n_threads helper threads -- for a run that knows its FASTQ files while it is still loading the index
 * (setting up the page tables of the text costs as much as parsing it).  A reader opened before
 * _finish finds the mappings in place; _finish stops the helpers and drops what nobody has open. */
/* File mappings that no reader uses any more (beyond skm_fastq_cache_bytes) are unmapped by a
 * background thread in small pieces.  _wait_unmapped returns once none of that is under way: for a
 * caller about to time a phase that faults pages or allocates -- both wait for the process's
 * memory-map lock, which the teardown takes over and over. */
int skm_fastq_wait_unmapped(void);
typedef struct skm_fastq_prefault skm_fastq_prefault;
int skm_fastq_prefault_start(const char *const *paths, int n_paths, int n_threads, skm_fastq_prefault **out);
int skm_fastq_prefault_finish(skm_fastq_prefault *handle);

/* ---- FASTQ text -> 2-bit reads on the host (skm_packed_reads, above) ------------------------- */
/* feed_single_ended_reads / feed_pair_ended_reads (seekmer/common.py:126-197) for plain (not
 * compressed) files, straight to packed pieces in ONE pass over the text: the files are cut into
 * `chunk_bytes` ranges that n_threads workers parse side by side, each from the first place in its
 * range that looks like a record start; the reader hands the pieces out in file order and checks,
 * piece by piece, that the walk before ended exactly where this one started -- the reference's
 * rule is purely line-number based (`i & 3`), and only that chain proves a guessed start to be a
 * line with i & 3 == 0.  A piece whose guess fails is parsed again from the proven place (one
 * thread), so the reads are the reference's for any text.  Paired input is two streams, mate 1
 * files and mate 2 files, each numbered by unit; a pair of files counts min(records) units
 * (zip(file1, file2)), and a piece whose first_read lies below what its stream already delivered
 * replaces the reads from there on.  A trailing name line without a bases line is dropped.
 * n_threads = 0 parses inside skm_fastq_packed_next.  SKM_ERR_IO: not regular files (pipes from
 * decompressors go through skm_fastq_open). */
typedef struct skm_fastq_packed skm_fastq_packed;
int skm_fastq_packed_open(const char *const *paths, int n_paths, int paired, int n_threads,
                          int64_t chunk_bytes, int want_names, skm_fastq_packed **out);
/* The same reader over a SHARE of the files, for a sample shared out over ranks
 * (NativeReadFeeder(shard=...) does this for the two-pass reader; seekmer/common.py:126-197 is one
 * process): file i is read from byte begin[i] to byte end[i], both starts of lines whose number is
 * a multiple of four (skm_fastq_locate_line), and the first unit carries the number first_unit --
 * pieces are numbered as the one-process reader numbers them.  A pair of files must hold the same
 * number of records in its two ranges. */
int skm_fastq_packed_open_ranges(const char *const *paths, int n_paths, int paired, int n_threads,
                                 int64_t chunk_bytes, int want_names, const int64_t *begin,
                                 const int64_t *end, int64_t first_unit, skm_fastq_packed **out);
/* Where the lines of a file start, without parsing it: the reference's records are lines 4u .. 4u + 3
 * of a file whatever they hold, so the place of unit u takes the number of newlines before it.
 * _count_newlines fills counts[k] for the chunks k = first_chunk, first_chunk + chunk_step, ... of
 * `chunk_bytes` bytes (n_chunks = ceil(file size / chunk_bytes), SKM_ERR_ARG otherwise; the other
 * entries are left alone: ranks count a share each and add the tables up), *last_line_open = the
 * file does not end in a newline (its last line counts all the same).  _locate_line: the byte at
 * which line `line` (from 0) starts given the counts of ALL chunks -- one chunk is walked --, the
 * file's size when it has fewer lines; SKM_ERR_STATE when the counts are not this file's. */
int skm_fastq_count_newlines(const char *path, int64_t chunk_bytes, int64_t first_chunk,
                             int64_t chunk_step, int n_threads, int64_t *counts, int64_t n_chunks,
                             int *last_line_open);
int skm_fastq_locate_line(const char *path, int64_t chunk_bytes, const int64_t *counts,
                          int64_t n_chunks, int64_t line, int64_t *byte_offset);
/* where the arrays that cross PCIe live (before the first _next; see skm_fastq_set_allocator) */
int skm_fastq_packed_set_allocator(skm_fastq_packed *reader, void *(*alloc)(size_t),
                                   void (*release)(void *));
/* next piece; piece->n_reads == 0 (and code_words != SKM_PACKED_CUT) at the end.  The arrays stay
 * valid through ONE more call (skm_packed_source's contract: the mapper's drain copies piece k to
 * the GPU while it asks for piece k + 1).  The signature is skm_packed_source (below) with the
 * reader as context. */
int skm_fastq_packed_next(void *reader, skm_packed_reads *piece);
/* stats[0]=pieces accepted as guessed [1]=pieces parsed again [2]=reads [3]=exceptions
 * [4]=parser variant in use (0 single characters, 1 16-byte blocks, 2 32-byte blocks)
 * [5]=units of the sample so far (paired: min over the two streams, per pair of files) */
int skm_fastq_packed_stats(const skm_fastq_packed *reader, int64_t stats[8]);
/* About how many units the files hold (each mate-1 / single-end file's size over the extent of its
 * first record): the hint for skm_mapper_expect_units. */
int skm_fastq_packed_estimate(const skm_fastq_packed *reader, int64_t *units);
int skm_fastq_packed_close(skm_fastq_packed *reader);
/* The same packing for reads that are already in memory (bases back to back + offsets, the layout
 * of skm_mapper_map_batch): codes[n_reads][code_words], lengths[n_reads]; exception arrays hold up
 * to cap_exceptions entries, *n_exceptions = how many there are (SKM_ERR_STATE when more than the
 * capacity, SKM_ERR_ARG when a read is longer than 32 * code_words).  variant: -1 best available,
 * or 0 / 1 / 2 as in skm_fastq_packed_stats (SKM_ERR_STATE when this CPU lacks it). */
int skm_pack_reads(const char *bases, const int64_t *offsets, int64_t n_reads, int32_t code_words,
                   uint64_t *codes, uint32_t *lengths, uint32_t *exception_reads,
                   uint32_t *exception_masks, int64_t cap_exceptions, int64_t *n_exceptions,
                   int variant);
/* force the parser variant of readers opened from now on (-1: best available); tests */
int skm_pack_set_variant(int variant);

/* Seeded synthetic data; integer arithmetic only. */
int skm_synth_transcriptome(uint64_t seed, int64_t n_genes, int64_t *n_tx,
                            char **pool, int64_t **offsets);
int skm_synth_free(void *p);
int skm_synth_reads(uint64_t seed, const char *pool, const int64_t *tx_offsets,
                    int64_t n_tx, int64_t first_unit, int64_t n_units,
                    int read_len, int paired, int n_threads, char *bases);

/* The same reads as FASTQ text (names r<ten digits>/<mate>, constant qualities), for measuring
 * the path from files: bases = [n_units][mates][read_len] as skm_synth_reads fills it. */
int skm_synth_fastq_write(const char *bases, int64_t n_units, int read_len, int paired,
                          int64_t first_unit, const char *path1, const char *path2,
                          int n_threads);

#ifdef __cplusplus
}
#endif
#endif
