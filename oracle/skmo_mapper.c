/* CPU ORACLE (test infrastructure only) -- the per-read state machine.
 * Restates seekmer/_mapper.pyx and the span-level half of seekmer/_common.pyx. */
#include "skmo.h"
#include <stdlib.h>
#include <string.h>

static skmo_coord invalid_coord(void)
{
    skmo_coord c; c.entry = 0; c.offset = -1;
    return c;
}

static void free_targets(skmo_span *s)        /* seekmer/_coordinate_array.pxd:64-66 */
{
    free(s->items);
    s->items = NULL;
    s->n = 0;
}

/* seekmer/_common.pyx:143-179 -- copy of the contig's target slice; reverse
 * hit: reversed order, every entry complemented, offsets kept. */
static void map_contig(const skmo_index *ix, skmo_coord c, skmo_span *span, skmo_stats *st)
{
    int index = c.entry;
    int forward = index >= 0;
    if (!forward) index = ~index;
    int start = (int)ix->contigs[index].target_offset;
    int length = (int)ix->contigs[index].target_length;
    if (st) { st->contig_reads++; st->targets_copied += length; }
    span->n = length;
    span->items = (skmo_coord *)malloc(sizeof(skmo_coord) * (size_t)(length > 0 ? length : 1));
    if (forward) {
        for (int i = 0, j = start; j < start + length; ++i, ++j)
            span->items[i] = ix->targets[j];
    } else {
        for (int i = 0, j = start + length - 1; j > start - 1; ++i, --j) {
            span->items[i] = ix->targets[j];
            span->items[i].entry = ~span->items[i].entry;
        }
    }
}

/* seekmer/_common.pyx:185-235 -- two-pointer merge on the signed entry; empty
 * intersection leaves the targets intact and returns 0; empty input returns 1. */
static int filter_on_contig(const skmo_index *ix, skmo_span *span, skmo_stats *st)
{
    if (st) st->merges++;
    if (span->n == 0) return 1;
    int contig_id = span->anchor.entry;
    int forward = contig_id >= 0;
    if (!forward) contig_id = ~contig_id;
    int start = (int)ix->contigs[contig_id].target_offset;
    int length = (int)ix->contigs[contig_id].target_length;
    if (st) st->contig_reads++;
    int read_index = 0, write_index = 0;
    int track_index = forward ? start : start + length - 1;
    int track_bound = forward ? start + length : start - 1;
    int step = forward ? 1 : -1;
    int first_track = track_index;
    while (read_index != span->n && track_index != track_bound) {
        int target_entry = span->items[read_index].entry;
        int index_entry = ix->targets[track_index].entry;
        if (!forward) index_entry = ~index_entry;
        if (target_entry == index_entry) {
            span->items[write_index] = span->items[read_index];
            read_index++; write_index++; track_index += step;
        } else if (target_entry < index_entry) {
            read_index++;
        } else {
            track_index += step;
        }
    }
    if (st) {
        /* index-side entries consumed: every entry compared at least once */
        int64_t consumed = (int64_t)(track_index - first_track) * step;
        if (track_index != track_bound) consumed += 1;
        st->targets_merged += consumed;
    }
    if (write_index == 0) return 0;
    span->n = write_index;
    return 1;
}

/* seekmer/_mapper.pyx:500-501 -- read characters outside upper-case ACGT match anything */
static int match_base(int reference, int query)
{
    return reference == query
        || !(query == 'A' || query == 'C' || query == 'G' || query == 'T');
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* seekmer/_mapper.pyx:404-445 -- note the query cursor starts one base short
 * of the reference cursor (lines 406-408). */
int skmo_sift4_align_left(const char *ref, int ref_len, const char *query,
                          int query_len, int offset)
{
    (void)query_len;
    int reference_cursor = ref_len - 1;
    int query_cursor = offset + SKMO_ALIGN_LENGTH - 1;
    query_cursor -= 1;
    int distance = 0;
    while (reference_cursor >= 0 && query_cursor >= offset) {
        if (match_base(ref[reference_cursor], query[query_cursor])) {
            reference_cursor -= 1;
            query_cursor -= 1;
            continue;
        }
        if (reference_cursor != query_cursor - offset) {
            reference_cursor = imin(query_cursor - offset, reference_cursor);
            query_cursor = reference_cursor + offset;
        }
        for (int i = 0; i < SKMO_MAX_OFFSET; ++i) {
            if (query_cursor - i >= offset - 1
                    && query_cursor - i >= 0
                    && match_base(ref[reference_cursor], query[query_cursor - i])) {
                distance += i - 1;
                query_cursor -= i - 1;
                reference_cursor += 1;
                break;
            }
            if (reference_cursor - i >= 0
                    && match_base(ref[reference_cursor - i], query[query_cursor])) {
                distance += i - 1;
                query_cursor += 1;
                reference_cursor -= i - 1;
                break;
            }
        }
        distance += 1;
        query_cursor -= 1;
        reference_cursor -= 1;
        if (distance > SKMO_MAX_DISTANCE) return SKMO_INVALID_SHIFT;
    }
    if (reference_cursor >= 0) return reference_cursor + 1;
    if (query_cursor >= offset) return -1 - query_cursor + offset;
    return 0;
}

/* seekmer/_mapper.pyx:452-493 */
int skmo_sift4_align_right(const char *ref, int ref_len, const char *query,
                           int query_len, int offset)
{
    int reference_cursor = 0;
    int query_cursor = offset;
    int distance = 0;
    while (reference_cursor < ref_len && query_cursor < offset + SKMO_ALIGN_LENGTH) {
        if (match_base(ref[reference_cursor], query[query_cursor])) {
            reference_cursor += 1;
            query_cursor += 1;
            continue;
        }
        if (reference_cursor != query_cursor - offset) {
            reference_cursor = imax(query_cursor - offset, reference_cursor);
            query_cursor = reference_cursor + offset;
        }
        for (int i = 0; i < SKMO_MAX_OFFSET; ++i) {
            if (query_cursor + i < offset + SKMO_ALIGN_LENGTH + 1
                    && query_cursor + i < query_len
                    && match_base(ref[reference_cursor], query[query_cursor + i])) {
                distance += i - 1;
                query_cursor += i - 1;
                reference_cursor -= 1;
                break;
            }
            if (reference_cursor + i < ref_len
                    && match_base(ref[reference_cursor + i], query[query_cursor])) {
                distance += i - 1;
                query_cursor -= 1;
                reference_cursor += i - 1;
                break;
            }
        }
        distance += 1;
        query_cursor += 1;
        reference_cursor += 1;
        if (distance > SKMO_MAX_DISTANCE) return SKMO_INVALID_SHIFT;
    }
    if (reference_cursor < ref_len) return ref_len - reference_cursor;
    if (query_cursor < offset + SKMO_ALIGN_LENGTH)
        return query_cursor - offset - SKMO_ALIGN_LENGTH;
    return 0;
}

/* seekmer/_mapper.pyx:199-216 */
static void find_first_kmer(const skmo_index *ix, const char *bases, int length,
                            skmo_span *span, skmo_stats *st)
{
    uint64_t kmer = skmo_kmer_encode(bases, span->begin);
    span->anchor = skmo_map_kmer(ix, kmer, st);
    if (span->anchor.offset >= 0) {
        span->end = span->begin;
        map_contig(ix, span->anchor, span, st);
        return;
    }
    for (int i = span->begin + SKMO_K; i < length; ++i) {
        kmer = skmo_kmer_append(kmer, bases[i]);
        span->anchor = skmo_map_kmer(ix, kmer, st);
        if (span->anchor.offset < 0) continue;
        span->begin = i + 1 - SKMO_K;
        span->end = span->begin;
        map_contig(ix, span->anchor, span, st);
        return;
    }
}

/* seekmer/_mapper.pyx:222-275 */
static void filter_targets_to_left(const skmo_index *ix, const char *bases, int length,
                                   skmo_span *span, skmo_stats *st)
{
    uint64_t kmer;
    int forward = span->anchor.entry >= 0;
    int32_t contig_index = forward ? span->anchor.entry : ~span->anchor.entry;
    int contig_length = (int)ix->contigs[contig_index].length;
    if (st) st->contig_reads++;
    int move = forward ? span->anchor.offset
                       : contig_length - span->anchor.offset - SKMO_K;
    char contig[SKMO_ALIGN_LENGTH + 1];
    int shift;
    while (span->begin > move) {
        span->begin -= move;
        span->anchor.offset -= forward ? move : -move;
        skmo_get_contig_sequence(ix, span->anchor, SKMO_ALIGN_LENGTH, contig, st);
        shift = skmo_sift4_align_left(contig, SKMO_ALIGN_LENGTH, bases, length, span->begin);
        if (shift == SKMO_INVALID_SHIFT || shift + 1 + move <= 0) {
            free_targets(span);
            return;
        }
        span->begin -= shift + 1;
        if (span->begin < 0) {
            span->begin = 0;
            return;
        }
        kmer = skmo_kmer_prepend(skmo_get_tail_kmer(ix, span->anchor, st),
                                 bases[span->begin]);
        span->anchor = skmo_map_kmer(ix, kmer, st);
        if (!(span->anchor.offset >= 0) || !filter_on_contig(ix, span, st)) {
            if (span->begin < SKMO_K) {
                span->begin = 0;
                return;
            }
            span->begin -= SKMO_K;
            kmer = skmo_kmer_encode(bases, span->begin);
            span->anchor = skmo_map_kmer(ix, kmer, st);
            if (!(span->anchor.offset >= 0) || !filter_on_contig(ix, span, st)) {
                free_targets(span);
                return;
            }
        }
        forward = span->anchor.entry >= 0;
        contig_index = forward ? span->anchor.entry : ~span->anchor.entry;
        contig_length = (int)ix->contigs[contig_index].length;
        if (st) st->contig_reads++;
        move = forward ? span->anchor.offset
                       : contig_length - span->anchor.offset - SKMO_K;
    }
    span->anchor.offset -= forward ? span->begin : -span->begin;
    skmo_get_contig_sequence(ix, span->anchor, SKMO_ALIGN_LENGTH, contig, st);
    shift = skmo_sift4_align_left(contig, SKMO_ALIGN_LENGTH, bases, length, 0);
    if (shift == SKMO_INVALID_SHIFT) free_targets(span);
}

/* seekmer/_mapper.pyx:281-343 -- lines 316-329 of the reference are dead code
 * (the identical test at 312-315 has already returned) and are not restated. */
static void filter_targets_to_right(const skmo_index *ix, const char *bases, int length,
                                    skmo_span *span, skmo_stats *st)
{
    uint64_t kmer = skmo_kmer_encode(bases, span->end);
    span->anchor = skmo_map_kmer(ix, kmer, st);
    int forward = span->anchor.entry >= 0;
    int32_t contig_index = forward ? span->anchor.entry : ~span->anchor.entry;
    int contig_length = (int)ix->contigs[contig_index].length;
    if (st) st->contig_reads++;
    int move = forward ? contig_length - span->anchor.offset - SKMO_K
                       : span->anchor.offset;
    char contig[SKMO_ALIGN_LENGTH + 1];
    int shift;
    while (length - span->end - SKMO_K > move) {
        span->end += move;
        span->anchor.offset += forward ? move : -move;
        skmo_get_contig_sequence(ix, span->anchor, -SKMO_ALIGN_LENGTH, contig, st);
        shift = skmo_sift4_align_right(contig, SKMO_ALIGN_LENGTH, bases, length,
                                       span->end + SKMO_K - SKMO_ALIGN_LENGTH);
        if (shift == SKMO_INVALID_SHIFT || shift + 1 + move <= 0) {
            free_targets(span);
            return;
        }
        span->end += shift + 1;
        if (span->end + SKMO_K > length) {
            span->end = length - SKMO_K;
            return;
        }
        kmer = skmo_kmer_append(skmo_get_tail_kmer(ix, span->anchor, st),
                                bases[span->end + SKMO_K - 1]);
        span->anchor = skmo_map_kmer(ix, kmer, st);
        if (!(span->anchor.offset >= 0) || !filter_on_contig(ix, span, st)) {
            free_targets(span);
            return;
        }
        forward = span->anchor.entry >= 0;
        contig_index = forward ? span->anchor.entry : ~span->anchor.entry;
        contig_length = (int)ix->contigs[contig_index].length;
        if (st) st->contig_reads++;
        move = forward ? contig_length - span->anchor.offset - SKMO_K
                       : span->anchor.offset;
    }
    if (forward)
        span->anchor.offset += length - span->end - SKMO_K;
    else
        span->anchor.offset -= length - span->end - SKMO_K;
    skmo_get_contig_sequence(ix, span->anchor, -SKMO_ALIGN_LENGTH, contig, st);
    shift = skmo_sift4_align_right(contig, SKMO_ALIGN_LENGTH, bases, length,
                                   length - SKMO_ALIGN_LENGTH);
    if (shift == SKMO_INVALID_SHIFT) free_targets(span);
}

/* seekmer/_mapper.pyx:151-193.  Reads shorter than k are undefined behaviour in
 * the reference (encode reads past the buffer); the oracle defines them as
 * unmapped with the initial span (documented deviation). */
skmo_span skmo_map_read(const skmo_index *ix, const char *bases, int length,
                        skmo_stats *st)
{
    skmo_span span;
    span.anchor = invalid_coord();
    span.begin = 0;
    span.end = span.begin;
    span.n = 0;
    span.items = NULL;
    if (st) { st->reads++; st->read_bases += length; }
    if (length < SKMO_K) return span;
    find_first_kmer(ix, bases, length, &span, st);
    if (span.n == 0) return span;
    if (span.begin > 0) filter_targets_to_left(ix, bases, length, &span, st);
    if (span.n != 0 && span.end < length - SKMO_K)
        filter_targets_to_right(ix, bases, length, &span, st);
    if (span.n != 0) return span;
    span.anchor = invalid_coord();
    span.n = 0; span.items = NULL;
    span.begin += SKMO_K;
    if (span.begin + SKMO_K > length) span.begin = length - SKMO_K;
    span.end = span.begin;
    find_first_kmer(ix, bases, length, &span, st);
    if (span.n == 0) return span;
    if (span.begin > 0) filter_targets_to_left(ix, bases, length, &span, st);
    if (span.n != 0 && span.end < length - SKMO_K)
        filter_targets_to_right(ix, bases, length, &span, st);
    return span;
}

/* seekmer/_mapper.pyx:350-397 -- returns 1 when mate 1 is empty, 0 when only
 * mate 2 is empty; match test entry1 == ~entry2 with mate 2 walked backwards. */
static int intersect(skmo_span *r1, skmo_span *r2)
{
    if (r1->n == 0) return 1;
    if (r2->n == 0) return 0;
    int cursor1_read = 0, cursor1_write = 0;
    int cursor2 = r2->n - 1;
    while (cursor1_read != r1->n && cursor2 != -1) {
        int entry1 = r1->items[cursor1_read].entry;
        int entry2 = ~r2->items[cursor2].entry;
        if (entry1 == entry2) {
            r1->items[cursor1_write] = r1->items[cursor1_read];
            cursor1_read++; cursor1_write++; cursor2--;
        } else if (entry1 < entry2) {
            cursor1_read++;
        } else {
            cursor2--;
        }
    }
    if (cursor1_write == 0) return 0;
    r1->n = cursor1_write;
    return 1;
}

/* seekmer/_mapper.pyx:111-145 */
skmo_span skmo_map_read_pair(const skmo_index *ix, const char *b1, int l1,
                             const char *b2, int l2, skmo_stats *st)
{
    skmo_span span1 = skmo_map_read(ix, b1, l1, st);
    skmo_span span2 = skmo_map_read(ix, b2, l2, st);
    int interval = 0;
    if (!intersect(&span1, &span2)) {
        free_targets(&span1);
        span1.begin = 0;
        span1.end = span1.begin - SKMO_K;
    } else if (span1.anchor.entry != ~span2.anchor.entry) {
        span1.begin = 0;
        span1.end = span1.begin - SKMO_K;
    } else {
        span1.end = l1 - SKMO_K;
        span2.end = l2 - SKMO_K;
        interval = span2.anchor.offset - span1.anchor.offset;
        if (span1.anchor.entry < 0) interval = -interval;
        span1.end += interval + span2.end - span2.begin;
    }
    free_targets(&span2);
    return span1;
}

/* seekmer/_mapper.pyx:59-105 (batch loop incl. the FLD rule at 90-94) */
int64_t skmo_map_batch(const skmo_index *ix, const char *bases,
                       const int64_t *offsets, int64_t n_units, int paired,
                       int32_t *out_begin, int32_t *out_end,
                       int32_t *out_anchor_entry, int32_t *out_anchor_offset,
                       int32_t *out_count, int32_t *out_entries,
                       int64_t cap_entries, int64_t *fld, skmo_stats *st)
{
    if (!ix || !bases || !offsets || n_units < 0) return -1;
    int64_t total = 0;
    int64_t j = 0;
    for (int64_t i = 0; i < n_units; ++i) {
        skmo_span span;
        if (!paired) {
            span = skmo_map_read(ix, bases + offsets[j], (int)(offsets[j + 1] - offsets[j]), st);
            j += 1;
        } else {
            span = skmo_map_read_pair(ix, bases + offsets[j], (int)(offsets[j + 1] - offsets[j]),
                                      bases + offsets[j + 1], (int)(offsets[j + 2] - offsets[j + 1]), st);
            j += 2;
        }
        int length = span.end - span.begin + SKMO_K;
        if (length > 0 && fld) {
            if (length >= SKMO_MAX_FRAGMENT_LENGTH) length = SKMO_MAX_FRAGMENT_LENGTH - 1;
            fld[length] += 1;
        }
        if (out_begin) out_begin[i] = span.begin;
        if (out_end) out_end[i] = span.end;
        if (out_anchor_entry) out_anchor_entry[i] = span.anchor.entry;
        if (out_anchor_offset) out_anchor_offset[i] = span.anchor.offset;
        if (out_count) out_count[i] = span.n;
        for (int t = 0; t < span.n; ++t) {
            if (out_entries && total + t < cap_entries)
                out_entries[total + t] = span.items[t].entry;
        }
        total += span.n;
        if (st) st->tuple_ids += span.n;
        free(span.items);
    }
    return total;
}
