/* CPU ORACLE (test infrastructure only) -- index queries.
 * Restates the query half of seekmer/_common.pyx (KMerIndex). */
#include "skmo.h"

static skmo_coord invalid_coord(void)
{
    skmo_coord c; c.entry = 0; c.offset = -1;   /* seekmer/_coordinate.pxd:13-24 */
    return c;
}

static skmo_coord rc_coord(skmo_coord c)
{
    c.entry = ~c.entry;                          /* seekmer/_coordinate.pxd:66-81 */
    return c;
}

/* seekmer/_common.pyx:54-97 -- home slot = hash(min(kmer, rc)) & (size-1);
 * linear probe offset..size-1 then 0..offset-1; per slot: empty -> miss,
 * == kmer -> position, == rc -> position with ~entry. */
skmo_coord skmo_map_kmer(const skmo_index *ix, uint64_t kmer, skmo_stats *st)
{
    uint64_t rc_kmer = skmo_kmer_reverse_complement(kmer);
    int size = (int)ix->n_kmers;
    int offset = skmo_kmer_hash(kmer < rc_kmer ? kmer : rc_kmer) & (size - 1);
    if (st) st->lookups++;
    for (int i = offset; i < size; ++i) {
        if (st) st->slots++;
        if (ix->kmers[i].kmer == SKMO_INVALID_KMER) return invalid_coord();
        if (ix->kmers[i].kmer == kmer) return ix->kmers[i].position;
        if (ix->kmers[i].kmer == rc_kmer) return rc_coord(ix->kmers[i].position);
    }
    for (int i = 0; i < offset; ++i) {
        if (st) st->slots++;
        if (ix->kmers[i].kmer == SKMO_INVALID_KMER) return invalid_coord();
        if (ix->kmers[i].kmer == kmer) return ix->kmers[i].position;
        if (ix->kmers[i].kmer == rc_kmer) return rc_coord(ix->kmers[i].position);
    }
    return invalid_coord();
}

/* seekmer/_common.pyx:103-137 -- `length` > 0: leading |length| bases of the
 * anchored k-mer in read orientation; < 0: trailing ones.  All index
 * arithmetic in C int as in the reference.  `out` needs |length| bytes. */
void skmo_get_contig_sequence(const skmo_index *ix, skmo_coord c, int length,
                              char *out, skmo_stats *st)
{
    int index = c.entry;
    if (index < 0) index = ~index;
    int offset = (int)(ix->contigs[index].offset + c.offset);
    if (st) { st->contig_reads++; st->seq_fetches++; }
    if (c.entry >= 0)
        offset += length > 0 ? length : SKMO_K;
    else
        offset += length > 0 ? SKMO_K : -length;
    if (length < 0) length = -length;
    for (int i = offset - length; i < offset; ++i)
        out[i - offset + length] = ix->sequences[i];
    if (c.entry < 0) skmo_sequence_reverse_complement(out, length);
}

/* seekmer/_common.pyx:241-266 */
uint64_t skmo_get_tail_kmer(const skmo_index *ix, skmo_coord c, skmo_stats *st)
{
    int index = c.entry;
    if (index < 0) index = ~index;
    uint64_t kmer;
    if (st) st->contig_reads++;
    if (c.offset == 0) kmer = ix->contigs[index].first_kmer;
    else kmer = ix->contigs[index].last_kmer;
    if (c.entry < 0) kmer = skmo_kmer_reverse_complement(kmer);
    return kmer;
}
