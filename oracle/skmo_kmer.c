/* CPU ORACLE (test infrastructure only) -- k-mer, hash and sequence primitives.
 * Restates seekmer/_kmer.pxd and seekmer/_sequence.pxd. */
#include "skmo.h"

/* seekmer/_kmer.pxd:31-39 -- ~(invalid << 50) */
uint64_t skmo_kmer_mask(void) { return ~(SKMO_INVALID_KMER << (SKMO_K * 2)); }

/* seekmer/_kmer.pxd:253-273 -- T/t=3, G/g=2, C/c=1, anything else 0 */
uint64_t skmo_two_bit_encode(char base)
{
    if (base == 'T' || base == 't') return 3;
    if (base == 'G' || base == 'g') return 2;
    if (base == 'C' || base == 'c') return 1;
    return 0;
}

/* seekmer/_kmer.pxd:46-68 -- no bounds check in the reference either */
uint64_t skmo_kmer_encode(const char *sequence, int offset)
{
    uint64_t kmer = 0;
    for (int i = offset; i < offset + SKMO_K; ++i) {
        kmer <<= 2;
        kmer |= skmo_two_bit_encode(sequence[i]);
    }
    return kmer;
}

/* seekmer/_kmer.pxd:71-87 */
uint64_t skmo_kmer_append(uint64_t kmer, char base)
{
    return ((kmer << 2) | skmo_two_bit_encode(base)) & skmo_kmer_mask();
}

/* seekmer/_kmer.pxd:90-106 */
uint64_t skmo_kmer_prepend(uint64_t kmer, char base)
{
    return (kmer >> 2) | (skmo_two_bit_encode(base) << (SKMO_K * 2 - 2));
}

/* seekmer/_kmer.pxd:113-143 */
void skmo_kmer_decode(uint64_t kmer, char *out25)
{
    static const char alphabet[4] = { 'A', 'C', 'G', 'T' };
    for (int i = SKMO_K - 1; i >= 0; --i) {
        out25[i] = alphabet[kmer & 3];
        kmer >>= 2;
    }
}

/* seekmer/_kmer.pxd:146-171 -- pairwise bit reversal, shift down, complement */
uint64_t skmo_kmer_reverse_complement(uint64_t kmer)
{
    kmer = ((kmer >> 2) & 0x3333333333333333ULL) | ((kmer & 0x3333333333333333ULL) << 2);
    kmer = ((kmer >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((kmer & 0x0f0f0f0f0f0f0f0fULL) << 4);
    kmer = ((kmer >> 8) & 0x00ff00ff00ff00ffULL) | ((kmer & 0x00ff00ff00ff00ffULL) << 8);
    kmer = ((kmer >> 16) & 0x0000ffff0000ffffULL) | ((kmer & 0x0000ffff0000ffffULL) << 16);
    kmer = (kmer >> 32) | (kmer << 32);
    kmer = kmer >> (64 - SKMO_K * 2);
    return ~kmer & skmo_kmer_mask();
}

/* seekmer/_kmer.pxd:222-231 */
static void half_round(uint64_t *a, uint64_t *b, uint64_t *c, uint64_t *d, int s, int t)
{
    *a += *b;
    *c += *d;
    *b = ((*b << s) | (*b >> (64 - s))) ^ *a;
    *d = ((*d << t) | (*d >> (64 - t))) ^ *c;
    *a = (*a << 32) | (*a >> 32);
}

/* seekmer/_kmer.pxd:174-219 -- SipHash-2-4 over one 8-byte word with the fixed
 * key (5381, 42); NOT the standard finalisation: after the length block the
 * reference xors v0 with 0 (line 209) instead of with the block, and the
 * 64-bit result is truncated to a C int (line 219). */
int32_t skmo_kmer_hash(uint64_t kmer)
{
    const uint64_t k0 = 5381, k1 = 42;
    uint64_t b = 8ULL << 56;
    uint64_t v0 = k0 ^ 0x736f6d6570736575ULL;
    uint64_t v1 = k1 ^ 0x646f72616e646f6dULL;
    uint64_t v2 = k0 ^ 0x6c7967656e657261ULL;
    uint64_t v3 = k1 ^ 0x7465646279746573ULL;
    uint64_t mi = kmer;
    v3 ^= mi;
    for (int r = 0; r < 2; ++r) {
        half_round(&v0, &v1, &v2, &v3, 13, 16);
        half_round(&v2, &v1, &v0, &v3, 17, 21);
    }
    v0 ^= mi;
    v3 ^= b;
    for (int r = 0; r < 2; ++r) {
        half_round(&v0, &v1, &v2, &v3, 13, 16);
        half_round(&v2, &v1, &v0, &v3, 17, 21);
    }
    v0 ^= 0;
    v2 ^= 0xff;
    for (int r = 0; r < 4; ++r) {
        half_round(&v0, &v1, &v2, &v3, 13, 16);
        half_round(&v2, &v1, &v0, &v3, 17, 21);
    }
    return (int32_t)(uint32_t)((v0 ^ v1) ^ (v2 ^ v3));
}

/* seekmer/_sequence.pxd:53-75 -- only upper-case ACGT are complemented */
void skmo_sequence_reverse_complement(char *bases, int length)
{
    for (int i = 0; i < length / 2; ++i) {
        char t = bases[length - i - 1];
        bases[length - i - 1] = bases[i];
        bases[i] = t;
    }
    for (int i = 0; i < length; ++i) {
        if (bases[i] == 'A') bases[i] = 'T';
        else if (bases[i] == 'T') bases[i] = 'A';
        else if (bases[i] == 'C') bases[i] = 'G';
        else if (bases[i] == 'G') bases[i] = 'C';
    }
}
