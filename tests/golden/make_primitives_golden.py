"""Generate tests/golden/primitives.json from the REFERENCE's own compiled
header-inline primitives (oracle/_ref, built by oracle/build_ref.py from
/root/reference/seekmer/_kmer.pxd, _coordinate.pxd, _coordinate_array.pxd,
_sequence.pxd).  Run in the build container only:

    python oracle/build_ref.py && python tests/golden/make_primitives_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
import ref_primitives as R   # noqa: E402


def main():
    rng = random.Random(20201004)
    alphabet = b'ACGTacgtNnXRY.-'
    kmers = [0, 1, (1 << 50) - 1, 0x2AAAAAAAAAAAA, 0x1555555555555] + \
            [rng.getrandbits(50) for _ in range(2000)]
    seqs = []
    for _ in range(300):
        n = rng.randint(25, 120)
        seqs.append(bytes(rng.choice(alphabet[:8] if rng.random() < 0.8 else alphabet)
                          for _ in range(n)).decode('latin1'))
    out = {
        'provenance': 'reference primitives compiled from /root/reference/seekmer/*.pxd '
                      '(oracle/ref_primitives.pyx harness), seed 20201004',
        'size': R.kmer_size(), 'mask': R.kmer_mask(), 'invalid': R.kmer_invalid(),
        'coordinate_invalid': list(R.coordinate_invalid()),
        'two_bit': {chr(c): R.two_bit_encode(bytes([c])) for c in range(32, 127)},
        'kmers': [],
        'encode': [],
        'sequence_rc': [],
        'coordinates': [],
        'coordinate_arrays': [],
    }
    for k in kmers:
        base = bytes([rng.choice(alphabet)])
        out['kmers'].append({
            'kmer': k, 'hash': R.kmer_hash(k), 'rc': R.kmer_reverse_complement(k),
            'decode': R.kmer_decode(k).decode(), 'base': base.decode('latin1'),
            'append': R.kmer_append(k, base), 'prepend': R.kmer_prepend(k, base),
            'valid': R.kmer_is_valid(k),
        })
    for s in seqs:
        b = s.encode('latin1')
        off = rng.randint(0, len(b) - 25)
        out['encode'].append({'seq': s, 'offset': off, 'kmer': R.kmer_encode(b, off)})
        out['sequence_rc'].append({'seq': s, 'rc': R.sequence_reverse_complement(b).decode('latin1')})
    for _ in range(300):
        e1, o1 = rng.randint(-2**31, 2**31 - 1), rng.randint(-5, 2**31 - 1)
        e2, o2 = (e1, o1) if rng.random() < 0.1 else (rng.randint(-50, 50), rng.randint(-5, 50))
        enc = R.coordinate_encode(e1, o1)
        out['coordinates'].append({
            'a': [e1, o1], 'b': [e2, o2], 'encode': enc, 'decode': list(R.coordinate_decode(enc)),
            'rc': list(R.coordinate_reverse_complement(e1, o1)),
            'valid': R.coordinate_is_valid(e1, o1), 'compare': R.coordinate_compare(e1, o1, e2, o2),
        })
    for _ in range(50):
        items = [(rng.randint(-1000, 1000), rng.randint(0, 1000)) for _ in range(rng.randint(0, 9))]
        out['coordinate_arrays'].append({
            'items': [list(i) for i in items],
            'rc': [list(i) for i in R.coordinate_array_reverse_complement(items)],
        })
    with open(os.path.join(HERE, 'primitives.json'), 'w') as f:
        json.dump(out, f)
    print('wrote primitives.json:', len(out['kmers']), 'k-mers,', len(seqs), 'sequences')


if __name__ == '__main__':
    main()
