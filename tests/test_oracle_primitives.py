"""The oracle's primitives against vectors produced by the REFERENCE's own
compiled header-inline code (tests/golden/primitives.json, generator:
tests/golden/make_primitives_golden.py) and, where oracle/_ref exists, against
that build directly."""
import json
import os
import random
import sys

import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope='module')
def golden():
    with open(os.path.join(GOLDEN, 'primitives.json')) as f:
        return json.load(f)


def test_constants(oracle, golden):
    assert golden['size'] == 25
    assert oracle.lib().skmo_kmer_mask() == golden['mask']
    assert golden['invalid'] == 0xFFFFFFFFFFFFFFFF
    assert golden['coordinate_invalid'] == [0, -1]


def test_two_bit_encode(oracle, golden):
    for ch, code in golden['two_bit'].items():
        assert oracle.kmer_append(0, ch.encode('latin1')) == code, ch


def test_kmer_functions(oracle, golden):
    for row in golden['kmers']:
        k = row['kmer']
        assert oracle.kmer_hash(k) == row['hash']
        assert oracle.kmer_reverse_complement(k) == row['rc']
        assert oracle.kmer_decode(k).decode() == row['decode']
        base = row['base'].encode('latin1')
        assert oracle.kmer_append(k, base) == row['append']
        assert oracle.kmer_prepend(k, base) == row['prepend']


def test_encode_and_sequence_rc(oracle, golden):
    for row in golden['encode']:
        assert oracle.kmer_encode(row['seq'].encode('latin1'), row['offset']) == row['kmer']
    for row in golden['sequence_rc']:
        assert oracle.sequence_reverse_complement(row['seq'].encode('latin1')).decode('latin1') == row['rc']


def test_coordinates(golden):
    # the oracle keeps coordinates as plain structs; pin the arithmetic it relies on
    for row in golden['coordinates']:
        e, o = row['a']
        assert row['rc'] == [~e, o]
        assert row['valid'] == (o >= 0)
        assert row['decode'] == [e, o]
        a, b = tuple(row['a']), tuple(row['b'])
        assert row['compare'] == (a > b) - (a < b)
    for row in golden['coordinate_arrays']:
        assert row['rc'] == [[~e, o] for e, o in reversed(row['items'])]


def test_against_reference_build_if_present(oracle):
    ref_dir = os.path.join(ROOT, 'oracle', '_ref')
    sys.path.insert(0, ref_dir)
    try:
        import ref_primitives as R
    except ImportError:
        pytest.skip('oracle/_ref not built (only in the build container)')
    rng = random.Random(5)
    for _ in range(20000):
        k = rng.getrandbits(50)
        assert R.kmer_hash(k) == oracle.kmer_hash(k)
        assert R.kmer_reverse_complement(k) == oracle.kmer_reverse_complement(k)
