"""The oracle's builder / mapper / EM on the reference's own test data
(seekmer/test/data, copied to tests/golden) against the reference's own
assertions (seekmer/test/*.py) and the observations of the real reference
recorded in SURVEY.md (tests/golden/reference_observations.json)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope='module')
def observed():
    with open(os.path.join(GOLDEN, 'reference_observations.json')) as f:
        return json.load(f)


def _index_invariants(index, n_transcripts):
    """The satisfiable structural assertions of seekmer/test/test_index_builder.py:56-74."""
    occupied = index.kmers['kmer'] != np.uint64(0xFFFFFFFFFFFFFFFF)
    entries = index.kmers['entry'][occupied]
    entries = np.where(entries < 0, ~entries, entries)
    assert entries.min() == 0
    assert entries.max() == index.contigs.size - 1
    assert index.contigs['offset'].min() == 0
    assert (index.contigs['offset'] + index.contigs['length']).max() == index.sequences.size
    assert b''.join(np.unique(index.sequences)) == b'ACGT'
    assert index.contigs['target_offset'].min() == 0
    assert (index.contigs['target_offset'] + index.contigs['target_count']).max() == index.targets.size
    t = index.targets['entry']
    t = np.where(t < 0, ~t, t)
    assert t.min() == 0
    assert t.max() == n_transcripts - 1


def test_build_with_extra_fasta(oracle, observed):
    """seekmer/test/test_index_builder.py:78-90"""
    ids, seqs = oracle.read_fasta(os.path.join(GOLDEN, 'human.cdna.21.with_extra.fa.gz'))
    assert len(ids) == observed['with_extra_index']['transcripts']
    index = oracle.build_index(seqs, ids)
    _index_invariants(index, 3)
    occupied = index.kmers['kmer'] != np.uint64(0xFFFFFFFFFFFFFFFF)
    assert index.kmers.size == observed['with_extra_index']['table_slots']
    assert int(occupied.sum()) == observed['with_extra_index']['occupied_slots']
    # every k-mer sits in exactly one contig
    assert int((index.contigs['length'] - 24).sum()) == int(occupied.sum())


def test_build_chr21(oracle, chr21, chr21_oracle_index, observed):
    """seekmer/test/test_index_builder.py:46-76 (+ test_read_transcripts:21-31)"""
    ids, seqs = chr21
    assert len(set(ids)) == len(ids) == observed['chr21_index']['transcripts']
    assert sum(len(s) for s in seqs) == observed['chr21_index']['total_bases']
    _index_invariants(chr21_oracle_index, len(ids))
    assert chr21_oracle_index.contigs['target_count'].max() == \
        observed['chr21_index']['max_targets_per_contig']


def test_every_transcript_kmer_maps_back(oracle, chr21, chr21_oracle_index):
    """Index self-consistency: map_kmer of every k-mer of a few transcripts
    returns a contig position whose pooled bases spell that k-mer."""
    import ctypes
    L = oracle.lib()
    L.skmo_map_kmer.restype = ctypes.c_int64
    L.skmo_map_kmer.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    pool = chr21_oracle_index.sequences.tobytes()
    for s in chr21[1][::97]:
        s = s.upper()
        for p in range(0, len(s) - 24, 7):
            kmer = oracle.kmer_encode(s, p)
            v = L.skmo_map_kmer(chr21_oracle_index.c, kmer, None)
            entry = ctypes.c_int32(v & 0xFFFFFFFF).value
            offset = ctypes.c_int32(v >> 32).value
            assert offset >= 0
            contig = entry if entry >= 0 else ~entry
            start = int(chr21_oracle_index.contigs['offset'][contig]) + offset
            stored = oracle.kmer_encode(pool[start:start + 25], 0)
            if entry < 0:
                stored = oracle.kmer_reverse_complement(stored)
            assert stored == kmer


def test_reference_21_pairs(oracle, chr21_oracle_index, pairs21, observed):
    """seekmer/test/test_mapper.py:71-76 and test_infer.py:21-27 + SURVEY.md section 4."""
    want = observed['pairs21_chr21']
    bases, offsets = oracle.pack_reads(pairs21)
    fld = np.zeros(2000, dtype=np.int64)
    result = oracle.map_batch(chr21_oracle_index, bases, offsets, 21, True, fld)
    classes = oracle.Classes()
    classes.update(result)
    class_map, class_count = classes.summarize()
    assert classes.unaligned == want['unaligned']            # the reference's own assertion
    assert int(class_count.sum()) == want['aligned']
    assert class_count.size == want['classes']
    assert np.nonzero(fld)[0].tolist() == want['fld_support']
    assert int(fld.sum()) == want['fld_pairs_counted']
    assert abs(oracle.harmonic_mean_fragment_length(fld) - want['harmonic_mean_fragment_length']) < 5e-4
    eff = oracle.effective_lengths(fld, chr21_oracle_index.lengths)
    tpm, iters = oracle.quantify(eff, class_map, class_count)
    assert int((tpm > 0).sum()) == want['transcripts_with_tpm_gt_0']
    assert abs(tpm.sum() - 1e6) < 1e-3


def test_read_feeder_rules(oracle):
    """seekmer/test/test_mapper.py:20-68: 21 reads per file, bases in ACTGNactg."""
    r1 = oracle.read_fastq_pairs(os.path.join(GOLDEN, '20_1.fastq'))
    assert len(r1) == 21
    both = oracle.read_fastq_pairs(os.path.join(GOLDEN, '20_1.fastq'), os.path.join(GOLDEN, '20_2.fastq'))
    assert len(both) == 42
    for read in both:
        assert set(read) <= set(b'ACTGNactg')
