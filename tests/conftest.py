import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure; never imported by seekmer_amd)."""
    from oracle import oracle as module
    module.build_library()
    return module


@pytest.fixture(scope='session')
def native_libs():
    """Build the product libraries if they are missing (hipcc cross-compiles)."""
    import subprocess
    from seekmer_amd import _native
    if not (os.path.exists(_native.HIP_LIB_PATH) and os.path.exists(_native.HOST_LIB_PATH)):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'seekmer_amd', 'csrc')],
                              stdout=subprocess.DEVNULL)
    return _native


@pytest.fixture(scope='session')
def chr21(oracle):
    """The reference's own test transcriptome (seekmer/test/data/human.cdna.21.fa.bz2)."""
    ids, seqs = oracle.read_fasta(os.path.join(GOLDEN, 'human.cdna.21.fa.bz2'))
    return ids, seqs


@pytest.fixture(scope='session')
def chr21_oracle_index(oracle, chr21):
    ids, seqs = chr21
    return oracle.build_index(seqs, ids)


@pytest.fixture(scope='session')
def pairs21(oracle):
    """The reference's 21 read pairs (seekmer/test/data/20_1.fastq, 20_2.fastq)."""
    return oracle.read_fastq_pairs(os.path.join(GOLDEN, '20_1.fastq'),
                                   os.path.join(GOLDEN, '20_2.fastq'))


def make_product_index(oracle_index, ids=None):
    """KMerIndex (product) holding exactly the oracle's arrays."""
    from seekmer_amd import common
    n = oracle_index.lengths.size
    if ids is None:
        ids = [b'T%07d' % i for i in range(n)]
    width = max(len(i) for i in ids)
    transcripts = np.zeros(n, dtype=[('transcript_id', 'S%d' % width), ('gene_id', 'S1'),
                                     ('length', 'f8')])
    transcripts['transcript_id'] = ids
    transcripts['length'] = oracle_index.lengths
    exons = np.zeros(0, dtype=[('transcript_id', 'S1')])
    return common.KMerIndex(oracle_index.kmers, oracle_index.contigs, oracle_index.sequences,
                            oracle_index.targets, transcripts, exons)
