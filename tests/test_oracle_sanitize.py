"""The oracle under AddressSanitizer + UBSan (CPU only): builder, mapper and EM
on the reference's test data must be clean -- the restatement indexes like the
reference does, so a clean run also says the reference's accesses stay in
bounds on these inputs."""
import os
import subprocess
import sys

from conftest import GOLDEN, ROOT

SCRIPT = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle as O
O._LIB_PATH = os.path.join(%(root)r, 'oracle', 'libskm_oracle_asan.so')
ids, seqs = O.read_fasta(os.path.join(%(golden)r, 'human.cdna.21.fa.bz2'))
ids, seqs = ids[:300], seqs[:300]
index = O.build_index(seqs, ids)
reads = O.read_fastq_pairs(os.path.join(%(golden)r, '20_1.fastq'), os.path.join(%(golden)r, '20_2.fastq'))
reads += [b'ACGT' * 7, b'N' * 40, b'acgtn' * 10, seqs[0][:25], seqs[1][10:90].lower(), b'A' * 24, b'']
reads += [seqs[5][i:i + 75] for i in range(0, 600, 7)]
if len(reads) %% 2:
    reads.append(b'ACGTACGTACGTACGTACGTACGTACGTA')
bases, offsets = O.pack_reads(reads)
fld = np.zeros(2000, dtype=np.int64)
r = O.map_batch(index, bases, offsets, len(reads) // 2, True, fld, O.Stats())
r1 = O.map_batch(index, bases, offsets, len(reads), False, fld)
classes = O.Classes(); classes.update(r); classes.update(r1)
class_map, class_count = classes.summarize()
eff = O.effective_lengths(fld, index.lengths)
tpm, iters = O.quantify(eff, class_map, class_count)
print('ok', len(reads), class_count.size, iters)
'''


def test_oracle_is_clean_under_asan_ubsan():
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'libskm_oracle_asan.so'],
                          stdout=subprocess.DEVNULL)
    asan = subprocess.check_output(['gcc', '-print-file-name=libasan.so']).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    proc = subprocess.run([sys.executable, '-c', SCRIPT % {'root': ROOT, 'golden': GOLDEN}],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert proc.stdout.startswith('ok')
    assert 'runtime error' not in proc.stderr and 'AddressSanitizer' not in proc.stderr, proc.stderr[-3000:]
