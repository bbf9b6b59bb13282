"""libseekmer_host.so under AddressSanitizer + UBSan (CPU only): the ASCII FASTQ reader's two engines
(ragged reads, CRLF, a batch carried across files, early exit with workers in flight, sharding), the
one-pass packed reader and skm_pack_reads over the same files (every parser variant of this CPU),
the index builder on the reference's 3-transcript file and the synthetic generator + FASTQ writer
must be clean."""
import os
import subprocess
import sys

from conftest import GOLDEN, ROOT

SCRIPT = r'''
import os, sys, tempfile
import numpy as np
sys.path.insert(0, %(root)r)
from seekmer_amd import _native
_native.HOST_LIB_PATH = os.path.join(%(root)r, 'seekmer_amd', 'libseekmer_host_asan.so')
from seekmer_amd import common, index_builder, synth
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(5)
paths = []
for f, n in enumerate((1501, 777)):
    for mate in (1, 2):
        lines = []
        for i in range(n):
            m = int(rng.integers(25, 140))
            seq = bytes(rng.choice(list(b'ACGTNacgt'), m).astype(np.uint8))
            eol = b'\r\n' if i %% 9 == 0 else b'\n'
            lines += [b'@r%%d/%%d x' %% (i, mate) + eol, b' ' + seq + eol, b'+' + eol, b'@' + b'I' * (m - 1) + eol]
        path = os.path.join(tmp, 's%%d_%%d.fastq' %% (f, mate))
        open(path, 'wb').write(b''.join(lines) if (f, mate) != (1, 2) else b''.join(lines).rstrip())
        paths.append(path)
plain = [(b.count, b.names, b.reads, b.first_unit) for b in common.NativeReadFeeder(paths, True, batch_units=400)]
fast = [(b.count, b.names, b.reads, b.first_unit) for b in common.NativeReadFeeder(paths, True, batch_units=400, threads=4)]
assert plain == fast and sum(p[0] for p in plain) == 2278
parts = []
for rank in range(3):
    for threads in (0, 2):
        mine = [(b.first_unit, b.reads) for b in common.NativeReadFeeder(paths, True, batch_units=400, threads=threads, shard=(rank, 3))]
        if threads:
            parts += mine
assert sorted(parts) == [(p[3], p[2]) for p in plain]
it = iter(common.NativeReadFeeder(paths, True, batch_units=50, threads=4))
next(it); it.close()
single = [b.count for b in common.NativeReadFeeder(paths[:1] + paths[2:3], False, batch_units=1000, threads=3)]
assert sum(single) == 2278
# the one-pass packed reader over the same files (N, lower case, CRLF, padded lines, a file without its
# last newline): every parser variant this CPU has, tiny and large ranges, with and without workers --
# the pieces hold the reads of the ASCII reader, packed as skm_pack_reads packs them
flat_reads = [r for p in plain for r in p[2]]
for variant in (0, 1, 2):
    if _native.host().skm_pack_set_variant(variant) != 0:
        continue
    for threads, chunk in ((0, 97), (4, 1500), (3, 1 << 20)):
        streams = [dict(), dict()]
        for piece in common.PackedReadFeeder(paths, True, threads=threads, chunk_bytes=chunk, want_names=True):
            keep = [r for r in streams[piece.stream] if r >= piece.first_read]
            for r in keep:
                del streams[piece.stream][r]
            codes, lengths = piece.codes, piece.lengths
            exc = dict(zip(piece.exceptions[0].tolist(), piece.exceptions[1]))
            for r in range(piece.n_reads):
                w = max(1, (int(lengths[r]) + 31) // 32)
                streams[piece.stream][piece.first_read + r] = (codes[r, :w].tolist(), int(lengths[r]),
                                                               exc[r][:w].tolist() if r in exc else None)
        for u in range(2278):
            for s in range(2):
                read = flat_reads[2 * u + s]
                want = common.PackedReads.from_ascii(np.frombuffer(read + b'\0', dtype=np.uint8),
                                                     np.asarray([0, len(read)], dtype=np.int64), variant=variant)
                w = max(1, (len(read) + 31) // 32)
                wexc = want.exceptions
                assert streams[s][u] == (want.codes[0, :w].tolist(), len(read),
                                         wexc[1][0][:w].tolist() if wexc[0].size else None), (variant, threads, u, s)
_native.host().skm_pack_set_variant(-1)
# a sample shared out over three ranks (newline counts per rank, added up, then each rank's byte ranges):
# the ranks' pieces together are the reads above under the same unit numbers
world, tables = 3, []
class Collect(Exception):
    pass
def capture(table):
    tables.append(table.copy())
    raise Collect()
for rank in range(world):
    feeder = common.PackedReadFeeder(paths, True, threads=2, shard=(rank, world), sum_over_ranks=capture)
    feeder.COUNT_CHUNK = 4096
    try:
        feeder.locate_share()
    except Collect:
        pass
total = sum(tables)
seen = [dict(), dict()]
for rank in range(world):
    feeder = common.PackedReadFeeder(paths, True, threads=2, chunk_bytes=3000, shard=(rank, world), sum_over_ranks=lambda t: total)
    feeder.COUNT_CHUNK = 4096
    for piece in feeder:
        assert not piece.is_cut
        for r in range(piece.n_reads):
            assert piece.first_read + r not in seen[piece.stream]
            seen[piece.stream][piece.first_read + r] = int(piece.lengths[r])
assert all(sorted(seen[s]) == list(range(2278)) for s in range(2))
assert all(seen[s][u] == len(flat_reads[2 * u + s]) for s in range(2) for u in range(2278))
ids, seqs = index_builder.read_transcripts(os.path.join(%(golden)r, 'human.cdna.21.with_extra.fa.gz'))
index = index_builder.build(ids, seqs)
assert index.contigs.size == 5
tids, pool, tx = synth.transcriptome(3, 12)
bases, offs = synth.reads(3, pool, tx, 0, 3000, 60, True, n_threads=3)
synth.write_fastq(bases, 3000, 60, True, os.path.join(tmp, 'w_1.fastq'), os.path.join(tmp, 'w_2.fastq'), n_threads=3)
back = [b for b in common.NativeReadFeeder([os.path.join(tmp, 'w_1.fastq'), os.path.join(tmp, 'w_2.fastq')], True, batch_units=3000, threads=2)]
assert len(back) == 1 and back[0].uniform_len == 60 and np.array_equal(back[0].bases[:-1], bases[:-1])
print('ok')
'''


def test_host_library_is_clean_under_asan_ubsan():
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'seekmer_amd', 'csrc'), '../libseekmer_host_asan.so'],
                          stdout=subprocess.DEVNULL)
    asan = subprocess.check_output(['g++', '-print-file-name=libasan.so']).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    proc = subprocess.run([sys.executable, '-c', SCRIPT % {'root': ROOT, 'golden': GOLDEN}],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert proc.stdout.strip().endswith('ok')
    assert 'runtime error' not in proc.stderr and 'AddressSanitizer' not in proc.stderr, proc.stderr[-3000:]
