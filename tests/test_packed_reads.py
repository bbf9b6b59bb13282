"""FASTQ text -> packed 2-bit pieces on the host (skm_fastq_packed_*, skm_pack_reads): the reads
are the reference feeders' reads (seekmer/common.py:126-197) under the reference's encoding
(seekmer/_kmer.pxd:253-273) and wildcard rule (seekmer/_mapper.pyx:500-501), for any text, any chunk
size, any number of threads and every parser variant this CPU has.  No GPU."""
import ctypes
import random

import numpy as np
import pytest

from seekmer_amd import _native, common


def expected_packing(read):
    """(code words, bit plane words, length) of one read by the reference's rules, one base at a time."""
    words = max(1, (len(read) + 31) // 32)
    codes = [0] * words
    mask = [0] * words
    for i, c in enumerate(read):
        ch = chr(c)
        code = {'T': 3, 't': 3, 'G': 2, 'g': 2, 'C': 1, 'c': 1}.get(ch, 0)      # _kmer.pxd:253-273
        codes[i >> 5] |= code << (62 - 2 * (i & 31))
        if ch in 'ACGT':                                                         # _mapper.pyx:500-501
            mask[i >> 5] |= 1 << (31 - (i & 31))
    return codes, mask, len(read)


def assemble(pieces, n_streams):
    """Reads of every stream from the pieces, by the rule the mapper applies: a piece replaces what
    its stream delivered from its first_read on."""
    streams = [dict() for _ in range(n_streams)]
    ends = [0] * n_streams
    names = {}
    for piece in pieces:
        s = piece.stream
        for r in [r for r in streams[s] if r >= piece.first_read]:
            del streams[s][r]
        if piece.is_cut:                  # (the longer mate-2 file of a pair of files, cut back)
            ends[s] = piece.first_read
            continue
        codes, lengths = piece.codes, piece.lengths
        exc_reads, exc_masks = piece.exceptions
        exc = {int(r): exc_masks[k] for k, r in enumerate(exc_reads)}
        piece_names = piece.names
        assert list(exc_reads) == sorted(set(exc_reads.tolist()))
        for r in range(piece.n_reads):
            length = int(lengths[r])
            words = max(1, (length + 31) // 32)
            assert words <= piece.code_words
            assert not codes[r, words:].any()                    # zero beyond the read's end
            if r in exc:
                mask = [int(v) for v in exc[r][:words]]
                assert not exc[r][words:].any()
            else:
                mask = [0] * words
                for i in range(length):
                    mask[i >> 5] |= 1 << (31 - (i & 31))
            streams[s][piece.first_read + r] = ([int(v) for v in codes[r, :words]], mask, length)
            if piece_names is not None:
                names[piece.first_read + r] = piece_names[r]
        ends[s] = piece.first_read + piece.n_reads
        if piece.uniform_len is not None:
            assert (lengths == piece.uniform_len).all()
    return streams, ends, names


def reference_reads(paths, paired):
    """The reads (and names) the Python feeders yield -- the mirror of the reference's."""
    feeder = common.feed_pair_ended_reads if paired else common.feed_single_ended_reads
    names, reads = [], []
    for count, batch_names, batch_reads in feeder(*paths):
        names += batch_names
        reads += batch_reads
    return names, reads


def check_files(paths, paired, threads, chunk_bytes, variant=-1, want_names=True):
    host = _native.host()
    assert host.skm_pack_set_variant(variant) == 0
    try:
        feeder = common.PackedReadFeeder(paths, paired, threads=threads, chunk_bytes=chunk_bytes, want_names=want_names)
        pieces = [p.copy_with_names() if want_names else p.copy() for p in feeder]
    finally:
        host.skm_pack_set_variant(-1)
    n_streams = 2 if paired else 1
    streams, ends, names = assemble(pieces, n_streams)
    ref_names, ref_reads = reference_reads(paths, paired)
    n_units = len(ref_reads) // n_streams
    assert min(ends) == n_units, (ends, n_units)
    for u in range(n_units):
        for s in range(n_streams):
            assert streams[s][u] == expected_packing(ref_reads[n_streams * u + s]), (u, s)
        if want_names:
            assert names[u] == ref_names[u]
    return feeder.stats


def _copy_with_names(piece):
    out = piece.copy()
    out_names = piece.names
    return _Named(out, out_names)


class _Named:
    def __init__(self, piece, names):
        self._piece, self.names = piece, names

    def __getattr__(self, name):
        return getattr(self._piece, name)


common.PackedReads.copy_with_names = _copy_with_names


def make_fastq(rng, n_records, kind):
    """FASTQ text of one of several shapes; every shape is legal input for the reference's line rule."""
    lines = []
    for i in range(n_records):
        if kind == 'plain':
            length = 100
        elif kind == 'ragged':
            length = rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 99, 100, 127, 128, 129, 150, 257])
        else:
            length = rng.randint(20, 140)
        read = ''.join(rng.choice('ACGT') for _ in range(length))
        if kind == 'dirty' and length and rng.random() < 0.4:
            chars = list(read)
            for _ in range(rng.randint(1, 4)):
                chars[rng.randrange(length)] = rng.choice('NnacgtRY.-* ')
            read = ''.join(chars)
        name = '@r%d%s' % (i, ' extra words' * rng.randint(0, 2) if kind != 'plain' else '/1')
        qual = ''.join(rng.choice('@+IIIIFFF#,:') for _ in range(length))
        if kind == 'dirty':
            pad = rng.choice(['', '', ' ', '\t', '  '])
            read = rng.choice(['', '', ' ']) + read + pad
        end = '\r\n' if kind == 'crlf' else '\n'
        lines += [name + end, read + end, '+' + (name[1:] if rng.random() < 0.2 else '') + end, qual + end]
    return ''.join(lines).encode()


@pytest.mark.parametrize('kind', ['plain', 'ragged', 'dirty', 'crlf', 'mixed'])
@pytest.mark.parametrize('threads', [0, 3])
def test_packed_feeder_equals_reference_feeders(tmp_path, kind, threads):
    rng = random.Random(hash(kind) & 0xffff)
    path = tmp_path / 'a.fastq'
    path.write_bytes(make_fastq(rng, 700, kind))
    for chunk in (64, 333, 4096, 1 << 20):
        stats = check_files([path], False, threads, chunk)
        assert stats['reads'] >= 700
    other = tmp_path / 'b.fastq'
    other.write_bytes(make_fastq(rng, 650, kind))                 # the shorter mate file ends the pair
    check_files([path, other], True, threads, 500)
    check_files([other, path], True, threads, 777)
    # two pairs of files: the second pair continues where min(records) of the first left off
    check_files([path, other, other, path], True, threads, 1000)
    check_files([path, other, path], False, threads, 900)


@pytest.mark.parametrize('tail', ['no_newline', 'name_only', 'name_only_no_newline', 'blank_lines', 'half_record',
                                  'empty_file', 'one_line'])
def test_packed_feeder_file_endings(tmp_path, tail):
    rng = random.Random(7)
    body = make_fastq(rng, 40, 'mixed')
    text = {
        'no_newline': body[:-1],
        'name_only': body + b'@last\n',
        'name_only_no_newline': body + b'@last',
        'blank_lines': body + b'\n\n\n',
        'half_record': body + b'@last\nACGTNACGT\n+\n',
        'empty_file': b'',
        'one_line': b'@only',
    }[tail]
    path = tmp_path / 'a.fastq'
    path.write_bytes(text)
    for threads in (0, 2):
        for chunk in (64, 100000):
            check_files([path], False, threads, chunk, want_names=False)


def test_packed_feeder_on_text_that_is_not_fastq(tmp_path):
    """The line rule does not care what the lines hold: '@' and '+' at the start of quality lines,
    records of other shapes, random text.  Guessed starts fail here and the proven walk takes over."""
    rng = random.Random(11)
    lines = []
    for i in range(3000):
        lines.append(''.join(rng.choice('@+ACGTN\r acgt') for _ in range(rng.randint(0, 60))))
    path = tmp_path / 'noise.txt'
    path.write_bytes(('\n'.join(lines) + '\n').encode())
    stats = check_files([path], False, 3, 256, want_names=False)
    assert stats['reparsed'] > 0
    # a FASTQ whose quality lines all begin with '@' and whose '+' lines repeat the name
    tricky = []
    for i in range(500):
        read = ''.join(rng.choice('ACGT') for _ in range(50))
        tricky += ['@n%d' % i, read, '+n%d' % i, '@' + 'I' * 49]
    path2 = tmp_path / 'tricky.fastq'
    path2.write_bytes(('\n'.join(tricky) + '\n').encode())
    stats = check_files([path2], False, 3, 300)
    assert stats['reparsed'] == 0 and stats['accepted'] > 10       # every guess proven


def _variants():
    host = _native.host()
    return [v for v in (0, 1, 2) if host.skm_pack_set_variant(v) == 0 and host.skm_pack_set_variant(-1) == 0]


def test_parser_variants_agree(tmp_path):
    variants = _variants()
    assert 0 in variants
    rng = random.Random(3)
    path = tmp_path / 'a.fastq'
    path.write_bytes(make_fastq(rng, 900, 'dirty') + make_fastq(rng, 300, 'ragged'))
    for v in variants:
        check_files([path], False, 2, 2000, variant=v)


def test_pack_reads_of_arrays():
    rng = random.Random(5)
    reads = []
    for i in range(3000):
        length = rng.choice([0, 1, 31, 32, 33, 64, 100, 100, 100, 128])
        read = ''.join(rng.choice('ACGT') for _ in range(length))
        if length and rng.random() < 0.1:
            at = rng.randrange(length)
            read = read[:at] + rng.choice('Nacgt ') + read[at + 1:]
        reads.append(read.encode())
    batch = common.ReadBatch.from_lists(len(reads), None, reads)
    for v in _variants():
        piece = common.PackedReads.from_ascii(batch.bases, batch.offsets, variant=v)
        assert piece.code_words == 4 and piece.n_reads == len(reads)
        streams, ends, _ = assemble([piece], 1)
        for r, read in enumerate(reads):
            assert streams[0][r] == expected_packing(read), r
    # a read longer than the words given is an argument error, not a truncation
    codes = np.zeros((1, 1), dtype=np.uint64)
    lengths = np.zeros(1, dtype=np.uint32)
    offsets = np.asarray([0, 40], dtype=np.int64)
    n_exc = ctypes.c_int64()
    code = _native.host().skm_pack_reads(b'A' * 40, _native.ptr(offsets, _native.c_i64p), 1, 1, codes.ctypes.data,
                                        lengths.ctypes.data, None, None, 0, ctypes.byref(n_exc), -1)
    assert code == _native.SKM_ERR_ARG


class _ThreadRanks:
    """An all-reduce among `world` threads of one process, standing in for the process group."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.lock = threading.Lock()
        self.total = None

    def sum_int64(self, table):
        with self.lock:
            self.total = table.copy() if self.total is None else self.total + table
        self.barrier.wait()
        out = self.total.copy()
        if self.barrier.wait() == 0:
            self.total = None
        self.barrier.wait()
        return out


def read_shares(paths, paired, world, threads, chunk_bytes, count_chunk):
    """Every rank's pieces of a sample shared out over `world` ranks, and every rank's share."""
    import threading
    ranks = _ThreadRanks(world)
    pieces, shares, errors = [None] * world, [None] * world, []

    def one(rank):
        try:
            feeder = common.PackedReadFeeder(paths, paired, threads=threads, chunk_bytes=chunk_bytes, want_names=True,
                                             shard=(rank, world), sum_over_ranks=ranks.sum_int64)
            feeder.COUNT_CHUNK = count_chunk
            pieces[rank] = [p.copy_with_names() for p in feeder]
            shares[rank] = feeder.share
        except BaseException as error:          # (a rank that fails must not leave the others at the barrier)
            errors.append(error)
            ranks.barrier.abort()

    workers = [threading.Thread(target=one, args=(rank,)) for rank in range(world)]
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    if errors:
        raise errors[0]
    return pieces, shares


@pytest.mark.parametrize('kind', ['plain', 'ragged', 'dirty', 'crlf'])
def test_a_sample_shared_out_over_ranks(tmp_path, kind):
    """PackedReadFeeder(shard=(rank, world)): every rank finds units [T r / N, T (r + 1) / N) of the
    sample from newline counts alone (the reference's records are lines 4u .. 4u + 3 whatever they
    hold, seekmer/common.py:126-197) and reads them in one pass; together the ranks deliver the
    one-process reader's reads under the one-process reader's unit numbers -- for files of unequal
    record counts and byte sizes, several pairs of files, a last line without a newline, and more
    ranks than chunks or than units."""
    rng = random.Random(hash(kind) & 0xffff)
    sizes = [(530, 530), (211, 260), (97, 40)]            # records in (mate 1, mate 2) of each pair of files
    paths = []
    for k, (n1, n2) in enumerate(sizes):
        for mate, n in ((1, n1), (2, n2)):
            text = make_fastq(rng, n, kind)
            if k == 1 and mate == 1:
                text = text.rstrip(b'\r\n')                # the last line stays open
            if k == 2 and mate == 2:
                text += b'@name without bases\n'           # a trailing name line: no read
            path = tmp_path / ('s%d_%d.fastq' % (k, mate))
            path.write_bytes(text)
            paths.append(path)
    for paired, files in ((True, paths), (False, paths[::2])):
        n_streams = 2 if paired else 1
        ref_names, ref_reads = reference_reads(files, paired)
        n_units = len(ref_reads) // n_streams
        for world, threads, chunk_bytes, count_chunk in ((2, 2, 4096, 8192), (3, 0, 1000, 700), (7, 3, 512, 3000),
                                                         (5, 1, 1 << 20, 1 << 20)):
            pieces, shares = read_shares(files, paired, world, threads, chunk_bytes, count_chunk)
            at = 0
            for rank in range(world):
                begin, end, first, count = shares[rank]
                assert first == at == n_units * rank // world and count == n_units * (rank + 1) // world - first
                streams, ends, names = assemble(pieces[rank], n_streams)
                assert all(not p.is_cut for p in pieces[rank])          # (the shares end where the shorter file does)
                for s in range(n_streams):
                    assert sorted(streams[s]) == list(range(first, first + count)), (rank, s)
                    for u in range(first, first + count):
                        assert streams[s][u] == expected_packing(ref_reads[n_streams * u + s]), (u, s)
                for u in range(first, first + count):
                    assert names[u] == ref_names[u]
                at += count
            assert at == n_units
    # more ranks than units: the empty shares are empty, the others as above
    tiny = tmp_path / 'tiny.fastq'
    tiny.write_bytes(make_fastq(rng, 3, kind))
    pieces, shares = read_shares([tiny], False, 5, 1, 4096, 64)
    assert [s[3] for s in shares] == [0, 1, 0, 1, 1]
    assert sum(p.n_reads for rank in pieces for p in rank) == 3
