"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle
on the same inputs.  Integer outputs must be bit-identical."""
import os

import numpy as np
import pytest

from conftest import make_product_index

pytestmark = pytest.mark.gpu


def _adversarial_reads(seqs, rng, n, read_len):
    """Reads that exercise the quirks: Ns, lower case, substitutions and indels
    near junctions, reads from the reverse strand, reads shorter than k,
    exactly k long, garbage."""
    comp = bytes.maketrans(b'ACGT', b'TGCA')
    long_tx = [s for s in seqs if len(s) > read_len + 50]
    out = []
    for i in range(n):
        s = long_tx[rng.integers(len(long_tx))]
        p = int(rng.integers(0, len(s) - read_len))
        r = bytearray(s[p:p + read_len].upper())
        kind = i % 12
        if kind == 1:
            r = bytearray(bytes(r).translate(comp)[::-1])
        elif kind == 2:
            for _ in range(3):
                r[int(rng.integers(len(r)))] = ord('N')
        elif kind == 3:
            q = int(rng.integers(len(r) - 10))
            r[q:q + 10] = bytes(r[q:q + 10]).lower()
        elif kind == 4:
            for _ in range(int(rng.integers(1, 6))):
                q = int(rng.integers(len(r)))
                r[q] = b'ACGT'[int(rng.integers(4))]
        elif kind == 5:
            q = int(rng.integers(5, len(r) - 5))
            del r[q]
        elif kind == 6:
            q = int(rng.integers(5, len(r) - 5))
            r.insert(q, b'ACGT'[int(rng.integers(4))])
        elif kind == 7:
            r = r[:int(rng.integers(0, 25))]
        elif kind == 8:
            r = r[:25]
        elif kind == 9:
            r = bytearray(rng.integers(0, 4, read_len).astype(np.uint8))
            r = bytearray(bytes(r).translate(bytes.maketrans(bytes(range(4)), b'ACGT')))
        elif kind == 10:
            r[:30] = bytes(rng.integers(0, 4, 30).astype(np.uint8)).translate(
                bytes.maketrans(bytes(range(4)), b'ACGT'))
        elif kind == 11:
            r = bytearray(bytes(r).lower())
        out.append(bytes(r))
    return out


def _run_gpu(index, bases, offsets, n_units, paired):
    from seekmer_amd import mapper, common
    result = mapper.MapResult(index, keep_spans=True)
    rm = mapper.ReadMapper(index, result)
    rm.map_batch(common.ReadBatch(n_units, bases, offsets, paired))
    return result, rm.last_batch(n_units)


def _compare_units(oracle_result, gpu_units):
    begin, end, a_entry, a_offset, counts, entries = gpu_units
    np.testing.assert_array_equal(counts, oracle_result.count)
    np.testing.assert_array_equal(entries, oracle_result.entries)
    np.testing.assert_array_equal(begin, oracle_result.begin)
    np.testing.assert_array_equal(end, oracle_result.end)
    np.testing.assert_array_equal(a_entry, oracle_result.anchor_entry)
    np.testing.assert_array_equal(a_offset, oracle_result.anchor_offset)


def _compare_tables(oracle, oracle_result, fld, map_result):
    classes = oracle.Classes()
    classes.update(oracle_result)
    offs, ids, counts = classes.export()
    g_offs, g_ids, g_counts, _, g_fld = map_result.export()
    np.testing.assert_array_equal(g_fld, fld)
    np.testing.assert_array_equal(g_offs, offs)
    np.testing.assert_array_equal(g_ids, ids)
    np.testing.assert_array_equal(g_counts, counts)
    assert map_result.sizes()[2] == classes.unaligned


def test_reference_21_pairs(oracle, native_libs, chr21, chr21_oracle_index, pairs21):
    """The reference's own integration datum (seekmer/test/test_mapper.py:71-76)."""
    index = make_product_index(chr21_oracle_index, chr21[0])
    bases, offsets = oracle.pack_reads(pairs21)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, 21, True, fld)
    result, units = _run_gpu(index, bases, offsets, 21, True)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)
    summarized = result.summarize()
    assert summarized.unaligned == 0            # the reference's assertion
    assert summarized.class_count.size == 13    # SURVEY.md section 4, observed on the reference


@pytest.mark.parametrize('paired', [True, False])
def test_chr21_adversarial(oracle, native_libs, chr21, chr21_oracle_index, paired):
    rng = np.random.default_rng(7)
    reads = _adversarial_reads(chr21[1], rng, 6000, 100)
    index = make_product_index(chr21_oracle_index, chr21[0])
    bases, offsets = oracle.pack_reads(reads)
    n_units = len(reads) // 2 if paired else len(reads)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, n_units, paired, fld)
    result, units = _run_gpu(index, bases, offsets, n_units, paired)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


@pytest.mark.parametrize('paired', [True, False])
def test_hops_that_must_look_their_junction_up(oracle, native_libs, chr21, chr21_oracle_index, paired, monkeypatch):
    """A hop reads map_kmer of its junction k-mer (_mapper.pyx:246-248, :308-310) from the record of
    the contig it leaves when that k-mer is the first or the last k-mer of its own contig -- true of
    every junction of a built index -- and looks it up in the table otherwise (SUCC_LOOKUP,
    skm_device.h).  SKM_TEST_SUCC_LOOKUP=1 marks every successor "look it up" at upload, so the
    fall-back carries all hops here, including the ones behind a kept first hit."""
    monkeypatch.setenv('SKM_TEST_SUCC_LOOKUP', '1')
    rng = np.random.default_rng(7)
    reads = _adversarial_reads(chr21[1], rng, 6000, 100)
    index = make_product_index(chr21_oracle_index, chr21[0])
    assert index.device_info()['successors'] == 1
    bases, offsets = oracle.pack_reads(reads)
    n_units = len(reads) // 2 if paired else len(reads)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, n_units, paired, fld)
    result, units = _run_gpu(index, bases, offsets, n_units, paired)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


@pytest.mark.parametrize('paired', [True, False])
def test_repeats_inside_transcripts_and_segments_shared_in_both_orientations(oracle, native_libs, paired):
    """The junction flags of the contig records (skm_device.h: SUCC_WHOLE, SUCC_MASKED, DevSide::kept)
    are worked out at upload from pairs of target lists; this transcriptome is made to stress them:
    transcripts are chains of segments drawn from a small pool, any segment forwards or reverse
    complemented and possibly several times in one transcript -- contigs that list a transcript
    twice (no mask then: which copy a merge keeps depends on the direction), hops that land on
    contigs in the other orientation, lists that are supersets and lists that are not at nearly
    every junction, lists longer than the eight inline targets.  Per unit and per table against the
    oracle (KMerIndex._filter_on_contig, seekmer/_common.pyx:185-235; _mapper.pyx:229-343)."""
    from seekmer_amd import index_builder
    rng = np.random.default_rng(2024)
    comp = bytes.maketrans(b'ACGT', b'TGCA')
    letters = np.frombuffer(b'ACGT', dtype=np.uint8)
    segments = [bytes(rng.choice(letters, int(rng.integers(26, 140)))) for _ in range(40)]
    transcripts = []
    for _ in range(90):
        parts = []
        for _ in range(int(rng.integers(3, 10))):
            seg = segments[int(rng.integers(len(segments)))]
            parts.append(seg.translate(comp)[::-1] if rng.random() < 0.35 else seg)
        if rng.random() < 0.5:                      # a segment twice in the same transcript, some way apart
            parts.insert(int(rng.integers(len(parts) + 1)), parts[0])
        transcripts.append(b''.join(parts))
    ids = [b'R%04d' % i for i in range(len(transcripts))]
    tx_offsets = np.concatenate([[0], np.cumsum([len(t) for t in transcripts])]).astype(np.int64)
    pool = np.frombuffer(b''.join(transcripts) + b'\0', dtype=np.uint8).copy()
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    counts = index.contigs['target_count']
    assert counts.max() > 8 and index.device_info()['successors'] == 1
    oindex = oracle.OracleIndex(index.kmers, index.contigs, index.sequences, index.targets,
                                lengths=np.diff(tx_offsets))
    # (a transcript listed twice by one contig: the case the masks must leave alone)
    entries = index.targets[index.targets.dtype.names[0]]
    starts = index.contigs['target_offset']
    assert any(len(set(entries[a:a + n].tolist())) < n for a, n in zip(starts.tolist(), counts.tolist()))
    if paired:                  # mates of one fragment (the adversarial reads are unrelated to one another)
        reads = []
        usable = [t for t in transcripts if len(t) > 320]
        for u in range(6000):
            t = usable[int(rng.integers(len(usable)))]
            size = int(rng.integers(150, 320))
            at = int(rng.integers(0, len(t) - size))
            fragment = t[at:at + size]
            mates = [bytearray(fragment[:100]), bytearray(fragment[-100:].translate(comp)[::-1])]
            if u % 7 == 3:
                mates.reverse()                                   # the fragment from the other strand
            for m in mates:
                roll = rng.random()
                if roll < 0.25:
                    m[int(rng.integers(len(m)))] = b'ACGT'[int(rng.integers(4))]
                elif roll < 0.32:
                    del m[int(rng.integers(5, len(m) - 5))]
                elif roll < 0.39:
                    m.insert(int(rng.integers(5, len(m) - 5)), b'ACGT'[int(rng.integers(4))])
            reads += [bytes(m) for m in mates]
    else:
        reads = _adversarial_reads(transcripts, rng, 12000, 100)
    bases, offsets = oracle.pack_reads(reads)
    n_units = len(reads) // 2 if paired else len(reads)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(oindex, bases, offsets, n_units, paired, fld)
    assert (expected.count > 0).mean() > 0.4
    result, units = _run_gpu(index, bases, offsets, n_units, paired)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


@pytest.mark.parametrize('signatures', [True, False])
def test_the_roll_with_and_without_signatures(oracle, native_libs, chr21, chr21_oracle_index, signatures, monkeypatch):
    """_find_first_kmer's roll (seekmer/_mapper.pyx:207-216) asks a table of signatures by minimizer
    before the k-mer table (skm_device.h: kmer_min_hash; skm_index_layout[7]); SKM_NO_SIGNATURES=1
    leaves it out and the roll asks the buckets of five k-mers at a time.  Reads made for the roll:
    substitutions in the first k bases (one, two, at either end of the first k-mer), the first 30 or
    60 bases random, Ns in front, reads that map nowhere, reads just k + 1 long -- the first hit and
    everything behind it as the oracle's."""
    if not signatures:
        monkeypatch.setenv('SKM_NO_SIGNATURES', '1')
    rng = np.random.default_rng(19)
    comp = bytes.maketrans(b'ACGT', b'TGCA')
    long_tx = [s for s in chr21[1] if len(s) > 400]
    reads = []
    for i in range(8000):
        t = long_tx[int(rng.integers(len(long_tx)))]
        at = int(rng.integers(0, len(t) - 130))
        r = bytearray(t[at:at + 100].upper())
        kind = i % 10
        if kind in (0, 1, 2):
            for _ in range(1 + (kind == 2)):
                q = int(rng.integers(0, 25 if kind else 3))
                r[q] = b'ACGT'[(b'ACGT'.index(bytes([r[q]])) + 1 + int(rng.integers(3))) % 4] if bytes([r[q]]) in b'ACGT' else ord('A')
        elif kind == 3:
            r[24] = ord('N')
        elif kind == 4:
            r[:30] = bytes(rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), 30))
        elif kind == 5:
            r[:60] = bytes(rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), 60))
        elif kind == 6:
            r = bytearray(bytes(rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), 100)))
        elif kind == 7:
            r = r[:26]
            r[0] = ord('N')
        elif kind == 8:
            r[:3] = b'NNN'
        if i % 3 == 0:
            r = bytearray(bytes(r).translate(comp)[::-1])
        reads.append(bytes(r))
    index = make_product_index(chr21_oracle_index, chr21[0])
    assert (index.device_info()['signature_slots'] > 0) == signatures
    bases, offsets = oracle.pack_reads(reads)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, len(reads), False, fld)
    result, units = _run_gpu(index, bases, offsets, len(reads), False)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


def test_edge_windows_and_pool_fallback(oracle, native_libs, chr21, chr21_oracle_index):
    """A built index takes the 8-base windows at contig ends from first_kmer/last_kmer
    (skm_index_info[6] == 1); an index whose edge k-mers do not spell the pooled bases
    must fall back to the pool and still follow the reference's arithmetic
    (get_tail_kmer reads the row, get_contig_sequence the pool, _common.pyx:103-137,
    241-266)."""
    rng = np.random.default_rng(23)
    reads = _adversarial_reads(chr21[1], rng, 4000, 100)
    bases, offsets = oracle.pack_reads(reads)
    index = make_product_index(chr21_oracle_index, chr21[0])
    assert index.device_info()['edge_windows'] == 1
    contigs = chr21_oracle_index.contigs.copy()
    names = contigs.dtype.names
    first, last = names[2], names[3]
    contigs[first][::3] ^= np.uint64(0x155)           # low bases of every third first_kmer
    contigs[last][1::3] ^= np.uint64(0x2AA) << np.uint64(34)
    tampered = oracle.OracleIndex(chr21_oracle_index.kmers, contigs, chr21_oracle_index.sequences,
                                  chr21_oracle_index.targets, lengths=chr21_oracle_index.lengths)
    t_index = make_product_index(tampered, chr21[0])
    assert t_index.device_info()['edge_windows'] == 0
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(tampered, bases, offsets, len(reads) // 2, True, fld)
    result, units = _run_gpu(t_index, bases, offsets, len(reads) // 2, True)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


def test_unsorted_target_slices_take_the_walks(oracle, native_libs, chr21, chr21_oracle_index):
    """The register paths for short lists assume every target slice ascends by signed entry
    (true of any built index, verified at upload: skm_index_info[7]).  An index whose slices are
    out of order must be mapped with the reference's two-pointer walks (_common.pyx:185-235,
    _mapper.pyx:350-397), whatever they make of it."""
    rng = np.random.default_rng(29)
    reads = _adversarial_reads(chr21[1], rng, 4000, 100)
    bases, offsets = oracle.pack_reads(reads)
    targets = chr21_oracle_index.targets.copy()
    contigs = chr21_oracle_index.contigs
    swapped = 0
    for c in range(0, contigs.size, 2):
        first, n = int(contigs['target_offset'][c]), int(contigs['target_count'][c])
        if n >= 2 and targets['entry'][first] != targets['entry'][first + n - 1]:
            targets[[first, first + n - 1]] = targets[[first + n - 1, first]]
            swapped += 1
    assert swapped > 100
    shuffled = oracle.OracleIndex(chr21_oracle_index.kmers, contigs, chr21_oracle_index.sequences,
                                  targets, lengths=chr21_oracle_index.lengths)
    index = make_product_index(shuffled, chr21[0])
    assert index.device_info()['sorted_targets'] == 0
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(shuffled, bases, offsets, len(reads) // 2, True, fld)
    result, units = _run_gpu(index, bases, offsets, len(reads) // 2, True)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


@pytest.mark.parametrize('paired', [True, False])
def test_unassigned_slots_are_misses(oracle, native_libs, chr21, chr21_oracle_index, paired):
    """An occupied slot whose position has offset < 0 is a miss for every caller
    (_mapper.pyx:203, 211; _common.pyx:84-97).  No built index holds one
    (tests/test_host_native.py::test_builder_never_leaves_an_occupied_slot_unassigned), so
    the branch is driven with a tampered table: a third of the occupied slots lose their
    position in three shapes -- the invalid coordinate (0, -1), the entry kept with offset -1,
    and the complement of the real offset."""
    rng = np.random.default_rng(41)
    reads = _adversarial_reads(chr21[1], rng, 6000, 100)
    bases, offsets = oracle.pack_reads(reads)
    n_units = len(reads) // 2 if paired else len(reads)
    kmers = chr21_oracle_index.kmers.copy()
    occupied = np.flatnonzero(kmers['kmer'] != np.uint64(0xFFFFFFFFFFFFFFFF))
    chosen = rng.choice(occupied, occupied.size // 3, replace=False)
    a, b, c = np.array_split(chosen, 3)
    kmers['entry'][a] = 0
    kmers['offset'][a] = -1
    kmers['offset'][b] = -1
    kmers['offset'][c] = ~kmers['offset'][c]
    assert (kmers['offset'][chosen] < 0).all()
    tampered = oracle.OracleIndex(kmers, chr21_oracle_index.contigs, chr21_oracle_index.sequences,
                                  chr21_oracle_index.targets, lengths=chr21_oracle_index.lengths)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(tampered, bases, offsets, n_units, paired, fld)
    fld_clean = np.zeros(2000, dtype=np.int64)
    clean = oracle.map_batch(chr21_oracle_index, bases, offsets, n_units, paired, fld_clean)
    changed = ((expected.begin != clean.begin) | (expected.end != clean.end) | (expected.count != clean.count)
               | (expected.anchor_entry != clean.anchor_entry) | (expected.anchor_offset != clean.anchor_offset))
    assert changed.sum() > n_units // 20              # the branch decides a good share of the units
    result, units = _run_gpu(make_product_index(tampered, chr21[0]), bases, offsets, n_units, paired)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


def test_table_the_reference_probe_does_not_reach_everywhere(oracle, native_libs, chr21, chr21_oracle_index):
    """The device probes a bucketised copy of the k-mer table (skm_index_layout), which is
    the reference's map_kmer exactly when the table is a set the reference's linear probe
    reaches everywhere -- checked slot by slot at upload.  A table that is not (here: k-mers
    deleted out of the middle of probe chains, which hides the ones stored behind them, and a
    k-mer stored a second time) must be probed in the reference's own layout and follow
    _common.pyx:75-97 to the letter: the hidden k-mers are misses."""
    rng = np.random.default_rng(53)
    reads = _adversarial_reads(chr21[1], rng, 6000, 100)
    bases, offsets = oracle.pack_reads(reads)
    n_units = len(reads) // 2
    built = make_product_index(chr21_oracle_index, chr21[0])
    info = built.device_info()
    assert info['bucketed'] == 1 and info['slots_unreached'] == 0 and info['kmers_twice'] == 0
    assert info['bucket_kmers'] == (chr21_oracle_index.kmers['kmer'] != np.uint64(0xFFFFFFFFFFFFFFFF)).sum()
    kmers = chr21_oracle_index.kmers.copy()
    invalid = np.uint64(0xFFFFFFFFFFFFFFFF)
    occupied = np.flatnonzero(kmers['kmer'] != invalid)
    gone = rng.choice(occupied, occupied.size // 20, replace=False)
    kmers['kmer'][gone] = invalid
    kmers['entry'][gone] = 0
    kmers['offset'][gone] = -1
    free = np.flatnonzero(kmers['kmer'] == invalid)
    kmers[free[:50]] = kmers[occupied[1000:1050]]               # (whichever copy the probe meets first wins)
    tampered = oracle.OracleIndex(kmers, chr21_oracle_index.contigs, chr21_oracle_index.sequences,
                                  chr21_oracle_index.targets, lengths=chr21_oracle_index.lengths)
    index = make_product_index(tampered, chr21[0])
    info = index.device_info()
    assert info['bucketed'] == 0 and info['slots_unreached'] > 100
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(tampered, bases, offsets, n_units, True, fld)
    result, units = _run_gpu(index, bases, offsets, n_units, True)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)
    # the buckets keep 50 bits of a k-mer (skm_device.h: DevBucket): a table with an entry that has
    # more is not theirs to hold either -- the entry can never be found, as in the reference
    kmers = chr21_oracle_index.kmers.copy()
    kmers['kmer'][occupied[::997]] |= np.uint64(1) << np.uint64(55)
    tampered = oracle.OracleIndex(kmers, chr21_oracle_index.contigs, chr21_oracle_index.sequences,
                                  chr21_oracle_index.targets, lengths=chr21_oracle_index.lengths)
    index = make_product_index(tampered, chr21[0])
    assert index.device_info()['bucketed'] == 0
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(tampered, bases, offsets, n_units, True, fld)
    result, units = _run_gpu(index, bases, offsets, n_units, True)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


def test_quantify_resident_without_classes(oracle, native_libs, chr21, chr21_oracle_index):
    """Nothing mapped: quantify() returns zeros (seekmer/infer.py:106-107), no EM step."""
    from seekmer_amd import infer
    index = make_product_index(chr21_oracle_index, chr21[0])
    rng = np.random.default_rng(31)
    reads = [bytes(rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), 60)) for _ in range(64)]
    bases, offsets = oracle.pack_reads(reads)
    result, _ = _run_gpu(index, bases, offsets, 32, True)
    assert result.sizes()[0] == 0
    tpm, iters = infer.quantify_resident(result, return_iters=True)
    assert iters == 0 and not tpm.any()


def test_config1_synthetic(oracle, native_libs):
    """BASELINE.json configs[0]: 1k-transcript synthetic index, 100k 2x75 pairs."""
    from seekmer_amd import synth, index_builder, infer
    ids, pool, tx_offsets = synth.transcriptome(1, 100)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    oindex = oracle.OracleIndex(index.kmers, index.contigs, index.sequences, index.targets,
                                lengths=np.diff(tx_offsets))
    n_units = 100000
    bases, offsets = synth.reads(1, pool, tx_offsets, 0, n_units, 75, True)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(oindex, bases, offsets, n_units, True, fld)
    result, units = _run_gpu(index, bases, offsets, n_units, True)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)

    # quantification: effective lengths bit-exact, EM within 1e-9 per fixed step count
    summarized = result.summarize()
    eff = oracle.effective_lengths(fld, oindex.lengths)
    np.testing.assert_array_equal(summarized.effective_lengths, eff)
    classes = oracle.Classes()
    classes.update(expected)
    class_map, class_count = classes.summarize()
    np.testing.assert_array_equal(summarized.class_map, class_map)
    np.testing.assert_array_equal(summarized.class_count, class_count)
    tpm_ref, iters_ref = oracle.quantify(eff, class_map, class_count)
    tpm_gpu, iters_gpu = infer.quantify(summarized, return_iters=True)
    assert iters_gpu == iters_ref
    mask = tpm_ref > 0
    np.testing.assert_array_equal(tpm_gpu > 0, mask)
    rel = np.abs(tpm_gpu[mask] - tpm_ref[mask]) / tpm_ref[mask]
    assert rel.max() < 1e-4, rel.max()          # north_star tolerance on TPM
    x0 = np.ones(eff.size) / eff
    x0 /= x0.sum()
    for k in (1, 2, 5):
        x_ref, _ = oracle.em(x0, eff, class_map, class_count, fixed_iters=k)
        x_gpu, it = infer.em(x0, eff, class_map, class_count, fixed_iters=k, return_iters=True)
        assert it == k
        np.testing.assert_allclose(x_gpu, x_ref, rtol=1e-9, atol=1e-300)


def test_quantify_resident_matches_host_path(oracle, native_libs):
    """skm_quant_infer (histogram -> effective lengths -> numpy-exact start vector -> EM ->
    TPM on the device) against MapResult.summarize() + quantify(), which does the vector
    arithmetic in numpy as the reference does, and against the oracle."""
    from seekmer_amd import synth, index_builder, infer
    ids, pool, tx_offsets = synth.transcriptome(3, 60)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    oindex = oracle.OracleIndex(index.kmers, index.contigs, index.sequences, index.targets,
                                lengths=np.diff(tx_offsets))
    n_units = 40000
    bases, offsets = synth.reads(3, pool, tx_offsets, 0, n_units, 100, True)
    result, _ = _run_gpu(index, bases, offsets, n_units, True)
    tpm_dev, iters_dev, eff_dev = infer.quantify_resident(result, return_iters=True,
                                                          return_effective_lengths=True)
    summarized = result.summarize()
    tpm_host, iters_host = infer.quantify(summarized, return_iters=True)
    np.testing.assert_array_equal(eff_dev, summarized.effective_lengths)
    assert iters_dev == iters_host
    # same start vector bit for bit (numpy's blocked pairwise sum restated on the device), same
    # EM kernels, same TPM arithmetic: the two paths agree to the last bit
    np.testing.assert_array_equal(tpm_dev, tpm_host)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(oindex, bases, offsets, n_units, True, fld)
    classes = oracle.Classes()
    classes.update(expected)
    class_map, class_count = classes.summarize()
    tpm_ref, iters_ref = oracle.quantify(oracle.effective_lengths(fld, oindex.lengths), class_map, class_count)
    assert iters_dev == iters_ref
    mask = tpm_ref > 0
    np.testing.assert_array_equal(tpm_dev > 0, mask)
    assert (np.abs(tpm_dev[mask] - tpm_ref[mask]) / tpm_ref[mask]).max() < 1e-4


@pytest.mark.parametrize('paired', [True, False])
def test_access_counters_equal_the_oracles(oracle, native_libs, chr21, chr21_oracle_index, paired):
    """The counting build of the map kernel (map_units_kernel<true>) performs the reference's
    access pattern and tallies it; its counters are what bench.py turns into algorithmic bytes
    (B_map), so they must equal the oracle's, field by field -- and the results of
    the counting build must be the production build's."""
    from seekmer_amd import mapper, common
    rng = np.random.default_rng(37)
    reads = _adversarial_reads(chr21[1], rng, 5000, 100)
    index = make_product_index(chr21_oracle_index, chr21[0])
    bases, offsets = oracle.pack_reads(reads)
    n_units = len(reads) // 2 if paired else len(reads)
    fld = np.zeros(2000, dtype=np.int64)
    ostats = oracle.Stats()
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, n_units, paired, fld, stats=ostats)
    result = mapper.MapResult(index, keep_spans=True)
    result.set_stats(True)
    rm = mapper.ReadMapper(index, result)
    rm.map_batch(common.ReadBatch(n_units, bases, offsets, paired))
    _compare_units(expected, rm.last_batch(n_units))
    _compare_tables(oracle, expected, fld, result)
    counted = result.access_stats()
    for name, value in ostats.as_dict().items():
        assert counted[name] == value, (name, counted[name], value)
    assert (counted['read_bases'] + 16 * counted['slots'] + 48 * counted['contig_reads']
            + 8 * (counted['targets_copied'] + counted['targets_merged']) + 8 * counted['seq_fetches']
            + 4 * counted['tuple_ids']) == ostats.algorithmic_bytes()


def test_batches_accumulate(oracle, native_libs, chr21, chr21_oracle_index):
    """Several batches into one MapResult == one big batch (first-seen order kept)."""
    from seekmer_amd import mapper, common
    rng = np.random.default_rng(11)
    reads = _adversarial_reads(chr21[1], rng, 4000, 90)
    index = make_product_index(chr21_oracle_index, chr21[0])
    bases, offsets = oracle.pack_reads(reads)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, 2000, True, fld)
    result = mapper.MapResult(index)
    rm = mapper.ReadMapper(index, result)
    for lo, hi in ((0, 700), (700, 701), (701, 2000)):
        sub_off = offsets[2 * lo:2 * hi + 1]
        rm.map_batch(common.ReadBatch(hi - lo, bases, np.ascontiguousarray(sub_off), True))
    _compare_tables(oracle, expected, fld, result)


def test_em_device_table_path_and_determinism(oracle, native_libs):
    """skm_quant_create_from_mapper (table never leaves HBM, classes ordered by
    first-seen on the device) == host CSR path == oracle; two runs are bitwise equal."""
    from seekmer_amd import synth, index_builder, infer
    ids, pool, tx_offsets = synth.transcriptome(6, 60)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    n_units = 30000
    bases, offsets = synth.reads(6, pool, tx_offsets, 0, n_units, 100, True)
    result, _ = _run_gpu(index, bases, offsets, n_units, True)
    summarized = result.summarize()
    eff = summarized.effective_lengths
    x0 = np.ones(eff.size) / eff
    x0 /= x0.sum()
    quant = infer._QuantHandle.from_map_result(result, eff.size)
    x_dev, it_dev = quant.em(x0, eff)
    x_dev2, it_dev2 = quant.em(x0, eff)
    quant.close()
    assert it_dev == it_dev2
    np.testing.assert_array_equal(x_dev, x_dev2)
    x_host, it_host = infer.em(x0, eff, summarized.class_map, summarized.class_count, return_iters=True)
    assert it_host == it_dev
    np.testing.assert_array_equal(x_host, x_dev)       # same class order -> same arithmetic
    x_ref, it_ref = oracle.em(x0, eff, summarized.class_map, summarized.class_count)
    assert it_ref == it_dev
    np.testing.assert_allclose(x_dev, x_ref, rtol=1e-9, atol=1e-300)


def test_class_views_when_first_seen_values_are_shared(oracle, native_libs):
    """The class views are ordered through a bitmap of the first-seen unit indices, which needs
    them distinct (they are, for tables the mapper filled: a unit belongs to one class).  A table
    merged from hand-made input may share them: the build then falls back to its sorting path.
    Same classes, same counts, EM to 1e-9 of the host path on the merged table's own export."""
    from seekmer_amd import synth, index_builder, infer, mapper
    ids, pool, tx_offsets = synth.transcriptome(8, 40)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    n_units = 20000
    bases, offsets = synth.reads(8, pool, tx_offsets, 0, n_units, 100, True)
    result, _ = _run_gpu(index, bases, offsets, n_units, True)
    offs, targets, counts, first, fld = result.export()
    assert np.unique(first).size == first.size
    merged = mapper.MapResult(index)
    merged.merge_table(offs, targets, counts, first // 3, result.sizes()[2], fld)
    m_offs, m_targets, m_counts, m_first, _ = merged.export()
    assert np.unique(m_first).size < m_first.size and (np.diff(m_first) >= 0).all()
    as_set = lambda o, t, c: sorted((tuple(t[o[k]:o[k + 1]]), int(c[k])) for k in range(c.size))
    assert as_set(m_offs, m_targets, m_counts) == as_set(offs, targets, counts)
    summarized = merged.summarize()
    eff = summarized.effective_lengths
    x0 = np.ones(eff.size) / eff
    x0 /= x0.sum()
    quant = infer._QuantHandle.from_map_result(merged, eff.size)
    x_dev, it_dev = quant.em(x0, eff)
    x_again, it_again = quant.em(x0, eff)
    quant.close()
    np.testing.assert_array_equal(x_dev, x_again)
    x_host, it_host = infer.em(x0, eff, summarized.class_map, summarized.class_count, return_iters=True)
    assert abs(it_host - it_dev) <= 1 and it_again == it_dev
    np.testing.assert_allclose(x_dev, x_host, rtol=1e-6 if it_host != it_dev else 1e-9, atol=1e-300)
    # and the table the mapper filled itself (distinct values: the bitmap path) gives the same EM
    quant = infer._QuantHandle.from_map_result(result, eff.size)
    x_own, _ = quant.em(x0, eff)
    quant.close()
    np.testing.assert_allclose(x_own, x_dev, rtol=1e-6 if it_host != it_dev else 1e-9, atol=1e-300)


def test_quantification_through_a_one_rank_communicator(oracle, native_libs, monkeypatch):
    """The N > 1 data path with N = 1: an RCCL communicator of one rank is created as the ranks of
    a sharded run create theirs, skm_quant_infer all-reduces the histogram + aligned total and the
    per-step numerators through it, a bootstrap handle with the communicator attached takes the
    one-by-one path (its collectives must stay matched).  A sum over one rank is the identity:
    the same bits as without the communicator, the same step count."""
    import ctypes
    from seekmer_amd import synth, index_builder, infer, parallel, _native
    ids, pool, tx_offsets = synth.transcriptome(9, 50)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    n_units = 30000
    bases, offsets = synth.reads(9, pool, tx_offsets, 0, n_units, 100, True)
    result, _ = _run_gpu(index, bases, offsets, n_units, True)
    tpm, iters, eff = infer.quantify_resident(result, return_iters=True, return_effective_lengths=True)
    hip = _native.hip()
    raw = ctypes.create_string_buffer(128)
    _native.check(hip.skm_comm_unique_id(raw))
    comm = parallel.create_comm(0, raw.raw, 0, 1)
    try:
        n = ctypes.c_int(0)
        _native.check(hip.skm_comm_count(comm, ctypes.byref(n)))
        assert n.value == 1
        tpm_c, iters_c, eff_c = infer.quantify_resident(result, comm=comm, return_iters=True,
                                                        return_effective_lengths=True)
        assert iters_c == iters
        np.testing.assert_array_equal(eff_c, eff)
        np.testing.assert_array_equal(tpm_c, tpm)
        x0 = tpm / tpm.sum()
        quant = infer._QuantHandle.from_map_result(result, len(ids))
        plain, _, steps = quant.bootstrap(11, 77, x0, eff)            # (the working set of eight)
        _native.check(hip.skm_quant_set_comm(quant.handle, comm))
        through, _, steps_c = quant.bootstrap(11, 77, x0, eff)        # (one by one, all-reduced)
        quant.close()
        np.testing.assert_array_equal(steps_c, steps)
        np.testing.assert_array_equal(through, plain)
        # The per-step numerators of a rank (rows summed per transcript, one launch:
        # em_rows_finalize_kernel<TO_ACC>) on a table with a transcript in thousands of classes --
        # several 512-entry rows, summed by different blocks, added in row order by the last to
        # arrive -- and with transcripts in no class: the same bits through the communicator.
        rng = np.random.default_rng(5)
        n_tx, n_classes = 300, 4000
        cls = np.repeat(np.arange(n_classes), rng.integers(1, 6, n_classes))
        tx = rng.integers(1, n_tx - 20, cls.size)
        tx[rng.random(cls.size) < 0.35] = 7
        offsets_csr, targets_csr = infer._csr_from_class_map(np.vstack([cls, tx]).astype(np.int64), n_classes)
        counts = rng.integers(0, 40, n_classes).astype('f8')
        lengths = rng.uniform(50, 3000, n_tx)
        start = 1.0 / lengths
        start /= start.sum()
        skewed = infer._QuantHandle.from_csr(n_tx, offsets_csr, targets_csr, counts)
        alone, steps_alone = skewed.em(start, lengths)
        _native.check(hip.skm_quant_set_comm(skewed.handle, comm))
        reduced, steps_reduced = skewed.em(start, lengths)
        skewed.close()
        assert steps_reduced == steps_alone
        np.testing.assert_array_equal(reduced, alone)
        # A rank whose shard produced NO class while the sample has aligned units elsewhere:
        # it must still go through the set-up with empty class views and every collective of the EM.
        # One rank cannot reach that state by itself, so a test hook adds "units the other ranks
        # aligned" to the reduced total.  With nobody contributing numerators the first step leaves
        # no abundance above the floor -- the defined outcome (numpy raises on max() of an empty
        # selection, seekmer/infer.py:160) -- after the step's collectives have all run.
        junk = np.random.default_rng(3).integers(0, 4, 2 * 5000 * 100).astype(np.uint8)
        junk = np.frombuffer(bytes(junk).translate(bytes.maketrans(bytes(range(4)), b'ACGT')), dtype=np.uint8)
        empty, _ = _run_gpu(index, np.concatenate([junk, np.zeros(1, np.uint8)]),
                            np.arange(2 * 5000 + 1, dtype=np.int64) * 100, 5000, True)
        assert empty.sizes()[0] == 0 and empty.sizes()[2] == 5000          # no class, everything unaligned
        zeros, it0 = infer.quantify_resident(empty, comm=comm, return_iters=True)
        assert it0 == 0 and not zeros.any()                                # alone: nothing to quantify
        monkeypatch.setenv('SKM_TEST_ALIGNED_GLOBAL', '123456')
        with pytest.raises(_native.NativeError) as raised:
            infer.quantify_resident(empty, comm=comm, return_iters=True)
        assert raised.value.code == _native.SKM_ERR_UNDEFINED
        monkeypatch.delenv('SKM_TEST_ALIGNED_GLOBAL')
        again, it_again = infer.quantify_resident(result, comm=comm, return_iters=True)    # the communicator is still in step
        assert it_again == iters
        np.testing.assert_array_equal(again, tpm)
    finally:
        parallel.destroy_comm(comm)


def test_em_skewed_and_degenerate_tables(oracle, native_libs):
    """A transcript in thousands of classes (several 512-entry rows), duplicate
    ids inside a tuple, transcripts in no class, zero counts (bootstrap), and
    the NaN -> 0 rule."""
    from seekmer_amd import infer
    rng = np.random.default_rng(5)
    n_tx, n_classes = 300, 4000
    sizes = rng.integers(1, 6, n_classes)
    cls = np.repeat(np.arange(n_classes), sizes)
    tx = rng.integers(1, n_tx - 20, cls.size)           # ids >= n_tx - 20 never appear
    tx[rng.random(cls.size) < 0.35] = 7                  # heavy transcript
    class_map = np.vstack([cls, tx]).astype(np.int64)
    class_count = rng.integers(0, 40, n_classes).astype('f8')
    class_count[:5] = 0
    l = rng.uniform(50, 3000, n_tx)
    x0 = 1.0 / l
    x0 /= x0.sum()
    for k in (1, 3):
        x_ref, _ = oracle.em(x0, l, class_map, class_count, fixed_iters=k)
        x_gpu, it = infer.em(x0, l, class_map, class_count, fixed_iters=k, return_iters=True)
        assert it == k
        np.testing.assert_allclose(x_gpu, x_ref, rtol=1e-9, atol=1e-300)
    x_ref, it_ref = oracle.em(x0, l, class_map, class_count)
    x_gpu, it_gpu = infer.em(x0, l, class_map, class_count, return_iters=True)
    assert it_gpu == it_ref
    np.testing.assert_allclose(x_gpu, x_ref, rtol=1e-8, atol=1e-300)
    # class_map rows in arbitrary order are accepted (pairs keep their order inside a class)
    perm = rng.permutation(cls.size)
    x_perm, _ = infer.em(x0, l, class_map[:, perm], class_count, fixed_iters=2, return_iters=True)
    x_ref2, _ = oracle.em(x0, l, class_map[:, np.sort(perm, kind='stable')], class_count, fixed_iters=2)
    np.testing.assert_allclose(x_perm, x_ref2, rtol=1e-9, atol=1e-300)


def _check_multinomial_dispersion(counts, class_count):
    """Second moments of B draws of multinomial(n, p = class_count / n) (mean AND
    variance): per class the sample variance over the replicates against n p (1 - p), and
    Pearson's statistic sum_c (x_c - n p_c)^2 / (n p_c) of every replicate, which is
    chi-square with C' - 1 degrees of freedom (C' = classes with p > 0).  Bounds are 6 sigma of
    the respective sampling distributions, so a correct generator fails with p < 1e-8."""
    counts = np.asarray(counts, dtype='f8')
    n_boot = counts.shape[0]
    n = class_count.sum()
    p = class_count / n
    live = p > 0
    assert (counts[:, ~live] == 0).all()
    var = counts[:, live].var(axis=0, ddof=1)
    expect = n * p[live] * (1 - p[live])
    big = expect > 25                                  # near-normal cells: var * (B-1) / expect ~ chi2(B-1)
    ratio = var[big] / expect[big]
    tol = 6 * np.sqrt(2.0 / (n_boot - 1))
    assert big.sum() > 0 and (np.abs(ratio - 1) < tol + 0.05).all(), (ratio.min(), ratio.max(), tol)
    # pooled: the mean of the ratios is far tighter than any single one
    assert abs(ratio.mean() - 1) < 6 * np.sqrt(2.0 / (n_boot - 1) / big.sum()) + 0.01
    cells = n * p >= 5                                 # chi-square approximation holds cell by cell
    dof = int(cells.sum()) - (1 if cells.all() else 0)
    pearson = (((counts[:, cells] - n * p[cells]) ** 2) / (n * p[cells])).sum(axis=1)
    if dof > 30:
        assert (np.abs(pearson - dof) < 6 * np.sqrt(2.0 * dof)).all(), (pearson.min(), pearson.max(), dof)
        assert abs(pearson.mean() - dof) < 6 * np.sqrt(2.0 * dof / n_boot) + 0.002 * dof
    # negative covariance between classes (-n p_i p_j): the two largest classes
    i, j = np.argsort(p)[-2:]
    cov = np.cov(counts[:, i], counts[:, j])[0, 1]
    sd = np.sqrt(n * p[i] * (1 - p[i]) * n * p[j] * (1 - p[j]) / n_boot)
    assert abs(cov + n * p[i] * p[j]) < 6 * sd * np.sqrt(2)


def test_bootstrap_draw_and_em(oracle, native_libs, monkeypatch):
    """The multinomial draw is only distributional (the reference draws from
    numpy's unseeded generator): totals exact, mean n*p within 6 sigma; the EM
    from the drawn counts equals the oracle EM on the same counts."""
    from seekmer_amd import infer
    rng = np.random.default_rng(8)
    n_tx, n_classes = 120, 500
    sizes = rng.integers(1, 5, n_classes)
    cls = np.repeat(np.arange(n_classes), sizes)
    tx = rng.integers(0, n_tx, cls.size)
    class_map = np.vstack([cls, tx]).astype(np.int64)
    class_count = rng.integers(1, 4000, n_classes).astype('f8')
    l = rng.uniform(100, 2000, n_tx)
    offsets, targets = infer._csr_from_class_map(class_map, n_classes)
    quant = infer._QuantHandle.from_csr(n_tx, offsets, targets, class_count)
    x0 = 1.0 / l
    x0 /= x0.sum()
    n_boot = 100
    out, counts, iters = quant.bootstrap(n_boot, 1234, x0, l, want_counts=True)
    out2, counts2, _ = quant.bootstrap(n_boot, 1234, x0, l, want_counts=True)
    # without the counts the replicates run eight at a time through the batched EM
    # (skm_em_batch.hip): the same draws, the same additions -- the same bits and step counts
    out8, _, iters8 = quant.bootstrap(n_boot, 1234, x0, l)
    out5, _, iters5 = quant.bootstrap(5, 1234, x0, l)             # (a short last group)
    # the working set is looked at every few steps and stopped replicates are replaced by the next
    # ones: whatever the interval (1: every step; 7, 40: several finish between two looks, or all
    # of them) and however the replicates meet in the working set -- the same bits again
    tpm8, _, _ = quant.bootstrap(n_boot, 1234, x0, l, tpm=True)
    for every, n in (('1', n_boot), ('7', 21), ('40', n_boot)):
        monkeypatch.setenv('SKM_BOOTSTRAP_CHUNK', every)
        late, _, iters_late = quant.bootstrap(n, 1234, x0, l)
        np.testing.assert_array_equal(late, out[:n])
        np.testing.assert_array_equal(iters_late, iters[:n])
        late_tpm, _, _ = quant.bootstrap(n, 1234, x0, l, tpm=True)
        np.testing.assert_array_equal(late_tpm, tpm8[:n])
    monkeypatch.delenv('SKM_BOOTSTRAP_CHUNK')
    quant.close()
    np.testing.assert_array_equal(counts, counts2)      # seeded: reproducible
    np.testing.assert_array_equal(out, out2)
    np.testing.assert_array_equal(out8, out)
    np.testing.assert_array_equal(iters8, iters)
    np.testing.assert_array_equal(out5, out[:5])
    np.testing.assert_array_equal(iters5, iters[:5])
    n = class_count.sum()
    assert (counts.sum(axis=1) == n).all()
    p = class_count / n
    mean = counts.mean(axis=0)
    sigma = np.sqrt(n * p * (1 - p) / n_boot)
    assert (np.abs(mean - n * p) < 6 * sigma + 1).all()
    assert len({c.tobytes() for c in counts}) == n_boot   # replicates differ
    _check_multinomial_dispersion(counts, class_count)
    for b in (0, n_boot - 1):
        x_ref, it_ref = oracle.em(x0, l, class_map, counts[b].astype('f8'))
        assert it_ref == iters[b]
        np.testing.assert_allclose(out[b], x_ref, rtol=1e-8, atol=1e-300)


def test_config4_bootstrap_on_a_mapped_table(oracle, native_libs, monkeypatch):
    """BASELINE.json configs[4] in shape (paired 2x100 reads mapped on the GPU, then `-b 100`,
    seekmer/infer.py:79-82, 108-111) at a size the oracle EM still handles: 1.2 M pairs are
    mapped, the class table stays in HBM, skm_quant_bootstrap draws 100 resamples from it.
    Totals are exact, first and second moments follow multinomial(n, count / n), and the EM on
    the drawn counts equals the oracle's EM on the same counts (iteration count equal, x within
    1e-8) for three replicates; the main estimate is the x0 of every replicate (:118)."""
    from seekmer_amd import synth, index_builder, mapper, common, infer
    ids, pool, tx_offsets = synth.transcriptome(4, 300)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    n_units = 1_200_000
    bases, offsets = synth.reads(4, pool, tx_offsets, 0, n_units, 100, True)
    result = mapper.MapResult(index)
    mapper.ReadMapper(index, result).map_batch(common.ReadBatch(n_units, bases, offsets, True))
    summarized = result.summarize()
    n_tx = len(ids)
    assert summarized.class_count.size > 1000 and summarized.aligned > 0.9 * n_units
    main = infer.quantify(summarized)
    eff = summarized.effective_lengths.astype('f8')
    x0 = main.copy()
    x0 /= x0.sum()
    quant = infer._QuantHandle.from_map_result(result, n_tx)
    n_boot = 100
    out, counts, iters = quant.bootstrap(n_boot, 20240, x0, eff, want_counts=True)
    out8, _, iters8 = quant.bootstrap(n_boot, 20240, x0, eff)    # batched EM: bit for bit the same
    np.testing.assert_array_equal(out8, out)
    np.testing.assert_array_equal(iters8, iters)
    monkeypatch.setenv('SKM_BOOTSTRAP_CHUNK', '11')                      # (another rhythm of looks and refills)
    out_late, _, iters_late = quant.bootstrap(n_boot, 20240, x0, eff)
    monkeypatch.delenv('SKM_BOOTSTRAP_CHUNK')
    np.testing.assert_array_equal(out_late, out)
    np.testing.assert_array_equal(iters_late, iters)
    # the handle holds the observed counts again afterwards
    x_main, it_main = quant.em(1.0 / eff / (1.0 / eff).sum(), eff)
    quant.close()
    class_count = summarized.class_count
    assert (counts.sum(axis=1) == class_count.sum()).all()
    mean = counts.mean(axis=0)
    n = class_count.sum()
    p = class_count / n
    assert (np.abs(mean - n * p) < 6 * np.sqrt(n * p * (1 - p) / n_boot) + 1).all()
    _check_multinomial_dispersion(counts, class_count)
    for b in (0, 41, n_boot - 1):
        x_ref, it_ref = oracle.em(x0, eff, summarized.class_map, counts[b].astype('f8'))
        assert it_ref == iters[b]
        np.testing.assert_allclose(out[b], x_ref, rtol=1e-8, atol=1e-300)
    x_ref, it_ref = oracle.em(1.0 / eff / (1.0 / eff).sum(), eff, summarized.class_map, class_count)
    assert it_ref == it_main
    np.testing.assert_allclose(x_main, x_ref, rtol=1e-8, atol=1e-300)
    # and through the module surface: bootstrap_quantify returns TPM vectors (sum 1e6)
    tpms = infer.bootstrap_quantify(summarized, main, 3, seed=20240)
    assert len(tpms) == 3 and all(abs(t.sum() - 1e6) < 1e-3 for t in tpms)
    np.testing.assert_array_equal(tpms[0], infer._tpm(out[0].copy()))   # numpy's sums restated on the device
    # `-b N` shared out over G ranks: rank r's share -- replicates r, r + G, ... of a
    # handle made from the merged table's arrays -- is, replicate by replicate, what the one-GPU loop
    # gives; with G = 1 the share IS that loop
    from seekmer_amd import parallel
    whole = np.asarray(infer.bootstrap_quantify(summarized, main, 11, seed=99))
    table = {'class_offsets': summarized.class_offsets, 'class_targets': summarized.class_targets,
             'class_count': summarized.class_count, 'effective_lengths': eff}
    np.testing.assert_array_equal(infer._bootstrap_share(table, main, 0, 1, 11, 99), whole)
    for world in (2, 3):
        for rank in range(world):
            first, step, count = parallel.replicate_share(11, rank, world)
            np.testing.assert_array_equal(infer._bootstrap_share(table, main, first, step, count, 99), whole[rank::world])
    # a handle that holds one rank's share of the classes must refuse to resample it (a communicator of
    # several ranks cannot be made on one GPU: the rule is checked on the handle's record of it)
    assert infer.bootstrap_ranks(summarized, main, 0, parallel.Ranks()) == []
    np.testing.assert_array_equal(np.asarray(infer.bootstrap_ranks(summarized, main, 11, parallel.Ranks(), seed=99)), whole)


def test_cli_end_to_end(oracle, native_libs, chr21, chr21_oracle_index, pairs21, tmp_path):
    """`seekmer_amd index -t` + `seekmer_amd infer -m -b 3` on the reference's
    own test data; abundance.tsv / run_info.json / readmap.txt against the oracle."""
    import json
    import shutil
    from conftest import GOLDEN
    from seekmer_amd import __main__ as cli
    gtf = tmp_path / 'empty.gtf'
    gtf.write_text('')
    index_path = tmp_path / 'index.npz'
    assert cli.main(['index', '-t', os.path.join(GOLDEN, 'human.cdna.21.fa.bz2'), str(gtf),
                     str(index_path)]) == 0
    out = tmp_path / 'out'
    assert cli.main(['infer', str(index_path), str(out), os.path.join(GOLDEN, '20_1.fastq'),
                     os.path.join(GOLDEN, '20_2.fastq'), '-m', '-b', '3', '--seed', '7']) == 0

    bases, offsets = oracle.pack_reads(pairs21)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, 21, True, fld)
    classes = oracle.Classes()
    classes.update(expected)
    class_map, class_count = classes.summarize()
    eff = oracle.effective_lengths(fld, chr21_oracle_index.lengths)
    tpm, _ = oracle.quantify(eff, class_map, class_count)
    est = oracle.est_counts(tpm, chr21_oracle_index.lengths, class_count.sum())

    rows = [line.rstrip('\n').split('\t') for line in (out / 'abundance.tsv').open()]
    assert rows[0] == ['target_id', 'length', 'eff_length', 'est_count', 'tpm']
    assert len(rows) == 1 + len(chr21[0])
    for i, row in enumerate(rows[1:]):
        assert row[0] == chr21[0][i].decode()
        assert row[1] == '%g' % len(chr21[1][i])
        assert row[2] == '%g' % np.float32(eff[i])
        assert abs(float(row[3]) - est[i]) <= 1e-4 * max(est[i], 1e-300) + 1e-6
        assert abs(float(row[4]) - tpm[i]) <= 1e-4 * max(tpm[i], 1e-300) + 1e-6
    info = json.load((out / 'run_info.json').open())
    assert info['n_targets'] == len(chr21[0]) and info['n_processed'] == 21
    assert info['n_pseudoaligned'] == 21 and info['n_bootstraps'] == 3
    unique = sum(int(c) for k, c in enumerate(class_count)
                 if (class_map[0] == k).sum() == 1)
    assert info['n_unique'] == unique
    readmap = (out / 'readmap.txt').read_text().splitlines()
    assert len(readmap) == 21
    tuples = expected.tuples()
    names = [line.strip()[1:].decode() for i, line in
             enumerate(open(os.path.join(GOLDEN, '20_1.fastq'), 'rb')) if i & 3 == 0]
    for line, name, t in zip(readmap, names, tuples):
        assert line.split('\t') == [name] + [chr21[0][i].decode() for i in t]
    arrays = np.load(out / 'abundance.npz')
    assert arrays['bootstrap/bs2'].shape == (len(chr21[0]),)
    np.testing.assert_array_equal(arrays['aux/fld'], fld.astype('i4'))
    # the same sample through the parallel FASTQ engine, page-locked slabs and three mapping threads
    out2 = tmp_path / 'out2'
    assert cli.main(['infer', str(index_path), str(out2), os.path.join(GOLDEN, '20_1.fastq'),
                     os.path.join(GOLDEN, '20_2.fastq'), '-j', '3', '--parse-threads', '2']) == 0
    assert (out2 / 'abundance.tsv').read_text() == (out / 'abundance.tsv').read_text()


def test_full_size_properties(oracle, native_libs):
    """Size-independent properties at a size the oracle would take minutes for:
    totals, unit-order invariance of the counter, batch-split invariance,
    single-ended fragment rule (every read counted)."""
    from seekmer_amd import synth, index_builder, mapper, common
    ids, pool, tx_offsets = synth.transcriptome(9, 400)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    n_units = 1_500_000
    bases, offsets = synth.reads(9, pool, tx_offsets, 0, n_units, 100, True)
    whole = mapper.MapResult(index)
    mapper.ReadMapper(index, whole).map_batch(common.ReadBatch(n_units, bases, offsets, True))
    c, rows, unaligned, total = whole.sizes()
    offs, targets, counts, first_seen, fld = whole.export()
    assert total == n_units and counts.sum() + unaligned == n_units
    assert fld[0] == 0 and fld.sum() <= counts.sum()
    assert (np.diff(first_seen) > 0).all()                    # first-seen order, unique
    assert unaligned < 0.05 * n_units

    # the same units in reverse order, in three uneven batches
    order = np.arange(n_units)[::-1]
    rev = bases[:-1].reshape(n_units, 200)[order].reshape(-1)
    rev = np.concatenate([rev, np.zeros(1, dtype=np.uint8)])
    other = mapper.MapResult(index)
    rm = mapper.ReadMapper(index, other)
    for lo, hi in ((0, 700_001), (700_001, 700_002), (700_002, n_units)):
        rm.map_batch(common.ReadBatch(hi - lo, rev, np.ascontiguousarray(offsets[2 * lo:2 * hi + 1]), True))
    assert other.sizes() == (c, rows, unaligned, total)
    np.testing.assert_array_equal(other.fragment_length_counts, fld)
    assert other.counter == whole.counter                      # multiset of (tuple -> count)

    single = mapper.MapResult(index)
    mapper.ReadMapper(index, single).map_batch(
        common.ReadBatch(2 * n_units, bases, offsets, False))
    assert single.fragment_length_counts.sum() == 2 * n_units  # every read counted


def _reference_cell_weights(transcripts, base_matrix, seed):
    """seekmer/impute.py:187-226 spelled out with numpy / sklearn only (nothing from
    seekmer_amd): i8 gene sums per cell, Pearson correlation, the off-diagonal correlations cut
    in two by 1-D 2-means, the upper cluster keeps its value."""
    import sklearn.cluster
    genes, transcript_gene_map = np.unique(transcripts['gene_id'], return_inverse=True)
    gene_matrix = np.zeros((len(base_matrix), len(genes)), dtype='i8')
    for i in range(len(genes)):
        gene_matrix[:, i] = base_matrix[:, transcript_gene_map == i].sum(axis=1)
    gene_matrix = gene_matrix[:, genes != b'']
    weights = np.corrcoef(gene_matrix)
    flattened = weights[(weights == weights) & (weights != 1.0)]
    kmean = sklearn.cluster.KMeans(2, random_state=seed)
    kmean.fit(flattened[:, None])
    weights[weights != weights] = 0.0
    keep = kmean.predict(weights.flatten()[:, None]) == kmean.cluster_centers_.argmax()
    return np.where(keep.reshape(weights.shape), weights, 0.0)


def test_impute_end_to_end(oracle, native_libs, tmp_path):
    """`seekmer impute` (seekmer/impute.py:54-125) on five small cells of two expression
    profiles: FASTQ files -> tpm.csv through the CLI, against the same pipeline spelled out on
    the oracle (per-cell class tables, pooled histogram, first-round EM, weights, blended
    counts, second-round EM).  The reference's own impute module cannot be run here (it imports
    logbook): parity of this stage is unpinned beyond the restated arithmetic."""
    import pandas
    from seekmer_amd import synth, index_builder, common
    from seekmer_amd.__main__ import main
    ids, pool, tx_offsets = synth.transcriptome(5, 30)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    transcripts = np.zeros(len(ids), dtype=[('transcript_id', index.transcripts.dtype['transcript_id']),
                                            ('gene_id', 'S8'), ('length', 'f8')])
    transcripts['transcript_id'] = index.transcripts['transcript_id']
    transcripts['length'] = index.transcripts['length']
    transcripts['gene_id'] = [b'GENE%04d' % (t // 4) for t in range(len(ids))]
    index = common.KMerIndex(index.kmers, index.contigs, index.sequences, index.targets, transcripts,
                             index.exons)
    index_path = tmp_path / 'index.npz'
    index.save(index_path)
    oindex = oracle.OracleIndex(index.kmers, index.contigs, index.sequences, index.targets,
                                lengths=np.diff(tx_offsets))
    n_units, read_len, paths, cells = 6000, 75, [], []
    for cell in range(5):
        bases, offsets = synth.reads(100 + cell % 2, pool, tx_offsets, cell * n_units, n_units, read_len, True)
        reads = bases[:-1].reshape(n_units, 2, read_len)
        for mate in (0, 1):
            path = tmp_path / ('cell%d_%d.fastq' % (cell, mate + 1))
            path.write_bytes(b''.join(b'@c%d/%d\n%s\n+\n%s\n' % (i, mate + 1, reads[i, mate].tobytes(),
                                                                  b'I' * read_len) for i in range(n_units)))
            paths.append(path)
        cells.append((bases, offsets))
    out = tmp_path / 'out'
    assert main(['impute', str(index_path), str(out), *map(str, paths), '-p', '4', '--seed', '0']) == 0
    table = pandas.read_csv(out / 'tpm.csv', index_col=0)
    assert table.shape == (len(ids), 5) and list(table.columns) == [str(p) for p in paths[::2]]

    # the same on the oracle
    fld_total = np.zeros(2000, dtype=np.int64)
    tables = []
    for bases, offsets in cells:
        fld = np.zeros(2000, dtype=np.int64)
        result = oracle.map_batch(oindex, bases, offsets, n_units, True, fld)
        classes = oracle.Classes()
        classes.update(result)
        tables.append(classes.summarize())
        fld_total += fld
    eff = oracle.effective_lengths(fld_total, oindex.lengths)
    base = np.asarray([oracle.quantify(eff, class_map, class_count)[0] for class_map, class_count in tables])
    weight = _reference_cell_weights(index.transcripts, base, seed=0) ** 4
    np.testing.assert_allclose(pandas.read_csv(out / 'weight.csv', index_col=0).to_numpy() ** 4, weight,
                               rtol=1e-9, atol=0)
    shifted, start = [], 0
    for class_map, _ in tables:
        shifted.append(np.vstack([class_map[0] + start, class_map[1]]))
        start += int(class_map[0].max()) + 1
    blended_map = np.concatenate(shifted, axis=1)
    for cell in range(5):
        total = tables[cell][1].sum()
        counts = np.concatenate([c * w * total / c.sum() for (_, c), w in zip(tables, weight[cell])])
        expected = oracle.quantify(eff, blended_map, counts)[0]
        got = table.iloc[:, cell].to_numpy()
        mask = expected > 0
        np.testing.assert_array_equal(got > 0, mask)
        assert (np.abs(got[mask] - expected[mask]) / expected[mask]).max() < 1e-4


@pytest.fixture(scope='module')
def t190k(native_libs):
    """The ~190k-transcript stand-in of BASELINE.json's ENSEMBL index (2 GiB k-mer table), built
    once for the full-size tests of configs[1], [3] and [4]."""
    from seekmer_amd import synth, index_builder
    ids, pool, tx_offsets = synth.transcriptome(1, 20000)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    assert index.kmers.size == 1 << 27 and len(ids) > 150000
    return ids, pool, tx_offsets, index


def _upload(native, array):
    import ctypes
    hip = native.hip()
    pointer = ctypes.c_void_p()
    native.check(hip.skm_device_malloc(0, array.nbytes, ctypes.byref(pointer)))
    native.check(hip.skm_device_upload(0, pointer, array.ctypes.data, array.nbytes))
    return pointer


@pytest.fixture(scope='module')
def t190k_oracle_index(oracle, t190k):
    ids, pool, tx_offsets, index = t190k
    return oracle.OracleIndex(index.kmers, index.contigs, index.sequences, index.targets,
                              lengths=np.diff(tx_offsets))


@pytest.mark.parametrize('config,seed,read_len,paired', [(1, 1, 100, True), (3, 3, 150, False)])
def test_t190k_slices_against_the_oracle(oracle, native_libs, t190k, t190k_oracle_index, config, seed,
                                         read_len, paired):
    """The benchmarked scale, unit by unit.  The first 300 k units of BASELINE.json configs[1]'s
    read set (2x100 pairs) and of configs[3]'s (150-base single-end reads) against the index the
    bench line is quoted on (2^27 slots, 4 GiB bucket copy, > 64-target mask extensions) through
    the production kernels AND through the counting build, each compared with the oracle on the
    same index arrays: per-unit span, anchor and signed target list, the class table in first-seen
    order, the fragment-length histogram, the quantification (iteration count equal, TPM <= 1e-4:
    north_star's tolerance), and every access counter of the counting build -- the numerator of
    bench.py's roofline.achieved -- field by field (seekmer/_common.pyx:54-97, _mapper.pyx:111-343)."""
    from seekmer_amd import synth, mapper, common, infer
    ids, pool, tx_offsets, index = t190k
    oindex = t190k_oracle_index
    n_units = 300_000
    bases, offsets = synth.reads(seed, pool, tx_offsets, 0, n_units, read_len, paired)
    fld = np.zeros(2000, dtype=np.int64)
    ostats = oracle.Stats()
    expected = oracle.map_batch(oindex, bases, offsets, n_units, paired, fld, stats=ostats)
    assert index.device_info()['bucketed'] == 1               # the production probe, not the fallback
    result, units = _run_gpu(index, bases, offsets, n_units, paired)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)

    classes = oracle.Classes()
    classes.update(expected)
    class_map, class_count = classes.summarize()
    eff = oracle.effective_lengths(fld, oindex.lengths)
    tpm_ref, iters_ref = oracle.quantify(eff, class_map, class_count)
    tpm_dev, iters_dev, eff_dev = infer.quantify_resident(result, return_iters=True,
                                                          return_effective_lengths=True)
    np.testing.assert_array_equal(eff_dev, eff)
    assert iters_dev == iters_ref, (iters_dev, iters_ref)
    mask = tpm_ref > 0
    np.testing.assert_array_equal(tpm_dev > 0, mask)
    rel = np.abs(tpm_dev[mask] - tpm_ref[mask]) / tpm_ref[mask]
    assert rel.max() < 1e-4, rel.max()

    counting = mapper.MapResult(index, keep_spans=True)
    counting.set_stats(True)
    rm = mapper.ReadMapper(index, counting)
    rm.map_batch(common.ReadBatch(n_units, bases, offsets, paired))
    _compare_units(expected, rm.last_batch(n_units))
    _compare_tables(oracle, expected, fld, counting)
    counted = counting.access_stats()
    for name, value in ostats.as_dict().items():
        assert counted[name] == value, (name, counted[name], value)
    assert (counted['read_bases'] + 16 * counted['slots'] + 48 * counted['contig_reads']
            + 8 * (counted['targets_copied'] + counted['targets_merged']) + 8 * counted['seq_fetches']
            + 4 * counted['tuple_ids']) == ostats.algorithmic_bytes()


def test_baseline_config2_properties(oracle, native_libs, t190k):
    """BASELINE.json configs[1] at its full size -- the ~190k-transcript stand-in index
    (2 GiB k-mer table) and 10 M 2x100 pairs, where the oracle would need a minute per run --
    through properties that do not need it: totals, first-seen order, the same counter from two
    uneven batches, from a second run (the scheduler is asynchronous, the integers must not care) and
    from the same reads packed on the host and pushed in pieces, and the device quantification
    against the numpy one bit for bit."""
    from seekmer_amd import synth, mapper, common, infer
    ids, pool, tx_offsets, index = t190k
    n_units = 10_000_000
    bases, offsets = synth.reads(1, pool, tx_offsets, 0, n_units, 100, True)
    whole = mapper.MapResult(index)
    mapper.ReadMapper(index, whole).map_batch(common.ReadBatch(n_units, bases, offsets, True))
    c, rows, unaligned, total = whole.sizes()
    offs, targets, counts, first_seen, fld = whole.export()
    assert total == n_units and counts.sum() + unaligned == n_units
    assert fld[0] == 0 and 0 < fld.sum() <= counts.sum()
    assert (np.diff(first_seen) > 0).all() and first_seen[0] >= 0 and first_seen[-1] < n_units
    assert unaligned < 0.03 * n_units and (targets >= 0).all() and targets.max() < len(ids)
    assert (np.diff(offs) > 0).all() and offs[-1] == rows

    split = mapper.MapResult(index)
    rm = mapper.ReadMapper(index, split)
    cut = 3_333_333
    rm.map_batch(common.ReadBatch(cut, bases[:offsets[2 * cut] + 1], offsets[:2 * cut + 1], True))
    rest = offsets[2 * cut:] - offsets[2 * cut]
    rm.map_batch(common.ReadBatch(n_units - cut, bases[offsets[2 * cut]:], np.ascontiguousarray(rest), True))
    offs2, targets2, counts2, first2, fld2 = split.export()
    np.testing.assert_array_equal(offs2, offs)          # same classes, same (first-seen) order
    np.testing.assert_array_equal(targets2, targets)
    np.testing.assert_array_equal(counts2, counts)
    np.testing.assert_array_equal(first2, first_seen)
    np.testing.assert_array_equal(fld2, fld)

    # the same reads packed on the host (2-bit code words, a piece per 1.5 M reads and stream, mate 2
    # pieces cut elsewhere than mate 1's) and pushed: the table of the ASCII batch bit for bit
    split.reset()
    flat = bases[:2 * n_units * 100].reshape(n_units, 2, 100)
    for s, size in ((0, 1_500_000), (1, 1_300_001)):
        mate = np.ascontiguousarray(flat[:, s, :]).reshape(-1)
        for lo in range(0, n_units, size):
            hi = min(n_units, lo + size)
            rm.push_packed(common.PackedReads.from_ascii(
                np.concatenate([mate[100 * lo:100 * hi], np.zeros(1, np.uint8)]),
                np.arange(hi - lo + 1, dtype=np.int64) * 100, stream=s, first_read=lo, paired=True))
    for got, want in zip(split.export(), (offs, targets, counts, first_seen, fld)):
        np.testing.assert_array_equal(got, want)
    assert split.sizes() == whole.sizes()

    tpm_dev, iters_dev = infer.quantify_resident(whole, return_iters=True)
    tpm_again, iters_again = infer.quantify_resident(split, return_iters=True)
    assert iters_dev == iters_again
    np.testing.assert_array_equal(tpm_dev, tpm_again)   # no floating-point atomics anywhere
    tpm_host, iters_host = infer.quantify(whole.summarize(), return_iters=True)
    assert iters_host == iters_dev
    np.testing.assert_array_equal(tpm_host, tpm_dev)
    assert abs(tpm_dev.sum() - 1e6) < 1e-3


def test_baseline_config3_properties(native_libs, t190k):
    """BASELINE.json configs[3] at its full size: 50 M single-end 150 bp reads (`-s`) against
    the same index, one resident batch.  Every single-ended read is counted in the
    fragment-length histogram -- unmapped ones at 25 (_mapper.pyx:90-94) -- the
    table does not depend on how the sample is cut into batches, two runs agree bit for bit, and
    the device quantification equals the numpy one."""
    from seekmer_amd import synth, mapper, infer, _native
    ids, pool, tx_offsets, index = t190k
    n_units, read_len = 50_000_000, 150
    bases, offsets = synth.reads(3, pool, tx_offsets, 0, n_units, read_len, False)
    d_bases, d_offsets = _upload(_native, bases), _upload(_native, offsets)
    whole = mapper.MapResult(index)
    whole.map_resident(d_bases, d_offsets, n_units, False, read_len)
    c, rows, unaligned, total = whole.sizes()
    offs, targets, counts, first_seen, fld = whole.export()
    assert total == n_units and counts.sum() + unaligned == n_units
    assert fld.sum() == n_units and fld[:25].sum() == 0 and fld[25] >= unaligned      # every read counted
    assert fld[read_len + 1:].sum() == 0                         # a single read spans at most its length
    assert (np.diff(first_seen) > 0).all() and unaligned < 0.03 * n_units
    assert (targets >= 0).all() and targets.max() < len(ids) and offs[-1] == rows
    # the same sample in three uneven resident batches (device pointers into the same buffers)
    import ctypes
    split = mapper.MapResult(index)
    cuts = [0, 11_111_111, 11_111_112, n_units]
    hip = _native.hip()
    for k in range(3):
        lo, hi = cuts[k], cuts[k + 1]
        sub = np.ascontiguousarray(offsets[lo:hi + 1] - offsets[lo])
        d_sub = _upload(_native, sub)
        split.map_resident(ctypes.c_void_p(d_bases.value + int(offsets[lo])), d_sub, hi - lo, False, read_len)
        _native.check(hip.skm_device_free(0, d_sub))
    for got, want in zip(split.export(), (offs, targets, counts, first_seen, fld)):
        np.testing.assert_array_equal(got, want)
    tpm, iters = infer.quantify_resident(whole, return_iters=True)
    tpm2, iters2 = infer.quantify_resident(split, return_iters=True)
    assert iters == iters2 and np.array_equal(tpm, tpm2) and abs(tpm.sum() - 1e6) < 1e-3
    tpm_host, iters_host = infer.quantify(whole.summarize(), return_iters=True)
    assert iters_host == iters and np.array_equal(tpm_host, tpm)
    _native.check(hip.skm_device_free(0, d_bases))
    _native.check(hip.skm_device_free(0, d_offsets))


def test_baseline_config4_properties(native_libs, t190k):
    """BASELINE.json configs[4] at its full size: 20 M 2x100 bp pairs mapped, then `-b 100` on the
    resident class table (GPU-side multinomial resampling + EM from the main estimate,
    seekmer/infer.py:79-82, 108-111).  Bootstrap totals are exact, the draws are reproducible
    from the seed and differ between replicates, their mean and dispersion follow
    multinomial(n, count / n) on the populated classes, every replicate is a TPM vector, and the
    handle holds the observed counts again afterwards.  (EM-on-drawn-counts against the oracle:
    test_config4_bootstrap_on_a_mapped_table, at a size the oracle handles.)"""
    from seekmer_amd import synth, mapper, infer, _native
    ids, pool, tx_offsets, index = t190k
    n_units, n_boot = 20_000_000, 100
    bases, offsets = synth.reads(4, pool, tx_offsets, 0, n_units, 100, True)
    d_bases, d_offsets = _upload(_native, bases), _upload(_native, offsets)
    result = mapper.MapResult(index)
    result.map_resident(d_bases, d_offsets, n_units, True, 100)
    tpm, iters, eff = infer.quantify_resident(result, return_iters=True, return_effective_lengths=True)
    class_count = result.export()[2].astype('f8')
    n = class_count.sum()
    assert n == result.sizes()[3] - result.sizes()[2]
    # the table is a large one -- from the generator, not from a run: reads come from the transcripts of
    # at least 300 bases, nearly all of them are drawn at this depth, and the isoforms of
    # a gene differ by construction, so there are more classes than such transcripts
    assert class_count.size > int((np.diff(tx_offsets) >= 300).sum())
    quant = infer._QuantHandle.from_map_result(result, len(ids))
    x0 = tpm / tpm.sum()
    out, counts, its = quant.bootstrap(n_boot, 99, x0, eff, want_counts=True)
    assert (counts.sum(axis=1) == n).all()                        # totals exact
    assert len({c.tobytes() for c in counts[:10]}) == 10
    out2, counts2, its2 = quant.bootstrap(3, 99, x0, eff, want_counts=True)
    np.testing.assert_array_equal(counts2, counts[:3])            # seeded
    np.testing.assert_array_equal(out2, out[:3])
    p = class_count / n
    mean = counts.mean(axis=0)
    assert (np.abs(mean - n * p) < 6 * np.sqrt(n * p * (1 - p) / n_boot) + 1).all()
    big = np.argsort(class_count)[-20000:]                        # the populated classes carry the test
    _check_multinomial_dispersion(np.concatenate([counts[:, big], (n - counts[:, big].sum(axis=1))[:, None]], axis=1),
                                  np.concatenate([class_count[big], [n - class_count[big].sum()]]))
    assert (its > 0).all() and np.isfinite(out).all() and (out >= 0).all()
    boot_tpm = infer._tpm(out[0].copy())
    assert abs(boot_tpm.sum() - 1e6) < 1e-3
    ok = tpm > 10
    assert np.median(np.abs(boot_tpm[ok] - tpm[ok]) / tpm[ok]) < 0.2   # a resample stays near the estimate
    x_again, it_again = quant.em(1.0 / eff / (1.0 / eff).sum(), eff)
    quant.close()
    assert it_again == iters and np.array_equal(infer._tpm(x_again), tpm)   # observed counts restored
    hip = _native.hip()
    _native.check(hip.skm_device_free(0, d_bases))
    _native.check(hip.skm_device_free(0, d_offsets))


@pytest.mark.parametrize('read_len,paired', [(150, False), (251, True), (33, True)])
def test_other_read_lengths(oracle, native_libs, read_len, paired):
    """BASELINE.json configs[3] shape (150 bp single-ended) and reads that need
    more than one 64-byte record / barely more than k bases."""
    from seekmer_amd import synth, index_builder
    ids, pool, tx_offsets = synth.transcriptome(11, 120)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    oindex = oracle.OracleIndex(index.kmers, index.contigs, index.sequences, index.targets,
                                lengths=np.diff(tx_offsets))
    n_units = 20000
    bases, offsets = synth.reads(11, pool, tx_offsets, 0, n_units, min(read_len, 299), paired)
    if read_len != min(read_len, 299):
        pytest.skip('generator limit')
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(oindex, bases, offsets, n_units, paired, fld)
    result, units = _run_gpu(index, bases, offsets, n_units, paired)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


def test_long_and_ragged_reads(oracle, native_libs, chr21, chr21_oracle_index):
    """Reads of 600-1500 bases cut from transcripts (several 64-byte records per read, long
    first-k-mer rolls and many contig jumps), mixed with very short ones in one batch, single-
    and pair-ended; then an empty batch and a batch of empty reads."""
    from seekmer_amd import common
    rng = np.random.default_rng(41)
    long_tx = [s for s in chr21[1] if len(s) > 1600]
    assert len(long_tx) > 20
    reads = []
    for i in range(600):
        s = long_tx[rng.integers(len(long_tx))]
        n = int(rng.integers(600, 1500))
        p = int(rng.integers(0, len(s) - n))
        r = bytearray(s[p:p + n].upper())
        for _ in range(int(rng.integers(0, 8))):                # a few substitutions
            r[int(rng.integers(len(r)))] = b'ACGT'[int(rng.integers(4))]
        reads.append(bytes(r) if i % 5 else bytes(r[:int(rng.integers(0, 40))]))
    index = make_product_index(chr21_oracle_index, chr21[0])
    bases, offsets = oracle.pack_reads(reads)
    for paired in (False, True):
        n_units = len(reads) // 2 if paired else len(reads)
        fld = np.zeros(2000, dtype=np.int64)
        expected = oracle.map_batch(chr21_oracle_index, bases, offsets, n_units, paired, fld)
        result, units = _run_gpu(index, bases, offsets, n_units, paired)
        _compare_units(expected, units)
        _compare_tables(oracle, expected, fld, result)
    # nothing in, nothing counted
    result, _ = _run_gpu(index, np.zeros(1, np.uint8), np.zeros(1, np.int64), 0, True)
    assert result.sizes() == (0, 0, 0, 0)
    empty_offsets = np.zeros(9, dtype=np.int64)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, np.zeros(1, np.uint8), empty_offsets, 4, True, fld)
    result, units = _run_gpu(index, np.zeros(1, np.uint8), empty_offsets, 4, True)
    _compare_units(expected, units)
    _compare_tables(oracle, expected, fld, result)


def test_module_surface(oracle, native_libs, chr21, chr21_oracle_index, pairs21, tmp_path):
    """The reference's module API on the device-backed MapResult: map_reads with
    worker threads and the Python feeders, Counter view, update(), clear(),
    merge_fragment_lengths(), harmonic mean, map_multiple_samples."""
    import collections
    import pathlib
    from conftest import GOLDEN
    from seekmer_amd import common, mapper
    index = make_product_index(chr21_oracle_index, chr21[0])
    paths = [pathlib.Path(GOLDEN) / '20_1.fastq', pathlib.Path(GOLDEN) / '20_2.fastq']

    bases, offsets = oracle.pack_reads(pairs21)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, 21, True, fld)
    want = collections.Counter(expected.tuples())

    for jobs in (1, 3):
        result = mapper.map_reads(index, common.feed_pair_ended_reads(*paths), job_count=jobs)
        assert result.counter == want
        np.testing.assert_array_equal(result.fragment_length_counts, fld)
        assert abs(result.harmonic_mean_fragment_length - oracle.harmonic_mean_fragment_length(fld)) < 1e-9
        summarized = result.summarize()
        assert (summarized.aligned, summarized.unaligned, summarized.total) == (21, 0, 21)
        assert summarized.class_map.dtype == np.int64 and summarized.class_map.shape[0] == 2
        np.testing.assert_array_equal(result.effective_lengths,
                                      oracle.effective_lengths(fld, chr21_oracle_index.lengths))

    # update() with tuples (Counter.update semantics), then clear() keeps the histogram
    result.update([b'x', b'y', b'z'], [(5, 6), (), (5, 6)])
    counter = result.counter
    assert counter[(5, 6)] == want.get((5, 6), 0) + 2 and counter[()] == 1
    assert result.sizes()[3] == 24
    result.merge_fragment_lengths(fld)
    np.testing.assert_array_equal(result.fragment_length_counts, 2 * fld)
    result.clear()
    assert result.counter == collections.Counter()
    np.testing.assert_array_equal(result.fragment_length_counts, 2 * fld)

    # many threads hammering one handle: integer results are order independent
    rng = np.random.default_rng(21)
    reads = _adversarial_reads(chr21[1], rng, 8000, 80)
    big_bases, big_offsets = oracle.pack_reads(reads)
    big_fld = np.zeros(2000, dtype=np.int64)
    big = oracle.map_batch(chr21_oracle_index, big_bases, big_offsets, 4000, True, big_fld)

    def feeder():
        for lo in range(0, 4000, 250):
            yield common.ReadBatch(250, big_bases, np.ascontiguousarray(big_offsets[2 * lo:2 * lo + 501]), True)
    threaded = mapper.map_reads(index, feeder(), job_count=6)
    assert threaded.counter == collections.Counter(big.tuples())
    np.testing.assert_array_equal(threaded.fragment_length_counts, big_fld)

    samples = mapper.map_multiple_samples(
        index, [common.feed_pair_ended_reads(*paths), common.feed_single_ended_reads(paths[0])], job_count=2)
    assert samples[0].counter == want
    single = oracle.map_batch(chr21_oracle_index, *oracle.pack_reads(pairs21[0::2]), 21, False)
    assert samples[1].counter == collections.Counter(single.tuples())


def test_async_batches_from_host_and_from_fastq(oracle, native_libs, tmp_path):
    """skm_mapper_map_batch_async: a sample handed over in pieces -- by one thread in order, by
    three threads in any order, and as FASTQ text through the native reader's parallel engine
    with page-locked slabs -- gives the table of the one-batch run bit for bit, class order
    included (every piece carries its place in the sample, `first_unit`); the oracle pins that
    table.  A piece with broken offsets fails the sync (not the process) and leaves the handle
    usable after a reset."""
    import threading
    from seekmer_amd import synth, index_builder, mapper, common
    ids, pool, tx_offsets = synth.transcriptome(6, 60)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    oindex = oracle.OracleIndex(index.kmers, index.contigs, index.sequences, index.targets,
                                lengths=np.diff(tx_offsets))
    n_units, read_len = 60000, 75
    bases, offsets = synth.reads(6, pool, tx_offsets, 0, n_units, read_len, True)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(oindex, bases, offsets, n_units, True, fld)
    whole = mapper.MapResult(index)
    mapper.ReadMapper(index, whole).map_batch(common.ReadBatch(n_units, bases, offsets, True))
    _compare_tables(oracle, expected, fld, whole)
    reference = whole.export()

    cut = [0, 7000, 7001, 25000, 41000, n_units]
    pieces = [common.ReadBatch(cut[k + 1] - cut[k], bases, offsets[2 * cut[k]:2 * cut[k + 1] + 1], True,
                               first_unit=cut[k]) for k in range(len(cut) - 1)]

    def same(result):
        for got, want in zip(result.export(), reference):
            np.testing.assert_array_equal(got, want)
        assert result.sizes() == whole.sizes()

    one = mapper.MapResult(index)
    rm = mapper.ReadMapper(index, one)
    for piece in pieces:
        rm.map_batch_async(piece)
    one.sync()
    same(one)
    # equal-length reads: the fixed-stride form (offsets never cross PCIe)
    one.reset()
    for piece in pieces:
        piece.uniform_len = read_len
        rm.map_batch_async(piece)
    same(one)
    for piece in pieces:
        piece.uniform_len = None

    many = mapper.MapResult(index)
    order = [3, 0, 4, 2, 1]
    threads = [threading.Thread(target=lambda ks=ks: [mapper.ReadMapper(index, many).map_batch_async(pieces[k])
                                                      for k in ks])
               for ks in (order[:2], order[2:4], order[4:])]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    same(many)                                           # (export waits for the queue)

    p1, p2 = tmp_path / 'r_1.fastq', tmp_path / 'r_2.fastq'
    synth.write_fastq(bases, n_units, read_len, True, p1, p2)
    for threads_, pinned in ((0, False), (3, True)):
        feeder = common.NativeReadFeeder([p1, p2], True, batch_units=9000, threads=threads_, pinned=pinned)
        text = mapper.map_reads(index, feeder, job_count=1 if threads_ == 0 else 3)
        assert feeder.parallel is (threads_ > 0)
        same(text)

    broken = offsets[:2 * 500 + 1].copy()
    broken[100] = broken[101] + 5
    one.reset()
    rm.map_batch_async(pieces[0])
    rm.map_batch_async(common.ReadBatch(500, bases, broken, True))
    with pytest.raises(ValueError):
        one.sync()
    one.reset()
    for piece in pieces:
        rm.map_batch_async(piece)
    same(one)


def _streams_of(common, bases, offsets, n_units, paired):
    """The reads of a flat batch as one packed piece per stream (mate 1 reads, mate 2 reads)."""
    mates = 2 if paired else 1
    lengths = np.diff(offsets)
    pieces = []
    for s in range(mates):
        sel = np.arange(s, mates * n_units, mates)
        sub_offsets = np.zeros(n_units + 1, dtype=np.int64)
        np.cumsum(lengths[sel], out=sub_offsets[1:])
        sub = np.concatenate([bases[offsets[r]:offsets[r + 1]] for r in sel] + [np.zeros(1, np.uint8)])
        pieces.append(common.PackedReads.from_ascii(sub, sub_offsets, stream=s, paired=paired))
    return pieces


def _cut(common, piece, borders):
    """`piece` cut at `borders` (unit numbers) into pieces of their own."""
    codes, lengths = piece.codes, piece.lengths
    exc_reads, exc_masks = piece.exceptions
    out = []
    for lo, hi in zip(borders[:-1], borders[1:]):
        sel = (exc_reads >= lo) & (exc_reads < hi)
        out.append(common.PackedReads.from_arrays(piece.stream, lo, codes[lo:hi], lengths[lo:hi],
                                                  exc_reads[sel] - lo, exc_masks[sel], paired=piece.paired))
    return out


@pytest.mark.parametrize('paired', [True, False])
def test_packed_reads_give_the_same_tables(oracle, native_libs, chr21, chr21_oracle_index, paired, tmp_path):
    """Reads packed on the host (2-bit code words + the bit planes of the reads with an N or a
    lower-case letter) and pushed piece by piece -- streams cut at different places, pushed in any
    order, interleaved with a stride, parsed from FASTQ text by the one-pass reader -- give the table
    of the ASCII batch bit for bit (classes, order, counts, FLD, unaligned), which the oracle pins."""
    from seekmer_amd import common, mapper
    rng = np.random.default_rng(77)
    reads = _adversarial_reads(chr21[1], rng, 6000, 100)
    reads += _adversarial_reads(chr21[1], rng, 600, 150)           # ragged: two record sizes in one sample
    index = make_product_index(chr21_oracle_index, chr21[0])
    bases, offsets = oracle.pack_reads(reads)
    mates = 2 if paired else 1
    n_units = len(reads) // mates
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, n_units, paired, fld)
    whole = mapper.MapResult(index)
    mapper.ReadMapper(index, whole).map_batch(common.ReadBatch(n_units, bases, offsets, paired))
    _compare_tables(oracle, expected, fld, whole)
    reference = whole.export()

    def same(result):
        for got, want in zip(result.export(), reference):
            np.testing.assert_array_equal(got, want)
        assert result.sizes() == whole.sizes()

    streams = _streams_of(common, bases, offsets, n_units, paired)
    assert sum(p.raw.n_exceptions for p in streams) > 500          # N / lower-case reads are there
    # one piece per stream
    one = mapper.MapResult(index)
    rm = mapper.ReadMapper(index, one)
    for piece in streams:
        rm.push_packed(piece)
    same(one)
    # the streams cut at different places, the pieces pushed in a shuffled order
    borders = [[0, 1, 700, 701, 2000, n_units], [0, 300, 1999, 2600, n_units]]
    pieces = [p for s, piece in enumerate(streams) for p in _cut(common, piece, borders[s])]
    for trial in range(3):
        order = rng.permutation(len(pieces))
        one.reset()
        for k in order:
            rm.push_packed(pieces[k])
        same(one)
    # three threads push their shares of the pieces and each syncs when it is done, while the others
    # are still pushing: a sync maps what has both mates and leaves the rest waiting for theirs
    import threading
    one.reset()
    order = rng.permutation(len(pieces)).tolist()

    def push_share(share):
        pusher = mapper.ReadMapper(index, one)
        for k in share:
            pusher.push_packed(pieces[k])
        one.sync()

    threads = [threading.Thread(target=push_share, args=(order[t::3],)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    same(one)
    # mate 1 has reads without a mate at the end: they are in no result (zip(file1, file2))
    if paired:
        one.reset()
        rm.push_packed(streams[0])
        rm.push_packed(_cut(common, streams[1], [0, n_units - 100])[0])
        assert one.sizes()[3] == n_units - 100
        # a piece that overlaps what its stream holds replaces the reads from there on
        one.reset()
        wrong = _cut(common, streams[0], [0, 500, 3000])[1]
        moved = common.PackedReads.from_arrays(1, 400, wrong.codes, wrong.lengths, *wrong.exceptions, paired=True)
        first, rest = _cut(common, streams[1], [0, 400, n_units])
        rm.push_packed(first)
        rm.push_packed(moved)               # mate 2 reads [400, 2900) that are not the sample's ...
        rm.push_packed(rest)                # ... replaced by the right ones before any mate 1 read arrives
        rm.push_packed(streams[0])
        same(one)
        # both mates in one array, [unit][mate][words]: two pieces over it with a stride
        cw = max(p.code_words for p in streams)
        inter = np.zeros((n_units, 2, cw), dtype=np.uint64)
        for s in range(2):
            inter[:, s, :streams[s].code_words] = streams[s].codes
        one.reset()
        for s in range(2):
            raw = native_libs.PackedReads()
            raw.stream, raw.code_words, raw.first_read, raw.n_reads = s, cw, 0, n_units
            raw.read_stride, raw.uniform_len = 2 * cw, -1
            raw.codes = inter.ctypes.data + 8 * cw * s
            lengths = np.ascontiguousarray(streams[s].lengths)
            exc_reads, exc_masks = streams[s].exceptions
            masks = np.zeros((exc_reads.size, cw), dtype=np.uint32)
            masks[:, :streams[s].code_words] = exc_masks
            raw.lengths = lengths.ctypes.data
            raw.n_exceptions = exc_reads.size
            raw.exception_reads = np.ascontiguousarray(exc_reads).ctypes.data
            raw.exception_masks = masks.ctypes.data
            rm.push_packed(common.PackedReads(raw, keep=(inter, lengths, exc_reads, masks), paired=True))
        same(one)
    # FASTQ text (N, lower case, CRLF line ends in one file) through the one-pass reader
    files = [tmp_path / ('r_%d.fastq' % s) for s in range(mates)]
    for s, path in enumerate(files):
        with open(path, 'wb') as f:
            end = b'\r\n' if s else b'\n'
            for u in range(n_units):
                read = reads[mates * u + s]
                f.write(b'@u%d extra' % u + end + read + end + b'+' + end + b'@' * len(read) + end)
    for threads, chunk, pinned in ((0, 0, False), (3, 50_000, True), (2, 1 << 20, False)):
        feeder = common.PackedReadFeeder(files, paired, threads=threads, chunk_bytes=chunk, pinned=pinned)
        text = mapper.map_reads(index, feeder, job_count=1)
        same(text)
        assert feeder.stats['units'] == n_units and feeder.stats['reparsed'] == 0
        # ... and piece by piece through Python
        one.reset()
        rm(common.PackedReadFeeder(files, paired, threads=threads, chunk_bytes=chunk))
        same(one)
    # a read longer than its code words hold fails the sync, not the process
    one.reset()
    bad = common.PackedReads.from_arrays(0, 0, np.zeros((40000, 2), dtype=np.uint64), np.full(40000, 65, dtype=np.uint32),
                                         paired=False)
    bad.raw.uniform_len = -1
    if not paired:
        rm.push_packed(bad)
        with pytest.raises(ValueError):
            one.sync()
        one.reset()
        for piece in streams:
            rm.push_packed(piece)
        same(one)


def test_pairs_of_files_of_unequal_length_and_a_pusher_at_the_byte_limit(oracle, native_libs, chr21, chr21_oracle_index,
                                                                         tmp_path, monkeypatch):
    """Two things the packed path must get right whatever the timing.
    (a) Several pairs of FASTQ files where a mate-2 file is the longer one: zip(file1, file2)
        (seekmer/common.py:180-197) ends at the shorter file, so the longer file's leftover reads must be
        cut off before the next pair of files delivers reads of the other stream -- they would meet as
        mates otherwise.  The one-pass reader sends a cut (SKM_PACKED_CUT) first; the table is the one of
        the Python feeders' batches, natively drained and piece by piece, for every chunk size.
    (b) A pusher that is held back by the byte limit while the only mappable run is shorter than the
        worker's minimum: the run is mapped (nobody else could extend it) instead of everyone waiting."""
    from seekmer_amd import common, mapper
    rng = np.random.default_rng(5)
    index = make_product_index(chr21_oracle_index, chr21[0])
    reads = _adversarial_reads(chr21[1], rng, 2 * 4000, 100)
    sizes = [(1500, 1700), (900, 600), (1000, 1300)]            # records of (mate 1 file, mate 2 file) per pair of files
    files, kept, at = [], [], 0
    for k, (n1, n2) in enumerate(sizes):
        for s, n in enumerate((n1, n2)):
            path = tmp_path / ('p%d_%d.fastq' % (k, s + 1))
            with open(path, 'wb') as f:
                for u in range(n):
                    read = reads[2 * (at + u) + s]
                    f.write(b'@r%d\n' % u + read + b'\n+\n' + b'I' * len(read) + b'\n')
            files.append(path)
        for u in range(min(n1, n2)):
            kept += [reads[2 * (at + u)], reads[2 * (at + u) + 1]]
        at += max(n1, n2)
    n_units = len(kept) // 2
    bases, offsets = oracle.pack_reads(kept)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, n_units, True, fld)
    for threads, chunk in ((0, 0), (3, 20_000), (2, 150_000)):
        drained = mapper.map_reads(index, common.PackedReadFeeder(files, True, threads=threads, chunk_bytes=chunk), job_count=1)
        _compare_tables(oracle, expected, fld, drained)
        stepwise = mapper.MapResult(index)
        mapper.ReadMapper(index, stepwise)(common.PackedReadFeeder(files, True, threads=threads, chunk_bytes=chunk))
        _compare_tables(oracle, expected, fld, stepwise)
    # the Python feeders agree (they are the reference's zip)
    python_fed = mapper.map_reads(index, common.feed_pair_ended_reads(*files), job_count=1)
    _compare_tables(oracle, expected, fld, python_fed)

    # (b) 2 KB of pieces may wait; mate 2 delivers 50 reads, then mate 1 everything, then mate 2 the rest
    monkeypatch.setenv('SKM_TEST_PACKED_MAX_PENDING', '2048')
    limited = mapper.MapResult(index)
    rm = mapper.ReadMapper(index, limited)
    streams = _streams_of(common, bases, offsets, n_units, True)
    head, tail = _cut(common, streams[1], [0, 50, n_units])
    rm.push_packed(head)
    for piece in _cut(common, streams[0], list(range(0, n_units, 400)) + [n_units]):
        rm.push_packed(piece)                 # (hung here before: 50 < PACKED_MIN_UNITS and nobody flushing)
    rm.push_packed(tail)
    _compare_tables(oracle, expected, fld, limited)


def test_shares_of_a_sample_through_the_packed_reader(oracle, native_libs, chr21, chr21_oracle_index, tmp_path):
    """What several ranks do with plain FASTQ files, on one GPU: every "rank" (a thread here, the
    all-reduce a barrier) reads its run of the sample's units through the one-pass packed reader
    (PackedReadFeeder(shard=...), skm_fastq_packed_open_ranges) into a mapper of its own, natively
    drained; the tables merged on the device are the oracle's table of the whole sample -- first-seen
    order included, i.e. the pieces carried the unit numbers of a one-process run.  Pairs of files of
    unequal length, 3 ranks."""
    import test_packed_reads as packed
    from seekmer_amd import common, mapper
    rng = np.random.default_rng(77)
    index = make_product_index(chr21_oracle_index, chr21[0])
    reads = _adversarial_reads(chr21[1], rng, 2 * 5000, 100)
    sizes = [(2100, 2300), (1900, 1500)]
    files, kept, at = [], [], 0
    for k, (n1, n2) in enumerate(sizes):
        for s, n in enumerate((n1, n2)):
            path = tmp_path / ('q%d_%d.fastq' % (k, s + 1))
            with open(path, 'wb') as f:
                for u in range(n):
                    read = reads[2 * (at + u) + s]
                    f.write(b'@r%d\n' % u + read + b'\n+\n' + b'I' * len(read) + b'\n')
            files.append(path)
        for u in range(min(n1, n2)):
            kept += [reads[2 * (at + u)], reads[2 * (at + u) + 1]]
        at += max(n1, n2)
    n_units = len(kept) // 2
    bases, offsets = oracle.pack_reads(kept)
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, n_units, True, fld)
    world = 3
    ranks = packed._ThreadRanks(world)
    feeders = [common.PackedReadFeeder(files, True, threads=2, chunk_bytes=30_000, shard=(r, world),
                                       sum_over_ranks=ranks.sum_int64) for r in range(world)]
    import threading
    failures = []

    def locate(feeder):
        try:
            feeder.COUNT_CHUNK = 50_000
            feeder.locate_share()
        except BaseException as error:
            failures.append(error)
            ranks.barrier.abort()

    workers = [threading.Thread(target=locate, args=(f,)) for f in feeders]
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    assert not failures, failures
    assert [f.share[2] for f in feeders] == [n_units * r // world for r in range(world)]
    shares = [mapper.map_reads(index, f, job_count=1) for f in feeders]
    assert [s.sizes()[3] for s in shares] == [f.share[3] for f in feeders]
    for other in shares[1:]:
        shares[0].merge_resident(other)
    _compare_tables(oracle, expected, fld, shares[0])


def test_tables_merge_on_the_device(oracle, native_libs, chr21, chr21_oracle_index):
    """SURVEY 8(e).1 without the host: a mapper's class table where it lies in HBM
    (skm_mapper_device_table: registry order, counts as doubles, the arena) merged into another
    mapper on the same GPU (skm_mapper_merge_device) -- what a hand-over between GPUs does after
    copying those arrays over xGMI -- against the host-array merge (skm_mapper_merge over
    skm_mapper_export), against one mapper that saw the whole sample, and against the oracle: the
    same classes in the same first-seen order, counts, histogram and totals, bit for bit, whichever
    of the three shares is the receiver."""
    from seekmer_amd import common, mapper, parallel
    rng = np.random.default_rng(41)
    reads = _adversarial_reads(chr21[1], rng, 2 * 6000, 100)
    index = make_product_index(chr21_oracle_index, chr21[0])
    bases, offsets = oracle.pack_reads(reads)
    n_units = 6000
    fld = np.zeros(2000, dtype=np.int64)
    expected = oracle.map_batch(chr21_oracle_index, bases, offsets, n_units, True, fld)
    cuts = [0, 2500, 2501, n_units]

    def shares():
        out = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            share = mapper.MapResult(index)
            sub = np.ascontiguousarray(offsets[2 * lo:2 * hi + 1])
            mapper.ReadMapper(index, share).map_batch_async(common.ReadBatch(hi - lo, bases, sub, True, first_unit=lo))
            share.sync()                  # (first-seen values are global unit numbers: first_unit)
            out.append(share)
        return out

    by_host = shares()
    parallel.merge_into(by_host[0], [parallel.rank_table(s) for s in by_host[1:]])
    _compare_tables(oracle, expected, fld, by_host[0])
    reference = by_host[0].export()
    for receiver in range(3):
        on_device = shares()
        for k, other in enumerate(on_device):
            if k != receiver:
                on_device[receiver].merge_resident(other)
        for got, want in zip(on_device[receiver].export(), reference):       # (first-seen values included)
            np.testing.assert_array_equal(got, want)
        assert on_device[receiver].sizes() == by_host[0].sizes()
    # an empty table on either side
    empty = mapper.MapResult(index)
    empty.merge_resident(by_host[0])
    for got, want in zip(empty.export(), reference):
        np.testing.assert_array_equal(got, want)
    by_host[0].merge_resident(mapper.MapResult(index))
    for got, want in zip(by_host[0].export(), reference):
        np.testing.assert_array_equal(got, want)



def test_table_hand_over_through_rccl(native_libs):
    """skm_mapper_exchange_tables: a mapper's table sent as it lies in HBM (ncclSend / ncclRecv) and
    merged by key on the receiving side -- on one GPU a rank sends to itself, which drives the
    whole data path (header, six arrays, merge kernel): the receiving mapper then holds the table of
    the whole sample, first-seen order included (scripts/micro/exchange_self.py asserts it).  In a
    process of its own and under a time limit: an unmatched send would wait for ever."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts', 'micro', 'exchange_self.py')
    done = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert done.returncode == 0 and b'exchange ok' in done.stdout, done.stdout.decode()[-2000:]
