"""The N > 1 path on CPU: world_size 2, gloo, driving the PRODUCT's host-side rank logic --
`parallel.Ranks.from_env`, the sharded `NativeReadFeeder`, `infer.finish` (quantify over the
rank-local tables first, then tables to rank 0, merge, summarize), `parallel.rank_table /
merge_into`.  There is no GPU here, so the two device-backed pieces are stood in for by the
oracle: a MapResult whose table is filled by the oracle's mapper, and a `quantify_ranks` that
runs the reference's EM with one all-reduce of the per-transcript numerators per step -- the
collective pattern of skm_quant_infer(comm).  Rank 0's results must equal one process mapping the
whole sample: class table bit for bit (order included), histogram, counts, and the TPM of the
sharded EM against the oracle's EM on the merged table (same iteration count, 1e-9).  `-b N` over the
ranks (infer.bootstrap_ranks: the merged table broadcast, rank r runs replicates r, r + G, ...,
results gathered in replicate order) with the oracle's EM as the per-rank piece equals the
one-rank loop replicate by replicate."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


N_UNITS, READ_LEN, BATCH = 6001, 75, 700


def _sample(tmp):
    """(oracle index, transcript pool, FASTQ paths) of the test sample, written once."""
    from oracle import oracle as O
    from seekmer_amd import synth
    ids, pool, tx_offsets = synth.transcriptome(4, 30)
    index = O.build_index(synth.sequences_of(pool, tx_offsets))
    p1, p2 = os.path.join(tmp, 'r_1.fastq'), os.path.join(tmp, 'r_2.fastq')
    if not os.path.exists(p2):
        bases, _ = synth.reads(4, pool, tx_offsets, 0, N_UNITS, READ_LEN, True, n_threads=1)
        synth.write_fastq(bases, N_UNITS, READ_LEN, True, p1, p2, n_threads=1)
    return index, (p1, p2)


class OracleMapResult:
    """The MapResult surface infer.finish / parallel use, with the oracle doing the mapping."""

    def __init__(self, O, index):
        self.O, self.index = O, index
        self.table = {}                       # tuple -> [count, first_seen]
        self.unaligned = 0
        self.fld = np.zeros(2000, dtype=np.int64)

    def map_batch(self, batch):
        fld = np.zeros(2000, dtype=np.int64)
        result = self.O.map_batch(self.index, batch.bases, batch.offsets, batch.count, batch.paired, fld)
        self.fld += fld
        for u, t in enumerate(result.tuples()):
            if not t:
                self.unaligned += 1
                continue
            entry = self.table.setdefault(t, [0, batch.first_unit + u])
            entry[0] += 1
            entry[1] = min(entry[1], batch.first_unit + u)

    def _ordered(self):
        return sorted(self.table.items(), key=lambda kv: kv[1][1])

    def export(self):
        rows = self._ordered()
        offsets = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum([len(k) for k, _ in rows], out=offsets[1:])
        targets = np.asarray([t for k, _ in rows for t in k], dtype=np.int32)
        return (offsets, targets, np.asarray([v[0] for _, v in rows], dtype=np.int64),
                np.asarray([v[1] for _, v in rows], dtype=np.int64), self.fld.copy())

    def sizes(self):
        return (len(self.table), sum(len(k) for k in self.table), self.unaligned,
                self.unaligned + sum(v[0] for v in self.table.values()))

    def merge_table(self, offsets, targets, counts, first_seen, unaligned, fld):
        ids = np.asarray(targets).tolist()
        for k in range(len(counts)):
            key = tuple(ids[offsets[k]:offsets[k + 1]])
            entry = self.table.setdefault(key, [0, int(first_seen[k])])
            entry[0] += int(counts[k])
            entry[1] = min(entry[1], int(first_seen[k]))
        self.unaligned += int(unaligned)
        self.fld += np.asarray(fld, dtype=np.int64)

    def summarize(self):
        from seekmer_amd import mapper
        offsets, targets, counts, _, fld = self.export()
        class_ids = np.repeat(np.arange(counts.size, dtype=np.int64), np.diff(offsets))
        aligned = int(counts.sum())
        return mapper.SummarizedResult(
            aligned=aligned, unaligned=self.unaligned, total=aligned + self.unaligned,
            class_map=np.vstack([class_ids, targets.astype(np.int64)]), class_count=counts.astype('f8'),
            fragment_length_frequencies=fld, effective_lengths=self.O.effective_lengths(fld, self.index.lengths),
            class_offsets=offsets, class_targets=targets)


def _quantify_ranks(O, index, ranks):
    """quantify() over rank-local tables: the histogram and n all-reduced once, the numerators
    all-reduced every step, the stopping rule judged redundantly (seekmer/infer.py:88-168)."""
    import torch

    def allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a).copy())
        ranks.dist.all_reduce(t)
        return t.numpy()

    def run(result):
        offsets, targets, counts, _, fld = result.export()
        eff = O.effective_lengths(allreduce(fld), index.lengths)
        n = float(allreduce(np.asarray([counts.sum()], dtype='f8'))[0])
        cls = np.repeat(np.arange(counts.size), np.diff(offsets))
        x = 1.0 / eff
        x /= x.sum()
        steps = 0
        while True:
            w = x[targets]
            inner = np.bincount(cls, weights=w, minlength=counts.size) / counts
            local = np.bincount(targets, weights=w / inner[cls], minlength=eff.size)
            new = allreduce(local) / eff / n
            new[new != new] = 0
            steps += 1
            change = (np.absolute(new - x) / new)[new > 1e-8].max()
            x = new
            if not change > 0.01:
                break
        x /= x.sum() / 1000000
        x[x < 0.001] = 0
        x /= x.sum() / 1000000
        run.steps = steps
        return x
    return run


N_BOOT, BOOT_SEED = 7, 12345


def _oracle_share(O):
    """The per-rank piece of `-b N` with the oracle standing in for the device: a replicate's draw is
    keyed by (seed, its number), the EM starts from the main estimate (seekmer/infer.py:108-118)."""
    def share(table, x0, first, step, count, seed, device):
        offsets = table['class_offsets']
        class_map = np.vstack([np.repeat(np.arange(offsets.size - 1), np.diff(offsets)),
                               table['class_targets'].astype(np.int64)])
        counts = table['class_count']
        out = np.zeros((count, table['effective_lengths'].size))
        for j in range(count):
            rng = np.random.default_rng([seed, first + j * step])
            draw = rng.multinomial(int(counts.sum()), counts / counts.sum()).astype('f8')
            out[j], _ = O.quantify(table['effective_lengths'], class_map, draw, x0=x0)
        return out
    return share


def _worker(rank, world, port, tmp, queue):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update({'RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank),
                       'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
    from oracle import oracle as O
    from seekmer_amd import common, infer, parallel
    ranks = parallel.Ranks.from_env()
    try:
        assert (ranks.rank, ranks.world, ranks.local_rank) == (rank, world, rank) and ranks.shard == (rank, world)
        index, paths = _sample(tmp)
        result = OracleMapResult(O, index)
        mine = []
        for batch in common.NativeReadFeeder(paths, paired=True, batch_units=BATCH, threads=2 * rank,
                                             shard=ranks.shard):
            mine.append(batch.first_unit)
            result.map_batch(batch)
        assert mine == [BATCH * k for k in range(rank, (N_UNITS + BATCH - 1) // BATCH, world)]
        quantify = _quantify_ranks(O, index, ranks)
        summarized, tpm = infer.finish(result, ranks, quantify)
        everyone = ranks.gather_to_root(tpm)
        boots = infer.bootstrap_ranks(summarized, tpm, N_BOOT, ranks, seed=BOOT_SEED, share=_oracle_share(O))
        assert parallel.replicate_share(N_BOOT, rank, world) == (rank, world, len(range(rank, N_BOOT, world)))
        if rank == 0:
            assert all(np.array_equal(t, tpm) for t in everyone)          # every rank holds the same TPM
            assert len(boots) == N_BOOT
            queue.put({'summarized': {k: getattr(summarized, k) for k in
                                      ('aligned', 'unaligned', 'total', 'class_map', 'class_count',
                                       'fragment_length_frequencies', 'effective_lengths')},
                       'export': result.export(), 'tpm': tpm, 'steps': quantify.steps, 'boots': np.asarray(boots)})
        else:
            assert summarized is None and boots == []
        # the one-pass packed reader over this rank's share of the sample (what infer.run opens for
        # plain files on several ranks): the newline counts are added up through the process group
        import test_packed_reads as packed
        feeder = infer._feeder(paths, True, 2, ranks.shard, False, sum_over_ranks=ranks.sum_int64)
        assert isinstance(feeder, common.PackedReadFeeder) and feeder.shard == (rank, world)
        pieces = [p.copy() for p in feeder]
        begin, end, first, count = feeder.share
        assert (first, count) == (N_UNITS * rank // world, N_UNITS * (rank + 1) // world - N_UNITS * rank // world)
        streams, ends, _ = packed.assemble(pieces, 2)
        pairs = O.read_fastq_pairs(*paths)
        for s in range(2):
            assert sorted(streams[s]) == list(range(first, first + count))
            for u in range(first, first + count, 37):
                assert streams[s][u] == packed.expected_packing(pairs[2 * u + s])
    finally:
        ranks.close()


def test_two_ranks_equal_one(oracle, native_libs, tmp_path):
    import torch.multiprocessing as mp
    index, paths = _sample(str(tmp_path))
    ctx = mp.get_context('spawn')
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), queue)) for r in range(2)]
    for p in procs:
        p.start()
    got = queue.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    pairs = oracle.read_fastq_pairs(*paths)
    bases, offsets = oracle.pack_reads(pairs)
    fld = np.zeros(2000, dtype=np.int64)
    result = oracle.map_batch(index, bases, offsets, N_UNITS, True, fld)
    classes = oracle.Classes()
    classes.update(result)
    offs, targets, counts = classes.export()
    g_offs, g_targets, g_counts, g_first, g_fld = got['export']
    np.testing.assert_array_equal(g_fld, fld)
    np.testing.assert_array_equal(g_offs, offs)      # same classes, same (-j1) order
    np.testing.assert_array_equal(g_targets, targets)
    np.testing.assert_array_equal(g_counts, counts)
    assert (np.diff(g_first) > 0).all()
    class_map, class_count = classes.summarize()
    s = got['summarized']
    np.testing.assert_array_equal(s['class_map'], class_map)
    np.testing.assert_array_equal(s['class_count'], class_count)
    assert (s['aligned'], s['unaligned'], s['total']) == (int(class_count.sum()), classes.unaligned, N_UNITS)
    eff = oracle.effective_lengths(fld, index.lengths)
    np.testing.assert_array_equal(s['effective_lengths'], eff)
    tpm_ref, iters_ref = oracle.quantify(eff, class_map, class_count)
    assert got['steps'] == iters_ref
    mask = tpm_ref > 0
    np.testing.assert_array_equal(got['tpm'] > 0, mask)
    np.testing.assert_allclose(got['tpm'][mask], tpm_ref[mask], rtol=1e-9, atol=0)
    # -b N over two ranks == the one-rank loop, replicate by replicate
    table = {'class_offsets': offs, 'class_targets': targets, 'class_count': class_count, 'effective_lengths': eff}
    alone = _oracle_share(oracle)(table, got['tpm'], 0, 1, N_BOOT, BOOT_SEED, 0)
    np.testing.assert_array_equal(got['boots'], alone)
    assert len({row.tobytes() for row in got['boots']}) == N_BOOT          # (the replicates differ)


def test_replicate_shares_partition_the_bootstraps():
    from seekmer_amd import parallel
    for n in (0, 1, 7, 8, 100):
        for world in (1, 2, 3, 8):
            numbers = []
            for r in range(world):
                first, step, count = parallel.replicate_share(n, r, world)
                numbers += [first + j * step for j in range(count)]
            assert sorted(numbers) == list(range(n))


def _failing_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update({'RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank),
                       'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'SKM_DIST_TIMEOUT_S': '20'})
    from seekmer_amd import parallel
    ranks = parallel.Ranks.from_env()
    try:
        if rank == 1:
            raise RuntimeError('this rank cannot go on')
        ranks.barrier()                      # the peer never arrives: this must not wait for ever
    except BaseException as error:           # noqa: B902
        ranks.fail(error)
    ranks.close()


def test_a_failing_rank_ends_the_job():
    """parallel.Ranks.fail: the failing rank leaves with a non-zero status at once (a launcher stops the
    others); a peer that is not stopped by a launcher gives up after the group's timeout instead of
    waiting for ever -- and fails in turn."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    assert procs[1].exitcode == 1
    assert procs[0].exitcode not in (0, None)


def test_one_rank_needs_no_process_group(monkeypatch):
    from seekmer_amd import parallel
    for name in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(name, raising=False)
    ranks = parallel.Ranks.from_env()
    assert (ranks.rank, ranks.world, ranks.shard, ranks.dist) == (0, 1, None, None)
    assert ranks.gather_to_root('x') == ['x']
    assert parallel.make_comm(ranks, 0) is None
    ranks.barrier()
    ranks.close()
