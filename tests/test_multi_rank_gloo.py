"""The N > 1 path on CPU: world_size 2, gloo.  Each rank maps its shard (with
the oracle as the compute stand-in -- no GPU here), then the PRODUCT's host
logic (seekmer_amd.parallel) all-reduces the histogram, gathers and merges the
class tables; the result must equal one process mapping everything, and one
sharded EM step (local numerators + all-reduce) must equal the global step."""
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


def _worker(rank, world, port, queue):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from seekmer_amd import parallel, synth
        ids, pool, tx_offsets = synth.transcriptome(4, 30)
        index = O.build_index(synth.sequences_of(pool, tx_offsets))
        n_units = 6001
        first, count = parallel.shard_range(n_units, rank, world)
        bases, offsets = synth.reads(4, pool, tx_offsets, first, count, 75, True, n_threads=1)
        fld = np.zeros(2000, dtype=np.int64)
        result = O.map_batch(index, bases, offsets, count, True, fld)
        classes = O.Classes()
        classes.update(result)
        offs, targets, counts = classes.export()
        # first-seen unit of each class within this shard, made global
        tuples = result.tuples()
        seen = {}
        for u, t in enumerate(tuples):
            if t and t not in seen:
                seen[t] = u
        tl = targets.tolist()
        first_seen = np.asarray([seen[tuple(tl[offs[k]:offs[k + 1]])] + first
                                 for k in range(counts.size)], dtype=np.int64)
        table = {'offsets': offs, 'targets': targets, 'counts': counts, 'first_seen': first_seen,
                 'unaligned': classes.unaligned}
        global_fld = parallel.allreduce_fld(fld, dist)
        merged = parallel.merge_class_tables(parallel.gather_tables(table, dist))

        # one sharded EM step: local numerators, all-reduce, finalise
        import torch
        eff = O.effective_lengths(global_fld, index.lengths)
        x = 1.0 / eff
        x /= x.sum()
        cls = np.repeat(np.arange(counts.size), np.diff(offs))
        w = x[targets]
        inner = np.bincount(cls, weights=w, minlength=counts.size) / counts
        local = np.bincount(targets, weights=w / inner[cls], minlength=eff.size)
        n_local = torch.tensor([float(counts.sum())], dtype=torch.float64)
        t = torch.from_numpy(local)
        dist.all_reduce(t)
        dist.all_reduce(n_local)
        x_new = t.numpy() / eff / float(n_local[0])
        x_new[x_new != x_new] = 0
        if rank == 0:
            queue.put({'fld': global_fld, 'merged': merged, 'x_new': x_new})
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_equal_one(oracle, native_libs):
    import torch.multiprocessing as mp
    from seekmer_amd import parallel, synth
    ctx = mp.get_context('spawn')
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    got = queue.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    ids, pool, tx_offsets = synth.transcriptome(4, 30)
    index = oracle.build_index(synth.sequences_of(pool, tx_offsets))
    bases, offsets = synth.reads(4, pool, tx_offsets, 0, 6001, 75, True, n_threads=1)
    fld = np.zeros(2000, dtype=np.int64)
    result = oracle.map_batch(index, bases, offsets, 6001, True, fld)
    classes = oracle.Classes()
    classes.update(result)
    offs, targets, counts = classes.export()
    np.testing.assert_array_equal(got['fld'], fld)
    np.testing.assert_array_equal(got['merged']['offsets'], offs)      # same classes, same order
    np.testing.assert_array_equal(got['merged']['targets'], targets)
    np.testing.assert_array_equal(got['merged']['counts'], counts)
    assert got['merged']['unaligned'] == classes.unaligned
    class_map, class_count = classes.summarize()
    eff = oracle.effective_lengths(fld, index.lengths)
    x0 = 1.0 / eff
    x0 /= x0.sum()
    x_ref, _ = oracle.em(x0, eff, class_map, class_count, fixed_iters=1)
    np.testing.assert_allclose(got['x_new'], x_ref, rtol=1e-12, atol=0)


def test_shard_range_covers_everything():
    from seekmer_amd import parallel
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
