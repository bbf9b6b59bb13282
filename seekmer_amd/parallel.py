"""Multi-GPU host logic (not in the reference, which is single-process).

Reads are independent units, so they shard across ranks with no collective
while mapping; every rank holds a full index replica.  Exchange steps:
  * once after mapping: the fragment-length histogram is all-reduced (the
    effective lengths need the global one) and, when the merged class table is
    wanted (bit-identical class counts / reference class order), the per-rank
    tables are all-gathered and merged by global first-seen unit index;
  * per EM step: one all-reduce(sum) of f64[T] -- done inside the HIP library
    with RCCL (skm_comm_create + skm_quant_set_comm); classes stay rank-local because the EM
    numerators are linear in the class counts.
`dist` is `torch.distributed` (backend nccl == RCCL on the GPUs, gloo in the
CPU tests); torch is used for rendezvous and these small host-side exchanges
only.
"""
import numpy


def shard_range(n_units, rank, world):
    """Contiguous shard [first, first + count) of `n_units` for `rank`."""
    base, extra = divmod(int(n_units), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def allreduce_fld(fld, dist, group=None):
    """MapResult.merge_fragment_lengths across ranks (seekmer/mapper.py:106-115)."""
    import torch
    t = torch.from_numpy(numpy.ascontiguousarray(fld, dtype=numpy.int64).copy())
    dist.all_reduce(t, group=group)
    return t.numpy()


def gather_tables(table, dist, group=None):
    """All-gather per-rank class tables.  `table` = dict(offsets, targets,
    counts, first_seen (GLOBAL unit indices), unaligned)."""
    world = dist.get_world_size(group)
    out = [None] * world
    dist.all_gather_object(out, table, group=group)
    return out


def merge_class_tables(tables):
    """Counter.update over several tables (seekmer/mapper.py:70): counts add,
    classes are ordered by the smallest global first-seen unit index, which is
    the insertion order a single `-j 1` run over all units would produce.
    Returns dict(offsets, targets, counts, first_seen, unaligned)."""
    merged = {}
    unaligned = 0
    for table in tables:
        unaligned += int(table['unaligned'])
        offsets = table['offsets']
        targets = numpy.asarray(table['targets']).tolist()
        counts = table['counts']
        first = table['first_seen']
        for k in range(len(counts)):
            key = tuple(targets[offsets[k]:offsets[k + 1]])
            entry = merged.get(key)
            if entry is None:
                merged[key] = [int(counts[k]), int(first[k])]
            else:
                entry[0] += int(counts[k])
                if first[k] < entry[1]:
                    entry[1] = int(first[k])
    order = sorted(merged.items(), key=lambda kv: kv[1][1])
    offsets = numpy.zeros(len(order) + 1, dtype=numpy.int64)
    numpy.cumsum([len(k) for k, _ in order], out=offsets[1:])
    return {
        'offsets': offsets,
        'targets': numpy.asarray([t for k, _ in order for t in k], dtype=numpy.int32),
        'counts': numpy.asarray([v[0] for _, v in order], dtype=numpy.int64),
        'first_seen': numpy.asarray([v[1] for _, v in order], dtype=numpy.int64),
        'unaligned': unaligned,
    }


def export_table(map_result, first_unit):
    """The table of a rank's MapResult with first-seen indices made global."""
    offsets, targets, counts, first_seen, _ = map_result.export()
    return {'offsets': offsets, 'targets': targets, 'counts': counts,
            'first_seen': first_seen + int(first_unit), 'unaligned': map_result.sizes()[2]}


def broadcast_comm_id(dist, rank, group=None):
    """RCCL unique id from rank 0 to everyone (128 bytes)."""
    import ctypes
    import torch
    from . import _native
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        raw = ctypes.create_string_buffer(128)
        _native.check(_native.hip().skm_comm_unique_id(raw))
        buf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
    dist.broadcast(buf, 0, group=group)
    return bytes(buf.numpy().tobytes())


class Ranks:
    """This process's place in a one-process-per-GPU job, as `python -m torch.distributed.run
    --nproc-per-node N` describes it in the environment (RANK, WORLD_SIZE, LOCAL_RANK,
    MASTER_ADDR/PORT).  With WORLD_SIZE > 1 a gloo process group is opened for the host-side
    exchanges (rendezvous, the RCCL id, class tables to rank 0); nothing here touches the GPU,
    so a launcher may still start the ranks after this has been read."""

    def __init__(self, rank=0, world=1, local_rank=0, dist=None, owns_group=False):
        self.rank, self.world, self.local_rank = int(rank), int(world), int(local_rank)
        self.dist = dist
        self._owns_group = owns_group

    @classmethod
    def from_env(cls, backend='gloo'):
        import os
        rank = int(os.environ.get('RANK', '0'))
        world = int(os.environ.get('WORLD_SIZE', '1'))
        local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
        if world <= 1:
            return cls()
        import torch.distributed as dist
        owns = False
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group(backend, rank=rank, world_size=world)
            owns = True
        return cls(rank, world, local_rank, dist, owns)

    @property
    def shard(self):
        return (self.rank, self.world) if self.world > 1 else None

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def gather_to_root(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on rank 0, None elsewhere."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out

    def close(self):
        if self.dist is not None and self._owns_group:
            self.dist.barrier()
            self.dist.destroy_process_group()
        self.dist = None


class stdout_to_stderr:
    """RCCL announces itself ("RCCL version : ...") on the C-level standard output when a
    communicator is made; a program whose standard output is its result (bench.py's one JSON line,
    a pipeline reading abundance tables) wants that on stderr.  File descriptor 1 is pointed at
    descriptor 2 for the duration and restored."""

    def __enter__(self):
        import os
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import os
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def create_comm(device, comm_id, rank, world):
    """skm_comm_create with RCCL's banner kept off the standard output."""
    import ctypes
    from . import _native
    comm = ctypes.c_void_p()
    with stdout_to_stderr():
        _native.check(_native.hip().skm_comm_create(device, comm_id, rank, world, ctypes.byref(comm)))
    return comm


def make_comm(ranks, device):
    """One RCCL communicator per process (skm_comm*), or None for a single rank."""
    if ranks.world <= 1:
        return None
    return create_comm(device, broadcast_comm_id(ranks.dist, ranks.rank), ranks.rank, ranks.world)


def destroy_comm(comm):
    from . import _native
    if comm:
        _native.check(_native.hip().skm_comm_destroy(comm))


def rank_table(map_result):
    """What rank 0 needs from a rank after mapping: its class table (first-seen values are global
    unit indices already: every batch was mapped with its `first_unit`), the unaligned count and
    the rank's fragment-length histogram."""
    offsets, targets, counts, first_seen, fld = map_result.export()
    return {'offsets': offsets, 'targets': targets, 'counts': counts, 'first_seen': first_seen,
            'unaligned': map_result.sizes()[2], 'fld': fld}


def merge_into(map_result, tables):
    """Counter.update + merge_fragment_lengths with the other ranks' tables, on the GPU
    (skm_mapper_merge): rank 0's table becomes the table of the whole sample, classes in the
    order a single -j1 run would have met them."""
    for table in tables:
        map_result.merge_table(table['offsets'], table['targets'], table['counts'], table['first_seen'],
                               table['unaligned'], table['fld'])


def shared_index(build, rank, world, barrier=None, cache=None):
    """One index for all ranks of a node: rank 0 builds it (or finds it in `cache`) and saves the
    container, the others wait at `barrier` and map the same file (KMerIndex.load memory-maps
    it, so the 2 GiB table exists once in the page cache instead of once per rank and is built
    once instead of `world` times)."""
    import os
    import tempfile
    from . import common
    if world == 1 and not cache:
        return build()
    path = cache or os.path.join(tempfile.gettempdir(), 'skm_shared_index_%s.npz'
                                 % os.environ.get('MASTER_PORT', str(os.getppid())))
    index = None
    if rank == 0:
        if cache and os.path.exists(cache):
            index = common.KMerIndex.load(cache)
        else:
            index = build()
            index.save(path)
    if barrier is not None:
        barrier()
    if index is None:
        index = common.KMerIndex.load(path)
    if barrier is not None:
        barrier()
    if rank == 0 and not cache and world > 1:
        try:
            os.unlink(path)          # (the other ranks hold their mappings)
        except OSError:
            pass
    return index
