"""Multi-GPU host logic (not in the reference, which is single-process).

Reads are independent units, so they shard across ranks with no collective
while mapping; every rank holds a full index replica.  Exchange steps:
  * once after mapping: the fragment-length histogram is all-reduced (the
    effective lengths need the global one; inside skm_quant_infer, RCCL);
  * per EM step: one all-reduce(sum) of f64[T] -- done inside the HIP library
    with RCCL (skm_comm_create + skm_quant_set_comm); classes stay rank-local because the EM
    numerators are linear in the class counts;
  * after the EM: the ranks' class tables go to rank 0 as raw arrays and are merged into its
    own on the GPU (skm_mapper_merge), classes ordered by global first-seen unit index;
  * `-b N`: rank 0 shares the merged table, every rank runs its share of the replicates with
    no collective, the results are gathered on rank 0.
`dist` is `torch.distributed` (backend nccl == RCCL on the GPUs, gloo in the
CPU tests); torch is used for rendezvous and these small host-side exchanges
only.
"""
import numpy


def replicate_share(n_boot, rank, world):
    """(first, step, count): the bootstrap replicates of `rank` -- numbers first, first + step, ...
    below n_boot (B / G replicates per GPU, no collective while they run)."""
    first, step = int(rank), int(world)
    return first, step, len(range(first, int(n_boot), step))


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
            import datetime
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            # how long a rank waits for its peers in a host-side exchange before giving up (a peer
            # that died without the launcher noticing); every rank takes part in every phase, so
            # the longest wait is the slowest rank's lag, not a whole phase
            timeout = datetime.timedelta(seconds=float(os.environ.get('SKM_DIST_TIMEOUT_S', '1800')))
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
            owns = True
        return cls(rank, world, local_rank, dist, owns)

    @property
    def shard(self):
        return (self.rank, self.world) if self.world > 1 else None

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def gather_to_root(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on rank 0, None elsewhere (small objects: pickled)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out

    def gather_arrays_to_root(self, arrays):
        """Every rank's dict of numpy arrays on rank 0 ([dict of rank 0, dict of rank 1, ...]; None
        elsewhere) without pickling them: shapes and dtypes travel as a small object, the data as
        raw bytes straight out of and into the arrays."""
        if self.dist is None:
            return [arrays]
        import torch
        arrays = {k: numpy.ascontiguousarray(v) for k, v in arrays.items()}
        described = self.gather_to_root({k: (v.shape, v.dtype.str) for k, v in arrays.items()})
        if self.rank != 0:
            for name in sorted(arrays):
                if arrays[name].size:
                    self.dist.send(torch.from_numpy(arrays[name].reshape(-1).view(numpy.uint8)), dst=0)
            return None
        out = [arrays]
        for source in range(1, self.world):
            got = {}
            for name in sorted(described[source]):
                shape, dtype = described[source][name]
                array = numpy.empty(shape, dtype=numpy.dtype(dtype))
                if array.size:
                    self.dist.recv(torch.from_numpy(array.reshape(-1).view(numpy.uint8)), src=source)
                got[name] = array
            out.append(got)
        return out

    def broadcast_arrays(self, arrays):
        """Rank 0's dict of numpy arrays on every rank (the argument is ignored elsewhere)."""
        if self.dist is None:
            return arrays
        import torch
        if self.rank == 0:
            arrays = {k: numpy.ascontiguousarray(v) for k, v in arrays.items()}
        box = [{k: (v.shape, v.dtype.str) for k, v in arrays.items()} if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        out = {}
        for name in sorted(box[0]):
            shape, dtype = box[0][name]
            array = arrays[name] if self.rank == 0 else numpy.empty(shape, dtype=numpy.dtype(dtype))
            if array.size:
                self.dist.broadcast(torch.from_numpy(array.reshape(-1).view(numpy.uint8)), src=0)
            out[name] = array
        return out

    def sum_int64(self, array):
        """The element-wise sum of every rank's int64 array, on every rank (an all-reduce on the host)."""
        array = numpy.ascontiguousarray(array, dtype=numpy.int64)
        if self.dist is None or array.size == 0:
            return array
        import torch
        total = torch.from_numpy(array.copy())
        self.dist.all_reduce(total)
        return total.numpy()

    def fail(self, error):
        """A rank that cannot go on must not leave its peers waiting in a barrier, a gather or an
        RCCL all-reduce: report, and end THIS process with a non-zero status at once (the launcher
        -- torch.distributed.run -- then stops the other ranks; nothing is restarted or re-executed).
        With one rank the error is simply raised."""
        if self.world <= 1:
            raise error
        import os
        import sys
        import traceback
        traceback.print_exception(type(error), error, error.__traceback__, file=sys.stderr)
        print('[seekmer_amd] rank %d of %d failed: ending the job' % (self.rank, self.world), file=sys.stderr, flush=True)
        os._exit(1)

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
    the rank's fragment-length histogram -- all numpy arrays (Ranks.gather_arrays_to_root)."""
    offsets, targets, counts, first_seen, fld = map_result.export()
    return {'offsets': offsets, 'targets': targets, 'counts': counts, 'first_seen': first_seen,
            'unaligned': numpy.asarray([map_result.sizes()[2]], dtype=numpy.int64), 'fld': fld}


def merge_into(map_result, tables):
    """Counter.update + merge_fragment_lengths with the other ranks' tables, on the GPU
    (skm_mapper_merge): rank 0's table becomes the table of the whole sample, classes in the
    order a single -j1 run would have met them."""
    for table in tables:
        map_result.merge_table(table['offsets'], table['targets'], table['counts'], table['first_seen'],
                               int(numpy.asarray(table['unaligned']).reshape(-1)[0]), table['fld'])


def hand_over_device(map_result, comm, ranks):
    """The ranks' tables to rank 0 from GPU to GPU (skm_mapper_exchange_tables): rank r sends its table
    as it lies in HBM, rank 0 receives the others' one after the other and merges each by key on its
    GPU.  Every rank calls this; the order of the calls pairs every send with its receive."""
    from . import _native
    hip = _native.hip()
    if ranks.rank == 0:
        for peer in range(1, ranks.world):
            _native.check(hip.skm_mapper_exchange_tables(None, -1, map_result._handle, peer, comm))
    else:
        _native.check(hip.skm_mapper_exchange_tables(map_result._handle, 0, None, -1, comm))


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
