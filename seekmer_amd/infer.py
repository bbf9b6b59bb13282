"""Abundance inference -- the `seekmer.infer` surface (reference:
seekmer/infer.py:1-353) over the MI355X engine: `run`, `quantify`, `em`,
`output_results`, `add_subcommand_parser` keep their signatures; the EM loop
and the bootstrap resampling run as HIP kernels behind the C ABI.
"""
import ctypes
import datetime
import json
import logging
import pathlib
import shlex
import sys

import numpy

from . import _native
from . import common
from . import mapper

__all__ = ['run', 'quantify', 'quantify_resident', 'em', 'output_results', 'bootstrap_quantify', 'bootstrap_ranks']

_LOG = logging.getLogger(__name__)

REL_TOL = 0.01          # seekmer/infer.py:160
X_FLOOR = 1e-8          # seekmer/infer.py:160


def run(index_path, output_path, fastq_paths, job_count, save_readmap,
        single_ended, bootstrap, debug, device=0, seed=None, parse_threads=None, **__):
    """The entrypoint of the inference module (seekmer/infer.py:27-85).

    Started as one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m
    seekmer_amd infer ...`; experimental: no multi-GPU node has run it yet) the sample is shared out
    batch by batch: every rank maps its batches on its own GPU against its own replica of the
    index, the EM runs over the rank-local class tables with one RCCL all-reduce per step, rank 0
    merges the ranks' tables on its GPU, the `-b N` replicates are shared out over the ranks again
    (the merged table on every rank, no collective while they run), and rank 0 writes the outputs
    of the whole sample.  A rank that fails ends the job (parallel.Ranks.fail)."""
    from . import parallel
    start_time = datetime.datetime.utcnow()
    ranks = parallel.Ranks.from_env()
    try:
        _run(ranks, start_time, index_path, output_path, fastq_paths, job_count, save_readmap, single_ended,
             bootstrap, debug, device, seed, parse_threads)
    except BaseException as error:          # noqa: B902 -- one rank: re-raised as it is
        ranks.fail(error)
    ranks.close()


def _run(ranks, start_time, index_path, output_path, fastq_paths, job_count, save_readmap, single_ended,
         bootstrap, debug, device, seed, parse_threads):
    from . import parallel
    if ranks.world > 1:
        device = ranks.local_rank
        if save_readmap:
            raise ValueError('-m/--save-readmap needs the reads of the whole sample in one process: run it on one GPU')
    if ranks.rank == 0:
        try:
            output_path.mkdir(parents=True)
        except FileExistsError:
            _LOG.warning('The output folder exists. Overriding...')
    ranks.barrier()
    readmap = (output_path / 'readmap.txt').open('wt') if save_readmap else None
    _LOG.info('Inferring transcript abundance')
    # (the readers' threads page-lock against THIS GPU; their arena is page-locked by a helper thread
    # from here on, under the index load and upload)
    _native.check(_native.hip().skm_pinned_set_device(device))
    read_feeder = _feeder(fastq_paths, not single_ended, parse_threads, ranks.shard, save_readmap,
                          sum_over_ranks=ranks.sum_int64)
    one_pass = isinstance(read_feeder, common.PackedReadFeeder)
    if one_pass and ranks.world > 1:
        read_feeder.locate_share()          # (a collective: every rank counts its share of the newlines, here)
    # (the one-pass reader maps the text: its page tables are set up by helper threads while the index loads)
    ahead = common.Prefault(fastq_paths if one_pass and ranks.world == 1 else [], threads=4)
    try:
        index = common.KMerIndex.load(index_path)
        index.device_handle(device)
        _LOG.info('Mapping all reads')
        map_result = mapper.map_reads(index, read_feeder, job_count=job_count,
                                      readmap=readmap, debug=debug, device=device)
    finally:
        ahead.finish()
    _LOG.info('Mapped all reads')
    if bootstrap > 0:
        _check_resample_limit(map_result, ranks)
    comm = parallel.make_comm(ranks, device)
    try:
        summarized_results, main_result = finish(map_result, ranks,
                                                 lambda result: quantify_resident(result, comm=comm), comm=comm)
    finally:
        parallel.destroy_comm(comm)
    if ranks.rank == 0:
        _LOG.info('Estimated fragment length: %.2f', map_result.harmonic_mean_fragment_length)
        _LOG.info('Aligned %d reads (%.2f%%)', summarized_results.aligned,
                  100.0 * summarized_results.aligned / summarized_results.total)
        _LOG.info('Quantified transcripts')
    bootstrapped_results = bootstrap_ranks(summarized_results, main_result, bootstrap, ranks, seed=seed, device=device)
    if ranks.rank == 0:
        output_results(output_path, index, start_time, summarized_results,
                       main_result, bootstrapped_results)
        _LOG.info('Wrote results to %s', output_path)


RESAMPLE_LIMIT = 2 ** 32 - 1        # units one multinomial draw can resample (skm_quant_bootstrap*)


def _check_resample_limit(map_result, ranks):
    """`-b N` draws n = class_count.sum() units per replicate (seekmer/infer.py:108-111); the device
    draw counts them in 32 bits.  Said now, before the EM, not after it."""
    _, _, unaligned, total = map_result.sizes()
    aligned = total - unaligned
    if ranks.world > 1:
        aligned = sum(ranks.gather_to_root(aligned) or [0])
    if ranks.rank == 0 and aligned > RESAMPLE_LIMIT:
        raise ValueError('-b/--bootstrap resamples at most %d aligned units per replicate; this sample has %d'
                         % (RESAMPLE_LIMIT, aligned))


def _feeder(fastq_paths, paired, parse_threads, shard, keep_reference_batches, sum_over_ranks=None):
    """The native readers as run() uses them.

    Plain files: the one-pass reader (common.PackedReadFeeder) -- the text is parsed straight to the
    mapper's 2-bit read codes by `parse_threads` workers (None = up to 16, the cores this process may
    use less one) and drained into the mapper natively; a third of the bytes of the ASCII batches
    cross PCIe.  Several ranks: each reads a contiguous run of the sample's units, found from
    newline counts that the ranks add up (`sum_over_ranks`; PackedReadFeeder.locate_share).
    Compressed inputs (every rank then takes every world-th batch of the sample off the two-pass
    reader's line index) and `-m` (readmap.txt is written batch by batch in the reference's batches
    of 65 536, seekmer/common.py:17) go through common.NativeReadFeeder: plain files parsed by up to 8 threads, into page-locked slabs when
    there is enough text (> 4 GiB) to pay for pinning them, in batches of 2^18 units (2^20 above
    16 GiB of text); parse_threads 0 = its sequential engine."""
    import os
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    total = 0
    for path in fastq_paths:
        try:
            total += os.path.getsize(str(path))
        except OSError:
            pass
    one_rank = shard is None or shard[1] <= 1
    if (one_rank or sum_over_ranks is not None) and not keep_reference_batches and parse_threads != 0 \
            and common.PackedReadFeeder.eligible(fastq_paths):
        threads = parse_threads if parse_threads else max(1, min(16, cores - 1))
        world = 1 if one_rank else shard[1]
        return common.PackedReadFeeder(fastq_paths, paired, threads=threads, pinned=total > (1 << 30) * world,
                                       shard=None if one_rank else shard, sum_over_ranks=sum_over_ranks)
    if parse_threads is None:
        parse_threads = 0 if keep_reference_batches else min(8, cores)
    if parse_threads <= 0:
        return common.NativeReadFeeder(fastq_paths, paired=paired, shard=shard)
    return common.NativeReadFeeder(fastq_paths, paired=paired, batch_units=1 << (20 if total > (16 << 30) else 18),
                                   threads=parse_threads, pinned=total > (4 << 30), shard=shard)


def finish(map_result, ranks, quantify_ranks, comm=None):
    """From the ranks' tables to the sample's results.  One rank: summarize + quantify as the
    reference does (seekmer/infer.py:66-78).  Several: `quantify_ranks(map_result)` runs the EM
    over the rank-local tables (collectives inside; every rank gets the TPM of the whole sample),
    THEN the other ranks' tables go to rank 0 and are merged into its own, whose summary is the
    whole sample's: (SummarizedResult, TPM) on rank 0, (None, TPM) elsewhere.  The tables travel as
    numpy arrays through the process group (host copy, rank 0 merges on its GPU) -- or, with
    SKM_HANDOVER=rccl and a communicator, from GPU to GPU as they lie in HBM
    (parallel.hand_over_device: ncclSend / ncclRecv + merge by key; experimental, no multi-GPU
    node has run it yet)."""
    from . import parallel
    if ranks.world == 1:
        summarized = map_result.summarize()
        _LOG.info('Quantifying transcripts')
        return summarized, quantify(summarized)
    _LOG.info('Quantifying transcripts')
    tpm = quantify_ranks(map_result)
    import os
    if comm and os.environ.get('SKM_HANDOVER') == 'rccl':
        parallel.hand_over_device(map_result, comm, ranks)
        return (map_result.summarize() if ranks.rank == 0 else None), tpm
    tables = ranks.gather_arrays_to_root(parallel.rank_table(map_result) if ranks.rank else {})
    if ranks.rank != 0:
        return None, tpm
    parallel.merge_into(map_result, tables[1:])
    return map_result.summarize(), tpm


# ------------------------------------------------------------- device tables
class _QuantHandle:
    """skm_quant* for one class table."""

    def __init__(self, handle, n_tx, n_classes):
        self.handle = handle
        self.n_tx = n_tx
        self.n_classes = n_classes

    @classmethod
    def from_csr(cls, n_tx, offsets, targets, counts, device=0):
        offsets = numpy.ascontiguousarray(offsets, dtype=numpy.int64)
        targets = numpy.ascontiguousarray(targets if len(targets) else [0], dtype=numpy.int32)
        counts = numpy.ascontiguousarray(counts if len(counts) else [0.0], dtype='f8')
        out = ctypes.c_void_p()
        _native.check(_native.hip().skm_quant_create(
            device, n_tx, offsets.size - 1, _native.ptr(offsets, _native.c_i64p),
            _native.ptr(targets, _native.c_i32p), _native.ptr(counts, _native.c_f64p),
            ctypes.byref(out)))
        return cls(out, n_tx, offsets.size - 1)

    @classmethod
    def from_map_result(cls, map_result, n_tx):
        out = ctypes.c_void_p()
        _native.check(_native.hip().skm_quant_create_from_mapper(
            map_result._handle, n_tx, ctypes.byref(out)))
        return cls(out, n_tx, map_result.sizes()[0])

    def em(self, x, l, fixed_iters=0, max_iters=0):
        x = numpy.array(x, dtype='f8', copy=True, order='C')
        l = numpy.ascontiguousarray(l, dtype='f8')
        iters = ctypes.c_int64()
        _native.check(_native.hip().skm_quant_em(
            self.handle, _native.ptr(x, _native.c_f64p), _native.ptr(l, _native.c_f64p),
            REL_TOL, X_FLOOR, max_iters, fixed_iters, ctypes.byref(iters)))
        return x, iters.value

    def set_counts(self, counts):
        counts = numpy.ascontiguousarray(counts, dtype='f8')
        _native.check(_native.hip().skm_quant_set_counts(self.handle,
                                                         _native.ptr(counts, _native.c_f64p)))

    def bootstrap(self, n_boot, seed, x0, l, want_counts=False, tpm=False):
        """(results [n_boot, n_tx], resampled counts or None, EM steps per replicate); tpm: the
        results already scaled as quantify() scales them (seekmer/infer.py:127-129)."""
        x0 = numpy.ascontiguousarray(x0, dtype='f8')
        l = numpy.ascontiguousarray(l, dtype='f8')
        out = numpy.zeros((n_boot, self.n_tx), dtype='f8')
        if tpm and not want_counts:
            iters = numpy.zeros(max(n_boot, 1), dtype=numpy.int64)
            _native.check(_native.hip().skm_quant_bootstrap_tpm(
                self.handle, n_boot, seed, _native.ptr(x0, _native.c_f64p),
                _native.ptr(l, _native.c_f64p), REL_TOL, X_FLOOR, 0,
                _native.ptr(out, _native.c_f64p), _native.ptr(iters, _native.c_i64p)))
            return out, None, iters[:n_boot]
        counts = numpy.zeros((n_boot, max(self.n_classes, 1)), dtype=numpy.int64) if want_counts else None
        iters = numpy.zeros(max(n_boot, 1), dtype=numpy.int64)
        _native.check(_native.hip().skm_quant_bootstrap(
            self.handle, n_boot, seed, _native.ptr(x0, _native.c_f64p),
            _native.ptr(l, _native.c_f64p), REL_TOL, X_FLOOR, 0,
            _native.ptr(out, _native.c_f64p),
            _native.ptr(counts, _native.c_i64p) if want_counts else None,
            _native.ptr(iters, _native.c_i64p)))
        return out, counts, iters[:n_boot]

    def timing(self):
        out = (ctypes.c_double * 4)()
        _native.check(_native.hip().skm_quant_timing(self.handle, out))
        return {'em_ns': out[0], 'iterations': int(out[1]), 'launches': int(out[2])}

    def close(self):
        if self.handle:
            _native.hip().skm_quant_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _csr_from_class_map(class_map, n_classes):
    """class_map int64[2, M] (seekmer/mapper.py:93) -> CSR keeping the pair
    order inside every class."""
    class_ids = numpy.asarray(class_map[0], dtype=numpy.int64)
    targets = numpy.asarray(class_map[1], dtype=numpy.int64)
    if class_ids.size > 1 and (numpy.diff(class_ids) < 0).any():
        order = numpy.argsort(class_ids, kind='stable')
        class_ids, targets = class_ids[order], targets[order]
    offsets = numpy.zeros(n_classes + 1, dtype=numpy.int64)
    numpy.cumsum(numpy.bincount(class_ids, minlength=n_classes), out=offsets[1:])
    return offsets, targets.astype(numpy.int32)


def _tpm(x):
    """seekmer/infer.py:127-129"""
    x /= x.sum() / 1000000
    x[x < 0.001] = 0
    x /= x.sum() / 1000000
    return x


def _quant_for(results, device=0):
    cached = getattr(results, '_quant', None) if hasattr(results, '__dict__') else None
    if cached is not None:
        return cached, False
    n_tx = results.effective_lengths.size
    offsets = getattr(results, 'class_offsets', None)
    targets = getattr(results, 'class_targets', None)
    if offsets is None or targets is None:
        offsets, targets = _csr_from_class_map(results.class_map, results.class_count.size)
    return _QuantHandle.from_csr(n_tx, offsets, targets, results.class_count, device), True


def quantify(results, x0=None, bootstrap=False, seed=None, fixed_iters=0, return_iters=False, device=0):
    """Estimate the transcript abundance (seekmer/infer.py:88-130) on GPU `device`."""
    transcript_length = results.effective_lengths.astype('f8')
    if results.class_map.size == 0:
        zeros = numpy.zeros(results.effective_lengths.size).astype('f8')
        return (zeros, 0) if return_iters else zeros
    if x0 is None:
        x = numpy.ones(transcript_length.size, dtype='f8') / transcript_length
    else:
        x = x0.copy()
    x /= x.sum()
    quant, owned = _quant_for(results, device)
    try:
        if bootstrap:
            if seed is None:
                seed = int.from_bytes(__import__('os').urandom(8), 'little')
            out, _, iters = quant.bootstrap(1, seed, x, transcript_length)
            x, iters = out[0], int(iters[0])
        else:
            x, iters = quant.em(x, transcript_length, fixed_iters=fixed_iters)
    finally:
        if owned:
            quant.close()
    x = _tpm(x)
    return (x, iters) if return_iters else x


def quantify_resident(map_result, comm=None, return_iters=False, return_effective_lengths=False):
    """quantify(map_result.summarize()) without the class table leaving the GPU
    (seekmer/infer.py:88-130 + seekmer/mapper.py:134-141) -- one native call:
    fragment-length histogram (all-reduced over `comm`, an skm_comm handle, when
    given) -> effective lengths -> start vector -> EM -> TPM."""
    length = map_result.transcript_lengths
    tpm = numpy.empty(length.size, dtype='f8')
    eff = numpy.empty(length.size, dtype='f8') if return_effective_lengths else None
    iters = ctypes.c_int64()
    _native.check(_native.hip().skm_quant_infer(
        map_result._handle, comm, _native.ptr(length, _native.c_f64p), length.size, REL_TOL, X_FLOOR, 0,
        _native.ptr(tpm, _native.c_f64p), _native.ptr(eff, _native.c_f64p) if eff is not None else None,
        ctypes.byref(iters)))
    out = (tpm,)
    if return_iters:
        out += (iters.value,)
    if return_effective_lengths:
        out += (eff,)
    return out if len(out) > 1 else tpm


def bootstrap_quantify(results, x0, n_boot, seed=None, device=0):
    """The `-b N` loop of run() (seekmer/infer.py:79-82) as one device call on GPU `device`."""
    if n_boot <= 0:
        return []
    transcript_length = results.effective_lengths.astype('f8')
    if results.class_map.size == 0:
        return [numpy.zeros(transcript_length.size, dtype='f8') for _ in range(n_boot)]
    if seed is None:
        seed = int.from_bytes(__import__('os').urandom(8), 'little')
    x = x0.copy()
    x /= x.sum()
    quant, owned = _quant_for(results, device)
    try:
        out, _, _ = quant.bootstrap(n_boot, seed, x, transcript_length, tpm=True)
    finally:
        if owned:
            quant.close()
    return list(out)


def _bootstrap_share(table, x0, first, step, count, seed, device=0):
    """Replicates first, first + step, ... (`count` of them) of the `-b N` loop on this process's GPU:
    TPM vectors [count, T] (skm_quant_bootstrap_share_tpm)."""
    n_tx = table['effective_lengths'].size
    out = numpy.zeros((count, n_tx), dtype='f8')
    if count == 0:
        return out
    quant = _QuantHandle.from_csr(n_tx, table['class_offsets'], table['class_targets'], table['class_count'], device)
    try:
        x = numpy.array(x0, dtype='f8')
        x /= x.sum()
        length = numpy.ascontiguousarray(table['effective_lengths'], dtype='f8')
        iters = numpy.zeros(count, dtype=numpy.int64)
        _native.check(_native.hip().skm_quant_bootstrap_share_tpm(
            quant.handle, count, first, step, seed, _native.ptr(x, _native.c_f64p), _native.ptr(length, _native.c_f64p),
            REL_TOL, X_FLOOR, 0, _native.ptr(out, _native.c_f64p), _native.ptr(iters, _native.c_i64p)))
    finally:
        quant.close()
    return out


def bootstrap_ranks(results, x0, n_boot, ranks, seed=None, device=0, share=_bootstrap_share):
    """The `-b N` loop of run() (seekmer/infer.py:79-82) over the ranks: rank 0
    (the only one that holds `results`, the summary of the merged table, and `x0`, the main estimate)
    shares the merged table; rank r runs the replicates r, r + G, ... on its GPU with no collective;
    the TPM vectors are gathered on rank 0 and returned there in replicate order ([] elsewhere).
    A replicate's draw depends on (seed, its number) alone, so the list does not depend on G.
    `share(table, x0, first, step, count, seed, device)` is the per-rank piece (tests stand the
    oracle in for it)."""
    from . import parallel
    if n_boot <= 0:
        return []
    if ranks.world == 1:
        return bootstrap_quantify(results, x0, n_boot, seed=seed, device=device)
    table = None
    if ranks.rank == 0:
        if seed is None:
            seed = int.from_bytes(__import__('os').urandom(8), 'little')
        table = {'class_offsets': results.class_offsets, 'class_targets': results.class_targets,
                 'class_count': results.class_count, 'effective_lengths': results.effective_lengths.astype('f8'),
                 'x0': numpy.asarray(x0, dtype='f8'), 'seed': numpy.asarray([seed], dtype=numpy.uint64)}
    table = ranks.broadcast_arrays(table)
    seed = int(table['seed'][0])
    first, step, count = parallel.replicate_share(n_boot, ranks.rank, ranks.world)
    n_tx = table['effective_lengths'].size
    if table['class_count'].size == 0:                      # (quantify: an empty class_map gives zeros)
        mine = numpy.zeros((count, n_tx), dtype='f8')
    else:
        mine = share(table, table['x0'], first, step, count, seed, device)
    parts = ranks.gather_arrays_to_root({'tpm': mine})
    if ranks.rank != 0:
        return []
    out = [None] * n_boot
    for r, part in enumerate(parts):
        for j, number in enumerate(range(r, n_boot, ranks.world)):
            out[number] = part['tpm'][j]
    return out


def em(x, l, class_map, class_count, fixed_iters=0, return_iters=False, device=0):
    """Expectation-maximization (seekmer/infer.py:133-168)."""
    class_count = numpy.asarray(class_count, dtype='f8')
    offsets, targets = _csr_from_class_map(class_map, class_count.size)
    quant = _QuantHandle.from_csr(numpy.asarray(l).size, offsets, targets, class_count, device)
    try:
        x, iters = quant.em(x, l, fixed_iters=fixed_iters)
    finally:
        quant.close()
    return (x, iters) if return_iters else x


# ------------------------------------------------------------------- outputs
def output_results(output_path, index, start_time, results, main_abundance,
                   bootstrapped_abundance):
    """Output the quantification results (seekmer/infer.py:171-197)."""
    run_info = _generate_run_info(bootstrapped_abundance, index, results, start_time)
    with (output_path / 'run_info.json').open('w') as f:
        json.dump(run_info, f)
    est_counts = _infer_est_counts(index, results, main_abundance)
    _output_abundance_table(output_path, index, results, est_counts, main_abundance)
    _output_arrays(output_path, index, results, run_info, est_counts, bootstrapped_abundance)


def _generate_run_info(bootstrapped_abundance, index, results, start_time):
    """seekmer/infer.py:200-216"""
    class_target_count = numpy.bincount(results.class_map[0].astype(numpy.int64),
                                        minlength=results.class_count.size) \
        if results.class_map.size else numpy.zeros(results.class_count.size, dtype=numpy.int64)
    unique_count = results.class_count[class_target_count == 1].sum()
    return {
        'n_targets': len(index.transcripts),
        'n_bootstraps': len(bootstrapped_abundance),
        'n_processed': results.total,
        'n_pseudoaligned': results.aligned,
        'n_unique': int(unique_count),
        'p_pseudoaligned': results.aligned / results.total,
        'p_unique': unique_count / results.total,
        'kallisto_version': '0.44.0',
        'index_version': 9000,
        'start_time': start_time.isoformat(sep=' '),
        'call': ' '.join([shlex.quote(arg) for arg in sys.argv]),
    }


def _output_abundance_table(output_path, index, results, est_counts, main_abundance):
    """abundance.tsv: target_id, length, eff_length (f4), est_count, tpm; tab
    separated, floats as %g (seekmer/infer.py:219-230)."""
    ids = index.transcripts['transcript_id']
    length = index.transcripts['length']
    eff = results.effective_lengths.astype('f4')
    with (output_path / 'abundance.tsv').open('w') as f:
        f.write('target_id\tlength\teff_length\test_count\ttpm\n')
        for i in range(len(ids)):
            f.write('%s\t%g\t%g\t%g\t%g\n' % (ids[i].decode(), float(length[i]), float(eff[i]),
                                            float(est_counts[i]), float(main_abundance[i])))


def _infer_est_counts(index, results, main_abundance):
    """Estimated read counts from the raw length (seekmer/infer.py:233-252)."""
    est_counts = main_abundance * index.transcripts['length']
    est_counts *= results.aligned / est_counts.sum()
    return est_counts


def _output_arrays(output_path, index, results, run_info, est_counts, bootstrapped_abundance):
    """The datasets of abundance.h5 (seekmer/infer.py:255-325) as abundance.npz:
    PyTables/h5py are not installed here, so the kallisto-compatible HDF5
    container itself is out of reach; dataset names
    are kept ('aux/ids', 'est_counts', 'bootstrap/bs0', ...)."""
    arrays = {
        'aux/call': numpy.frombuffer(run_info['call'].encode() or b' ', dtype='S1'),
        'aux/index_version': numpy.asarray([run_info['index_version']]),
        'aux/start_time': numpy.frombuffer(run_info['start_time'].encode(), dtype='S1'),
        'aux/num_bootstrap': numpy.asarray([run_info['n_bootstraps']]),
        'aux/num_processed': numpy.asarray([run_info['n_processed']]),
        'aux/kallisto_version': numpy.frombuffer(run_info['kallisto_version'].encode(), dtype='S1'),
        'aux/ids': numpy.asarray(index.transcripts['transcript_id']),
        'aux/lengths': numpy.asarray(index.transcripts['length']),
        'aux/fld': results.fragment_length_frequencies.astype('i4'),
        'aux/eff_lengths': results.effective_lengths.astype('f8'),
        'aux/bias_observed': numpy.ones(4096, dtype='i4'),
        'aux/bias_normalized': numpy.ones(4096, dtype='f8'),
        'est_counts': est_counts.astype('f8'),
    }
    for i, bootstrap in enumerate(bootstrapped_abundance):
        arrays['bootstrap/bs{}'.format(i)] = bootstrap
    with (output_path / 'abundance.npz').open('wb') as f:
        numpy.savez(f, **arrays)


def add_subcommand_parser(subparsers):
    """Add an infer command to the subparsers (seekmer/infer.py:328-353)."""
    parser = subparsers.add_parser('infer', help='infer transcript abundance')
    parser.add_argument('index_path', type=pathlib.Path, metavar='index',
                        help='specify a Seekmer index file')
    parser.add_argument('output_path', type=pathlib.Path, metavar='output',
                        help='specify a output folder')
    parser.add_argument('fastq_paths', type=pathlib.Path, metavar='fastq',
                        nargs='+', help='specify a FASTQ read file')
    parser.add_argument('-j', '--jobs', type=int, dest='job_count', metavar='N', default=1,
                        help='specify the maximum parallel job number')
    parser.add_argument('-m', '--save-readmap', action='store_true', dest='save_readmap',
                        help='output an readmap file')
    parser.add_argument('-s', '--single-ended', action='store_true', dest='single_ended',
                        help='specify whether the reads are single-ended')
    parser.add_argument('-b', '--bootstrap', type=int, dest='bootstrap', default=0,
                        help='specify the number of bootstrapped estimation')
    parser.add_argument('--device', type=int, default=0, help='GPU ordinal (default 0)')
    parser.add_argument('--seed', type=int, default=None,
                        help='seed of the bootstrap resampling (default: random)')
    parser.add_argument('--parse-threads', type=int, dest='parse_threads', default=None, metavar='N',
                        help='parse plain FASTQ files with N threads (default: up to 8; 0: one thread, '
                             'the batches of the reference)')
