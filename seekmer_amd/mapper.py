"""Read mapping -- the `seekmer.mapper` surface (reference: seekmer/mapper.py,
seekmer/_mapper.pyx:31-105) over the MI355X engine.

``MapResult`` is backed by one device-resident mapper handle: the
equivalence-class counter and the fragment-length histogram live in HBM and
are materialised as ``collections.Counter`` / ``numpy`` objects only when the
attributes are read.  ``ReadMapper(index, map_result)(reads_iterator)`` packs
each batch and hands it to ``skm_mapper_map_batch``; there is no CPU mapping
path.
"""
import collections
import ctypes
import multiprocessing.pool
import queue
import threading

import numpy

from . import _native
from .common import PackedReadFeeder, PackedReads, ReadBatch

__all__ = ('MAX_FRAGMENT_LENGTH', 'MapResult', 'ReadMapper', 'SummarizedResult',
           'map_reads', 'map_multiple_samples')

MAX_FRAGMENT_LENGTH = 2000          # seekmer/_mapper.pyx:18-20

EPS = numpy.finfo('f4').eps


class SummarizedResult:
    """seekmer/mapper.py:18-37"""
    __slots__ = ['aligned', 'unaligned', 'total', 'class_map', 'class_count',
                 'fragment_length_frequencies', 'effective_lengths',
                 'class_offsets', 'class_targets', '_map_result']

    def __init__(self, aligned, unaligned, total, class_map, class_count,
                 fragment_length_frequencies, effective_lengths,
                 class_offsets=None, class_targets=None, map_result=None):
        self.aligned = aligned
        self.unaligned = unaligned
        self.total = total
        self.class_map = class_map
        self.class_count = class_count
        self.fragment_length_frequencies = fragment_length_frequencies
        self.effective_lengths = effective_lengths
        self.class_offsets = class_offsets
        self.class_targets = class_targets
        self._map_result = map_result


class MapResult:
    """A mapping result collection with a lock (seekmer/mapper.py:40-145)."""

    def __init__(self, index, readmap=None, device=0, keep_spans=False):
        """keep_spans: also store every unit's MappedSpan (begin, end, anchor) for
        ReadMapper.last_batch -- parity tests and diagnostics; inference does not read them."""
        self.lock = threading.Lock()
        self.index = index
        self.readmap = readmap
        self.device = device
        self._handle = ctypes.c_void_p()
        _native.check(_native.hip().skm_mapper_create(index.device_handle(device),
                                                      ctypes.byref(self._handle)))
        self._extra_fld = numpy.zeros(MAX_FRAGMENT_LENGTH, dtype='i8')
        self.keep_spans = bool(keep_spans)
        if keep_spans:
            _native.check(_native.hip().skm_mapper_keep_spans(self._handle, 1))

    def __del__(self):
        handle = getattr(self, '_handle', None)
        if handle:
            try:
                _native.hip().skm_mapper_destroy(handle)
            except Exception:
                pass
            self._handle = None

    # -- raw views of the device table --------------------------------------
    def sizes(self):
        """(classes, class_map rows, unaligned, total units)"""
        out = (ctypes.c_int64 * 4)()
        _native.check(_native.hip().skm_mapper_summary(self._handle, out))
        return tuple(int(v) for v in out)

    def export(self):
        """(class_offsets, class_targets, class_counts, first_seen, fld) in
        first-seen class order."""
        n_classes, n_rows, _, _ = self.sizes()
        offsets = numpy.zeros(n_classes + 1, dtype=numpy.int64)
        targets = numpy.zeros(max(n_rows, 1), dtype=numpy.int32)
        counts = numpy.zeros(max(n_classes, 1), dtype=numpy.int64)
        first = numpy.zeros(max(n_classes, 1), dtype=numpy.int64)
        fld = numpy.zeros(MAX_FRAGMENT_LENGTH, dtype=numpy.int64)
        _native.check(_native.hip().skm_mapper_export(
            self._handle, _native.ptr(offsets, _native.c_i64p), _native.ptr(targets, _native.c_i32p),
            _native.ptr(counts, _native.c_i64p), _native.ptr(first, _native.c_i64p),
            _native.ptr(fld, _native.c_i64p)))
        return offsets, targets[:n_rows], counts[:n_classes], first[:n_classes], fld

    @property
    def fragment_length_counts(self):
        fld = numpy.zeros(MAX_FRAGMENT_LENGTH, dtype=numpy.int64)
        _native.check(_native.hip().skm_mapper_export(
            self._handle, None, None, None, None, _native.ptr(fld, _native.c_i64p)))
        return fld

    @property
    def counter(self):
        """collections.Counter keyed by the id tuple; () = unaligned
        (seekmer/mapper.py:54, 70)."""
        offsets, targets, counts, _, _ = self.export()
        counter = collections.Counter()
        ids = targets.tolist()
        for k in range(counts.size):
            counter[tuple(ids[offsets[k]:offsets[k + 1]])] = int(counts[k])
        unaligned = self.sizes()[2]
        if unaligned:
            counter[()] = unaligned
        return counter

    # -- the reference's methods ---------------------------------------------
    def update(self, read_names, iterable):
        """Add mapping results given as id tuples (seekmer/mapper.py:60-75)."""
        tuples = [tuple(t) for t in iterable]
        local = collections.Counter(tuples)
        unaligned = local.pop((), 0)
        keys = list(local.keys())
        offsets = numpy.zeros(len(keys) + 1, dtype=numpy.int64)
        numpy.cumsum([len(k) for k in keys], out=offsets[1:])
        targets = numpy.asarray([t for k in keys for t in k] or [0], dtype=numpy.int32)
        counts = numpy.asarray([local[k] for k in keys] or [0], dtype=numpy.int64)
        base = self.sizes()[3]
        first_index = {}
        for i, t in enumerate(tuples):
            first_index.setdefault(t, i)
        first = numpy.asarray([base + first_index[k] for k in keys] or [0], dtype=numpy.int64)
        _native.check(_native.hip().skm_mapper_merge(
            self._handle, len(keys), _native.ptr(offsets, _native.c_i64p),
            _native.ptr(targets, _native.c_i32p), _native.ptr(counts, _native.c_i64p),
            _native.ptr(first, _native.c_i64p), unaligned, None))
        self._write_readmap(read_names, tuples)

    def _write_readmap(self, read_names, tuples):
        if self.readmap is None:
            return
        for read_name, targets in zip(read_names, tuples):
            ids = self.index.transcripts[list(targets),]['transcript_id']
            print(read_name.decode(), *[id_.decode() for id_ in ids], sep='\t', file=self.readmap)

    def merge_table(self, offsets, targets, counts, first_seen, unaligned, fld):
        """Counter.update + merge_fragment_lengths with another table."""
        _native.check(_native.hip().skm_mapper_merge(
            self._handle, counts.size,
            _native.ptr(numpy.ascontiguousarray(offsets, dtype=numpy.int64), _native.c_i64p),
            _native.ptr(numpy.ascontiguousarray(targets if targets.size else [0], dtype=numpy.int32), _native.c_i32p),
            _native.ptr(numpy.ascontiguousarray(counts if counts.size else [0], dtype=numpy.int64), _native.c_i64p),
            _native.ptr(numpy.ascontiguousarray(first_seen if first_seen.size else [0], dtype=numpy.int64), _native.c_i64p),
            int(unaligned),
            _native.ptr(numpy.ascontiguousarray(fld, dtype=numpy.int64), _native.c_i64p)))

    def device_table(self):
        """The table where it lies in HBM (skm_device_table): what a GPU-to-GPU hand-over copies."""
        table = _native.DeviceTable()
        _native.check(_native.hip().skm_mapper_device_table(self._handle, ctypes.byref(table)))
        return table

    def merge_resident(self, other):
        """Counter.update + merge_fragment_lengths with a table that lies on the same GPU (another
        MapResult, or a DeviceTable whose arrays were copied there): nothing crosses the host."""
        table = other if isinstance(other, _native.DeviceTable) else other.device_table()
        _native.check(_native.hip().skm_mapper_merge_device(self._handle, ctypes.byref(table)))

    def summarize(self):
        """seekmer/mapper.py:77-104: classes in Counter insertion order,
        class_map = int64[2, M] (row 0 class id, row 1 transcript id in tuple
        order, duplicates kept), class_count f8[C]."""
        offsets, targets, counts, _, fld = self.export()
        n_classes = counts.size
        _, _, unaligned, total = self.sizes()
        if targets.size:
            class_ids = numpy.repeat(numpy.arange(n_classes, dtype=numpy.int64), numpy.diff(offsets))
            class_map = numpy.vstack([class_ids, targets.astype(numpy.int64)])
        else:
            class_map = numpy.asarray([]).T
        class_count = counts.astype('f8')
        aligned = class_count.sum()
        return SummarizedResult(
            aligned=int(aligned),
            unaligned=int(unaligned),
            total=int(aligned + unaligned),
            class_map=class_map,
            class_count=class_count,
            fragment_length_frequencies=fld,
            effective_lengths=self._effective_lengths(fld),
            class_offsets=offsets,
            class_targets=targets,
            map_result=self,
        )

    def merge_fragment_lengths(self, fragment_length_counts):
        """seekmer/mapper.py:106-115"""
        fld = numpy.ascontiguousarray(fragment_length_counts, dtype=numpy.int64)
        empty = numpy.zeros(1, dtype=numpy.int64)
        _native.check(_native.hip().skm_mapper_merge(
            self._handle, 0, _native.ptr(empty, _native.c_i64p), None, None, None, 0,
            _native.ptr(fld, _native.c_i64p)))

    @property
    def harmonic_mean_fragment_length(self):
        """seekmer/mapper.py:117-132"""
        fld = self.fragment_length_counts
        assert fld[0] == 0
        numerator = fld.sum()
        if numerator == 0:
            return 0
        denominator = (fld[1:].astype('f8') / numpy.arange(1, MAX_FRAGMENT_LENGTH)).sum()
        return numerator / denominator

    @property
    def transcript_lengths(self):
        """index.transcripts['length'] as a contiguous f8 array (extracted once)."""
        cached = getattr(self, '_lengths', None)
        if cached is None:
            cached = numpy.ascontiguousarray(self.index.transcripts['length'], dtype='f8')
            self._lengths = cached
        return cached

    def _effective_lengths(self, fld):
        length = self.transcript_lengths
        out = numpy.zeros(length.shape, dtype='f8')
        _native.check(_native.hip().skm_effective_lengths(
            self.device, _native.ptr(numpy.ascontiguousarray(fld, dtype=numpy.int64), _native.c_i64p),
            _native.ptr(length, _native.c_f64p), length.size, _native.ptr(out, _native.c_f64p)))
        return out

    @property
    def effective_lengths(self):
        """seekmer/mapper.py:134-141 (computed on the GPU)"""
        return self._effective_lengths(self.fragment_length_counts)

    def clear(self):
        """Clear the counter (seekmer/mapper.py:143-145)."""
        _native.check(_native.hip().skm_mapper_clear(self._handle))

    def sync(self):
        """Wait for every batch handed over with map_batch_async; raises the first failure."""
        _native.check(_native.hip().skm_mapper_sync(self._handle))

    def reset(self):
        """Fresh-MapResult state (counter, totals and histogram) on the same buffers."""
        _native.check(_native.hip().skm_mapper_reset(self._handle))

    def map_resident(self, d_bases, d_offsets, n_units, paired, max_read_len):
        """Map a batch that already lives in HBM (device pointers)."""
        _native.check(_native.hip().skm_mapper_map_batch_device(
            self._handle, d_bases, d_offsets, n_units, int(bool(paired)), max_read_len))

    def set_stats(self, enable):
        _native.check(_native.hip().skm_mapper_set_stats(self._handle, int(enable)))

    def access_stats(self):
        out = (ctypes.c_int64 * 48)()
        _native.check(_native.hip().skm_mapper_access_stats(self._handle, out))
        names = ('reads', 'read_bases', 'lookups', 'slots', 'contig_reads', 'targets_copied',
                 'targets_merged', 'seq_fetches', 'merges', 'tuple_ids')
        stats = {n: int(out[i]) for i, n in enumerate(names)}
        census = ('rounds', 'start_exec', 'start_lanes', 'lookup_exec', 'lookup_lanes', 'merge_exec',
                  'merge_lanes', 'left_exec', 'left_lanes', 'right_exec', 'right_lanes', 'emit_exec',
                  'emit_lanes', 'scan_exec', 'scan_lanes')
        stats['census'] = {n: int(out[16 + i]) for i, n in enumerate(census)}
        cycles = ('schedule', 'barrier', 'start', 'lookup', 'merge', 'left', 'right', 'emit', 'scan',
                  'emit_mate1', 'emit_intersect', 'emit_fld', 'emit_scan', 'emit_entries', 'emit_store')
        stats['wave_cycles'] = {n: int(out[32 + i]) for i, n in enumerate(cycles)}
        return stats

    def timing(self):
        out = (ctypes.c_double * 8)()
        _native.check(_native.hip().skm_mapper_timing(self._handle, out))
        return {'pack_ns': out[0], 'map_ns': out[1], 'class_ns': out[2],
                'batches': int(out[3]), 'units': int(out[4]), 'em_ns': out[5], 'em_iterations': int(out[6])}


class ReadMapper:
    """A read mapper (seekmer/_mapper.pyx:31-105)."""

    def __init__(self, index, map_result):
        self.index = index
        self.map_result = map_result

    def map_batch(self, batch):
        """Map one batch and wait for it."""
        hip = _native.hip()
        _native.check(hip.skm_mapper_map_batch(
            self.map_result._handle, batch.bases.ctypes.data,
            _native.ptr(batch.offsets, _native.c_i64p), batch.count, int(batch.paired)))

    def map_batch_async(self, batch):
        """Hand one batch over (returns when its arrays are free again; the kernels run behind
        the next batch's parsing and copy).  MapResult.sync() -- or any read of the result --
        waits for what is queued.  A batch that knows its place in the sample
        (``batch.first_unit``) keeps the -j1 class order whatever the submission order."""
        hip = _native.hip()
        first_unit = getattr(batch, 'first_unit', None)
        first_unit = -1 if first_unit is None else int(first_unit)
        uniform = getattr(batch, 'uniform_len', None)
        if uniform is not None and batch.count:       # equal-length reads: the offsets stay on the host
            _native.check(hip.skm_mapper_map_batch_uniform_async(
                self.map_result._handle, batch.bases.ctypes.data + int(batch.offsets[0]), int(uniform),
                batch.count, int(batch.paired), first_unit))
            return
        _native.check(hip.skm_mapper_map_batch_async(
            self.map_result._handle, batch.bases.ctypes.data,
            _native.ptr(batch.offsets, _native.c_i64p), batch.count, int(batch.paired), first_unit))

    def push_packed(self, piece):
        """Hand over reads that are already packed (common.PackedReads): copied to HBM now, mapped in
        the background as soon as every stream of the sample covers a run of units."""
        _native.check(_native.hip().skm_mapper_push_packed(
            self.map_result._handle, ctypes.byref(piece.raw), int(piece.paired)))

    def drain_packed(self, feeder):
        """A PackedReadFeeder straight into the mapper without a Python step per piece
        (skm_mapper_map_packed_source over skm_fastq_packed_next); returns the pieces pushed."""
        reader = feeder.open()
        try:
            pieces = ctypes.c_int64()
            # (the files' sizes say about how many units are coming: tables sized once, not grown)
            _native.check(_native.hip().skm_mapper_expect_units(self.map_result._handle, reader.estimate()))
            _native.check(_native.hip().skm_mapper_map_packed_source(
                self.map_result._handle, ctypes.cast(_native.host().skm_fastq_packed_next, ctypes.c_void_p),
                reader.handle, int(feeder.paired), ctypes.byref(pieces)))
        finally:
            feeder.stats = reader.stats()
            reader.close()
        return pieces.value

    def last_batch(self, n_units):
        """(begin, end, anchor_entry, anchor_offset, counts, signed entries); the spans need a
        MapResult created with keep_spans=True."""
        spans = [numpy.zeros(max(n_units, 1), dtype=numpy.int32) for _ in range(4)]
        hip = _native.hip()
        _native.check(hip.skm_mapper_last_batch(
            self.map_result._handle, *[_native.ptr(a, _native.c_i32p) for a in spans],
            None, None, 0, None))
        counts, entries = self.last_tuples(n_units)
        return tuple(a[:n_units] for a in spans) + (counts, entries)

    def last_tuples(self, n_units):
        """(counts, signed entries) of the last batch, units in order."""
        hip = _native.hip()
        counts = numpy.zeros(max(n_units, 1), dtype=numpy.int32)
        needed = ctypes.c_int64()
        _native.check(hip.skm_mapper_last_batch(
            self.map_result._handle, None, None, None, None, _native.ptr(counts, _native.c_i32p),
            None, 0, ctypes.byref(needed)))
        entries = numpy.zeros(max(needed.value, 1), dtype=numpy.int32)
        _native.check(hip.skm_mapper_last_batch(
            self.map_result._handle, None, None, None, None, None,
            _native.ptr(entries, _native.c_i32p), entries.size, ctypes.byref(needed)))
        return counts[:n_units], entries[:needed.value]

    def __call__(self, reads_iterator):
        """Run the mapping loop (seekmer/_mapper.pyx:59-105)."""
        if isinstance(reads_iterator, PackedReadFeeder) and self.map_result.readmap is None:
            self.drain_packed(reads_iterator)
            self.map_result.sync()
            return
        for item in reads_iterator:
            if isinstance(item, PackedReads):
                if self.map_result.readmap is not None:
                    raise ValueError('-m/--save-readmap needs the reads as text: use NativeReadFeeder')
                self.push_packed(item)
                continue
            if isinstance(item, ReadBatch):
                batch = item
            else:
                read_count, read_names, reads = item
                batch = ReadBatch.from_lists(read_count, read_names, reads)
            if self.map_result.readmap is None:
                self.map_batch_async(batch)
                continue
            # -m/--save-readmap: the per-unit tuples of THIS batch are needed, so
            # keep other threads off the handle until they are fetched
            with self.map_result.lock:
                self.map_batch(batch)
                counts, entries = self.last_tuples(batch.count)
                ids = numpy.where(entries < 0, ~entries, entries).tolist()
                bounds = numpy.concatenate([[0], numpy.cumsum(counts)]).tolist()
                tuples = [tuple(ids[bounds[i]:bounds[i + 1]]) for i in range(batch.count)]
                self.map_result._write_readmap(batch.names, tuples)
        self.map_result.sync()            # (raises here what a queued batch failed with)


def _drain_worker(mapper, reads_queue, errors):
    """Thread body of map_reads: run the mapper over the queue; on a failure (device error,
    tag collision, bad batch) remember the first exception and keep taking batches up to the
    sentinel so the feeding thread never blocks on a full queue."""
    batches = iter(reads_queue.get, None)
    try:
        mapper(batches)
    except BaseException as error:        # noqa: B902 -- re-raised by map_reads in the caller's thread
        errors.append(error)
        for __ in batches:
            pass


def map_reads(index, read_feeder, job_count=1, readmap=None, debug=False, device=0):
    """Map reads (seekmer/mapper.py:148-193).  Unlike the reference's CPU workers the device
    calls can fail; a worker's exception is re-raised here once every thread has stopped,
    instead of being lost with its thread."""
    map_result = MapResult(index, readmap, device=device)
    try:
        if debug or job_count <= 1 or isinstance(read_feeder, PackedReadFeeder):
            # (a packed feeder parses with its own threads and is drained natively: the GIL is not held)
            ReadMapper(index, map_result)(read_feeder)
        else:
            reads_queue = queue.Queue(job_count * 2)
            threads, errors = [], []
            for __ in range(job_count):
                thread = threading.Thread(target=_drain_worker,
                                          args=(ReadMapper(index, map_result), reads_queue, errors))
                threads.append(thread)
                thread.start()
            try:
                for batch in read_feeder:
                    if errors:
                        break
                    reads_queue.put(batch)
            finally:
                for __ in range(job_count):
                    reads_queue.put(None)
                for thread in threads:
                    thread.join()
                threads.clear()
            if errors:
                raise errors[0]
    finally:
        if readmap is not None:
            readmap.close()
    return map_result


def map_multiple_samples(index, read_feeders, job_count=1, debug=False, device=0):
    """Map reads for multiple samples (seekmer/mapper.py:196-234); a failed sample raises."""
    map_results = []
    if debug:
        for read_feeder in read_feeders:
            result = MapResult(index, device=device)
            map_results.append(result)
            ReadMapper(index, result)(read_feeder)
    else:
        pool = multiprocessing.pool.ThreadPool(job_count)
        pending = []
        for read_feeder in read_feeders:
            result = MapResult(index, device=device)
            map_results.append(result)
            pending.append(pool.apply_async(_map, args=(index, result, read_feeder)))
        pool.close()
        pool.join()
        for job in pending:
            job.get()                     # re-raises what the worker raised
    return map_results


def _map(index, map_result, read_feeder):
    ReadMapper(index, map_result)(read_feeder)
    return None
