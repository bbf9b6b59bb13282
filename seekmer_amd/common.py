"""Index container and FASTQ feeders -- the `seekmer.common` surface
(reference: seekmer/common.py:1-197, seekmer/_common.pyx:19-313) over the
MI355X engine.

``KMerIndex`` keeps the reference's six public numpy attributes with the
reference's dtypes and lazily uploads the four mapping
arrays to HBM through the C ABI.  ``load``/``save`` use a flat ``.npz``
container: the reference's HDF5 layout needs PyTables, which this image lacks
(abundance.h5 / index HDF5 parity: unpinned).
"""
import bz2
import contextlib
import ctypes
import gzip
import io
import lzma
import pathlib
import subprocess
import threading

import numpy

from . import _native

__all__ = ('INVALID_INDEX', 'BUFFER_SIZE', 'decompress_and_open', 'read_fasta',
           'iterate_by_group', 'KMerIndex', 'feed_single_ended_reads',
           'feed_pair_ended_reads', 'NativeReadFeeder', 'ReadBatch', 'PackedReads', 'PackedReadFeeder')

INVALID_INDEX = 0x7FFFFFFF          # seekmer/_common.pxd:10
BUFFER_SIZE = 65536                 # seekmer/common.py:16

KMER_DTYPE = numpy.dtype([('kmer', '<u8'), ('entry', '<i4'), ('offset', '<i4')])
CONTIG_DTYPE = numpy.dtype([('offset', '<i8'), ('length', '<i8'), ('first_kmer', '<u8'),
                            ('last_kmer', '<u8'), ('target_offset', '<i8'),
                            ('target_count', '<i8')])
TARGET_DTYPE = numpy.dtype([('entry', '<i4'), ('offset', '<i4')])

_INDEX_VERSION = '2019.0.0'         # seekmer/_common.pyx:278


class KMerIndex:
    """The core index (seekmer/_common.pyx:19-48)."""

    def __init__(self, kmers, contigs, sequences, targets, transcripts, exons):
        self.kmers = numpy.ascontiguousarray(kmers)
        self.contigs = numpy.ascontiguousarray(contigs)
        self.sequences = numpy.ascontiguousarray(sequences)
        self.targets = numpy.ascontiguousarray(targets)
        self.transcripts = transcripts
        self.exons = exons
        if self.kmers.dtype.itemsize != 16 or self.contigs.dtype.itemsize != 48 \
                or self.targets.dtype.itemsize != 8 or self.sequences.dtype.itemsize != 1:
            raise ValueError('index arrays do not have the KMerIndex layout')
        self._handles = {}
        self._lock = threading.Lock()

    # -- device residency -------------------------------------------------
    def device_handle(self, device=0):
        """Opaque skm_index* for `device`, created on first use."""
        with self._lock:
            handle = self._handles.get(device)
            if handle is None:
                out = ctypes.c_void_p()
                _native.check(_native.hip().skm_index_create(
                    self.kmers.ctypes.data, self.kmers.size,
                    self.contigs.ctypes.data, self.contigs.size,
                    self.sequences.ctypes.data, self.sequences.size,
                    self.targets.ctypes.data, self.targets.size,
                    device, ctypes.byref(out)))
                handle = out
                self._handles[device] = handle
            return handle

    def device_info(self, device=0):
        """skm_index_info of the device copy, as a dict."""
        info = (ctypes.c_int64 * 8)()
        _native.check(_native.hip().skm_index_info(self.device_handle(device), info))
        names = ('n_slots', 'n_contigs', 'n_bases', 'n_targets', 'max_target_count', 'device_bytes',
                 'edge_windows', 'sorted_targets')
        out = dict(zip(names, info))
        layout = (ctypes.c_int64 * 8)()
        _native.check(_native.hip().skm_index_layout(self.device_handle(device), layout))
        out.update(zip(('bucketed', 'buckets', 'bucket_kmers', 'bucket_overflowed', 'kmers_twice',
                        'slots_unreached', 'successors', 'signature_slots'), layout))
        return out

    def release(self):
        with self._lock:
            for handle in self._handles.values():
                _native.hip().skm_index_destroy(handle)
            self._handles.clear()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # -- persistence --------------------------------------------------------
    def save(self, path):
        """Save the index (flat .npz container, see module docstring)."""
        with open(str(path), 'wb') as f:
            numpy.savez(f, seekmer_version=numpy.asarray(_INDEX_VERSION),
                        kmers=self.kmers, contigs=self.contigs, sequences=self.sequences,
                        targets=self.targets, transcripts=numpy.asarray(self.transcripts),
                        exons=numpy.asarray(self.exons))

    @classmethod
    def load(cls, path):
        """Load an index written by :meth:`save`.  The big arrays are memory-mapped
        straight out of the (uncompressed) container instead of being copied through
        the zip reader -- 2.2 GB at 190k transcripts, 2.0 s of a `seekmer infer` run --
        so the pages go from the page cache to the GPU upload and nowhere else."""
        mapped = _map_npz_members(str(path))
        with numpy.load(str(path), allow_pickle=False) as data:
            if str(data['seekmer_version']) != _INDEX_VERSION:
                raise RuntimeError('invalid index version.')     # seekmer/_common.pyx:303-304

            def member(name):
                return mapped[name] if name in mapped else data[name]

            return cls(member('kmers'), member('contigs'), member('sequences'), member('targets'),
                       data['transcripts'], data['exons'])


def _map_npz_members(path, names=('kmers', 'contigs', 'sequences', 'targets')):
    """numpy.memmap views of the named members of an .npz, for members that are stored
    (not deflated) plain-dtype .npy files; anything else is left to numpy.load."""
    import struct
    import zipfile
    out = {}
    try:
        with zipfile.ZipFile(path) as archive, open(path, 'rb') as raw:
            for info in archive.infolist():
                name = info.filename[:-4] if info.filename.endswith('.npy') else info.filename
                if name not in names or info.compress_type != zipfile.ZIP_STORED:
                    continue
                raw.seek(info.header_offset)
                header = raw.read(30)
                if header[:4] != b'PK\x03\x04':
                    continue
                name_len, extra_len = struct.unpack('<HH', header[26:30])
                raw.seek(info.header_offset + 30 + name_len + extra_len)
                version = numpy.lib.format.read_magic(raw)
                if version == (1, 0):
                    shape, fortran, dtype = numpy.lib.format.read_array_header_1_0(raw)
                elif version == (2, 0):
                    shape, fortran, dtype = numpy.lib.format.read_array_header_2_0(raw)
                else:
                    continue
                if fortran or dtype.hasobject or len(shape) != 1 or shape[0] == 0:
                    continue
                out[name] = numpy.memmap(path, dtype=dtype, mode='r', offset=raw.tell(), shape=shape)
    except (OSError, ValueError, zipfile.BadZipFile):
        return {}
    return out


# ------------------------------------------------------------------ file input
@contextlib.contextmanager
def decompress_and_open(path):
    """Open `path` for binary reading, decompressing by suffix
    (seekmer/common.py:23-75: external zcat/bzcat/xzcat first, stdlib second)."""
    path = pathlib.Path(path)
    tools = {'.gz': ('zcat', gzip), '.bz2': ('bzcat', bz2), '.xz': ('xzcat', lzma),
             '.lzma': ('xzcat', lzma)}
    if path.suffix in tools:
        tool, module = tools[path.suffix]
        try:
            process = subprocess.Popen([tool, str(path)], stdout=subprocess.PIPE)
        except OSError:
            process = None
        if process is not None:
            with process:
                yield process.stdout
        else:
            with module.open(str(path), 'rb') as raw, io.BufferedReader(raw) as f:
                yield f
    else:
        with path.open('rb') as f:
            yield f


def read_fasta(path):
    """Yield (name, sequence) of a FASTA file (seekmer/common.py:78-105)."""
    name = None
    chunks = []
    with decompress_and_open(path) as file:
        for line in file:
            if line[0] != ord(b'>'):
                chunks.append(line.strip())
                continue
            if name is not None:
                yield name, b''.join(chunks)
            name = line[1:].strip()
            chunks = []
        if name is not None:
            yield name, b''.join(chunks)


def iterate_by_group(iterator, group_size):
    """Groups of `group_size` consecutive items (seekmer/common.py:108-123)."""
    return zip(*([iter(iterator)] * group_size))


def feed_single_ended_reads(*paths):
    """Yield (count, names, reads) batches (seekmer/common.py:126-158)."""
    names, reads = [], []
    for path in paths:
        with decompress_and_open(path) as file:
            for i, line in enumerate(file):
                phase = i & 3
                if phase == 0:
                    names.append(line.strip()[1:])
                elif phase == 1:
                    reads.append(line.strip())
                    if len(names) >= BUFFER_SIZE:
                        yield len(names), names, reads
                        names, reads = [], []
    if reads:
        yield len(names), names, reads


def feed_pair_ended_reads(*paths):
    """Yield (count, names, interleaved mates) batches (seekmer/common.py:161-197)."""
    if len(paths) % 2 != 0:
        raise ValueError('cannot process odd numbers of pair-ended files')
    names, reads = [], []
    for path1, path2 in iterate_by_group(paths, 2):
        with decompress_and_open(path1) as file1, decompress_and_open(path2) as file2:
            for i, (line1, line2) in enumerate(zip(file1, file2)):
                phase = i & 3
                if phase == 0:
                    names.append(line1.strip()[1:])
                elif phase == 1:
                    reads.append(line1.strip())
                    reads.append(line2.strip())
                    if len(names) >= BUFFER_SIZE:
                        yield len(names), names, reads
                        names, reads = [], []
    if reads:
        yield len(names), names, reads


# ----------------------------------------------------------- native batch feed
class ReadBatch:
    """A batch in the flat layout the C ABI takes: `bases` (uint8, reads back
    to back) and `offsets` (int64[n_reads + 1]).  Unpacks like the reference's
    (count, names, reads) triple, so code written against the feeders works."""

    __slots__ = ('count', 'bases', 'offsets', 'paired', '_names', '_name_offsets', 'first_unit', 'uniform_len')

    def __init__(self, count, bases, offsets, paired, names=None, name_offsets=None, first_unit=None,
                 uniform_len=None):
        self.first_unit = first_unit          # index of the batch's first unit in its sample, if known
        self.uniform_len = uniform_len        # the length every read of the batch has, if they all agree
        self.count = int(count)
        self.bases = bases
        self.offsets = offsets
        self.paired = bool(paired)
        self._names = names
        self._name_offsets = name_offsets

    @property
    def names(self):
        if self._names is None:
            return [b''] * self.count
        raw = self._names.tobytes()
        o = self._name_offsets
        return [raw[o[i]:o[i + 1]] for i in range(self.count)]

    @property
    def reads(self):
        raw = self.bases.tobytes()
        o = self.offsets
        return [raw[o[i]:o[i + 1]] for i in range(o.size - 1)]

    def __iter__(self):
        return iter((self.count, self.names, self.reads))

    @classmethod
    def from_lists(cls, count, names, reads):
        """Pack the reference's triple (paired iff len(reads) == 2 * count,
        seekmer/_mapper.pyx:73-75)."""
        offsets = numpy.zeros(len(reads) + 1, dtype=numpy.int64)
        numpy.cumsum([len(r) for r in reads], out=offsets[1:])
        bases = numpy.frombuffer(b''.join(reads) + b'\0', dtype=numpy.uint8)
        batch = cls(count, bases, offsets, count != len(reads))
        if names is not None:
            name_offsets = numpy.zeros(len(names) + 1, dtype=numpy.int64)
            numpy.cumsum([len(n) for n in names], out=name_offsets[1:])
            batch._names = numpy.frombuffer(b''.join(names) + b'\0', dtype=numpy.uint8)
            batch._name_offsets = name_offsets
        return batch


class _FastqReader:
    """Owns one skm_fastq handle; closed when the feeder loop and every batch
    that borrowed a slab from it are gone."""

    def __init__(self, names, paired, batch_units, threads=0, pinned=False, shard=None):
        array = (ctypes.c_char_p * len(names))(*names)
        self.handle = ctypes.c_void_p()
        self.lock = threading.Lock()
        self.parallel = False
        host = _native.host()
        _native.check_host(host.skm_fastq_open(
            array, len(names), int(paired), batch_units, ctypes.byref(self.handle)),
            'skm_fastq_open')
        if pinned:          # slabs in page-locked memory: batches cross PCIe by DMA
            hip = _native.hip()
            _native.check_host(host.skm_fastq_set_allocator(
                self.handle, ctypes.cast(hip.skm_pinned_alloc, ctypes.c_void_p),
                ctypes.cast(hip.skm_pinned_free, ctypes.c_void_p)), 'skm_fastq_set_allocator')
        if shard is not None and shard[1] > 1:
            _native.check_host(host.skm_fastq_set_shard(self.handle, int(shard[0]), int(shard[1])),
                               'skm_fastq_set_shard')
        if threads > 0:
            enabled = ctypes.c_int(0)
            _native.check_host(host.skm_fastq_set_parallel(self.handle, int(threads), ctypes.byref(enabled)),
                               'skm_fastq_set_parallel')
            self.parallel = bool(enabled.value)

    def recycle(self, slab):
        with self.lock:
            _native.host().skm_fastq_recycle(self.handle, slab)

    def close(self):
        with self.lock:
            if self.handle:
                _native.host().skm_fastq_close(self.handle)
                self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Slab:
    """Storage of one batch, detached from the reader (skm_fastq_detach) and
    given back for re-use when the batch is dropped."""

    def __init__(self, reader, slab):
        self.reader = reader
        self.slab = slab

    def __del__(self):
        try:
            self.reader.recycle(self.slab)
        except Exception:
            pass


class NativeReadFeeder:
    """FASTQ batches from the native reader (skm_fastq_*): the same batching
    rule as the feeders above, without per-read Python objects.  The arrays of
    a batch are views of a slab owned by the reader; the slab goes back to the
    reader when the batch is dropped, so large batches do not pay for fresh
    pages again.  Compressed inputs are piped through zcat/bzcat/xzcat as the
    reference does.

    ``threads`` > 0 asks for the parallel engine (plain, uncompressed files whose line counts
    are multiples of four: whole batches parsed side by side, handed out in file order; any
    other input silently takes the sequential engine).  ``pinned`` puts the slabs in
    page-locked memory (needs the GPU library).  Every batch carries ``first_unit``, its place
    in the sample.  ``shard=(rank, world)``: this feeder hands out every world-th batch of the
    sample starting with batch `rank` (one feeder per GPU, no coordination needed)."""

    def __init__(self, paths, paired, batch_units=BUFFER_SIZE, threads=0, pinned=False, shard=None):
        paths = [pathlib.Path(p) for p in paths]
        if paired and len(paths) % 2 != 0:
            raise ValueError('cannot process odd numbers of pair-ended files')
        self.paths = paths
        self.paired = bool(paired)
        self.batch_units = int(batch_units)
        self.threads = int(threads)
        self.pinned = bool(pinned)
        self.shard = shard                   # (rank, world): this feeder's share of the sample's batches
        self.parallel = None                 # set when iteration starts

    def __iter__(self):
        tools = {'.gz': 'zcat', '.bz2': 'bzcat', '.xz': 'xzcat', '.lzma': 'xzcat'}
        processes, names = [], []
        host = _native.host()
        reader = None
        try:
            for path in self.paths:
                if path.suffix in tools:
                    process = subprocess.Popen([tools[path.suffix], str(path)],
                                               stdout=subprocess.PIPE)
                    processes.append(process)
                    names.append(('/dev/fd/%d' % process.stdout.fileno()).encode())
                else:
                    names.append(str(path).encode())
            threads = 0 if processes else self.threads          # (pipes are read sequentially)
            reader = _FastqReader(names, self.paired, self.batch_units, threads, self.pinned, self.shard)
            self.parallel = reader.parallel
            n = ctypes.c_int64()
            k = ctypes.c_int64()
            p_bases, p_off = ctypes.c_void_p(), ctypes.c_void_p()
            p_names, p_noff = ctypes.c_void_p(), ctypes.c_void_p()
            while True:
                _native.check_host(host.skm_fastq_next(
                    reader.handle, ctypes.byref(n), ctypes.byref(p_bases), ctypes.byref(p_off),
                    ctypes.byref(p_names), ctypes.byref(p_noff)), 'skm_fastq_next')
                if n.value == 0:
                    break
                slab = ctypes.c_void_p()
                _native.check_host(host.skm_fastq_detach(reader.handle, ctypes.byref(slab)),
                                   'skm_fastq_detach')
                owner = _Slab(reader, slab)
                def view(pointer, count, ctype, dtype):
                    # the array keeps the ctypes block alive and the block the slab
                    block = (ctype * count).from_address(pointer.value)
                    block._slab = owner
                    return numpy.frombuffer(block, dtype=dtype)

                n_reads = n.value * (2 if self.paired else 1)
                offsets = view(p_off, n_reads + 1, ctypes.c_int64, numpy.int64)
                bases = view(p_bases, int(offsets[-1]) + 1, ctypes.c_uint8, numpy.uint8)
                name_offsets = view(p_noff, n.value + 1, ctypes.c_int64, numpy.int64)
                name_bytes = view(p_names, max(int(name_offsets[-1]), 1), ctypes.c_uint8, numpy.uint8)
                _native.check_host(host.skm_fastq_batch_index(reader.handle, ctypes.byref(k)),
                                   'skm_fastq_batch_index')
                same = ctypes.c_int64(-1)
                _native.check_host(host.skm_fastq_batch_read_length(reader.handle, ctypes.byref(same)),
                                   'skm_fastq_batch_read_length')
                batch = ReadBatch(n.value, bases, offsets, self.paired, name_bytes, name_offsets,
                                  first_unit=k.value * self.batch_units,
                                  uniform_len=same.value if same.value >= 0 else None)
                yield batch
                del batch, owner, bases, offsets, name_bytes, name_offsets, view
        finally:
            # The native reader holds its own descriptors of the decompressors' pipes
            # (/dev/fd/N): close them first, or a zcat blocked on a full pipe never sees
            # SIGPIPE and wait() below never returns (early exit from the loop, an error
            # while mapping).
            if reader is not None:
                reader.close()
            for process in processes:
                process.stdout.close()
                if process.poll() is None:
                    process.terminate()
                process.wait()


# ------------------------------------------------------------- packed pieces
PACKED_CUT = -1        # include/seekmer_hip.h: SKM_PACKED_CUT


class PackedReads:
    """Reads as the mapper takes them: 2-bit code words (32 bases per uint64, first base in the top
    two bits; seekmer/_kmer.pxd:253-273), lengths, and -- only for reads that hold a character
    other than upper-case ACGT -- the bit plane the SIFT4 checks need (seekmer/_mapper.pyx:500-501).
    ``stream`` 0 = single-end reads or mate 1, 1 = mate 2; ``first_read`` = the unit reads[0]
    belongs to.  Wraps an ``skm_packed_reads`` whose arrays belong to a native reader (valid until
    the reader's next piece) or to numpy arrays kept alive here."""

    __slots__ = ('raw', '_keep', 'paired')

    def __init__(self, raw, keep=None, paired=None):
        self.raw = raw
        self._keep = keep
        self.paired = bool(raw.stream) if paired is None else bool(paired)    # one of two streams?

    stream = property(lambda self: self.raw.stream)
    first_read = property(lambda self: self.raw.first_read)
    n_reads = property(lambda self: self.raw.n_reads)
    code_words = property(lambda self: self.raw.code_words)
    uniform_len = property(lambda self: self.raw.uniform_len if self.raw.uniform_len >= 0 else None)

    @staticmethod
    def _view(pointer, count, ctype, dtype):
        if not pointer or count <= 0:
            return numpy.zeros(0, dtype=dtype)
        return numpy.frombuffer((ctype * count).from_address(pointer), dtype=dtype)

    @property
    def codes(self):
        """uint64[n_reads, code_words] (a view)"""
        r = self.raw
        flat = self._view(r.codes, (r.n_reads - 1) * r.read_stride + r.code_words if r.n_reads else 0,
                          ctypes.c_uint64, numpy.uint64)
        if r.n_reads == 0:
            return flat.reshape(0, max(r.code_words, 1))
        return numpy.lib.stride_tricks.as_strided(flat, (r.n_reads, r.code_words), (8 * r.read_stride, 8),
                                                  writeable=False)

    @property
    def lengths(self):
        r = self.raw
        if r.lengths:
            return self._view(r.lengths, r.n_reads, ctypes.c_uint32, numpy.uint32)
        return numpy.full(r.n_reads, r.uniform_len, dtype=numpy.uint32)

    @property
    def exceptions(self):
        """(read indices uint32[n], bit planes uint32[n, code_words])"""
        r = self.raw
        reads = self._view(r.exception_reads, r.n_exceptions, ctypes.c_uint32, numpy.uint32)
        masks = self._view(r.exception_masks, r.n_exceptions * r.code_words, ctypes.c_uint32, numpy.uint32)
        return reads, masks.reshape(r.n_exceptions, max(r.code_words, 1))

    @property
    def names(self):
        r = self.raw
        if not r.name_offsets:
            return None
        offsets = self._view(r.name_offsets, r.n_reads + 1, ctypes.c_int64, numpy.int64)
        raw = ctypes.string_at(r.names, int(offsets[-1])) if offsets[-1] else b''
        return [raw[offsets[i]:offsets[i + 1]] for i in range(r.n_reads)]

    is_cut = property(lambda self: self.raw.n_reads == 0 and self.raw.code_words == PACKED_CUT)

    @classmethod
    def cut(cls, stream, first_read, paired=True):
        """The piece that drops what `stream` holds from `first_read` on (SKM_PACKED_CUT)."""
        raw = _native.PackedReads()
        raw.stream, raw.code_words, raw.first_read, raw.n_reads = int(stream), PACKED_CUT, int(first_read), 0
        raw.uniform_len = -1
        return cls(raw, paired=paired)

    def copy(self):
        """The same piece over arrays of its own."""
        if self.is_cut:
            return PackedReads.cut(self.stream, self.first_read, self.paired)
        return PackedReads.from_arrays(self.stream, self.first_read, numpy.array(self.codes), numpy.array(self.lengths),
                                       *[numpy.array(a) for a in self.exceptions], paired=self.paired)

    @classmethod
    def from_arrays(cls, stream, first_read, codes, lengths, exception_reads=None, exception_masks=None,
                    paired=None):
        codes = numpy.ascontiguousarray(codes, dtype=numpy.uint64)
        if codes.ndim != 2:
            raise ValueError('codes must be [n_reads, code_words]')
        lengths = numpy.ascontiguousarray(lengths, dtype=numpy.uint32)
        n_reads, code_words = codes.shape
        if lengths.shape != (n_reads,):
            raise ValueError('one length per read')
        if exception_reads is None:
            exception_reads = numpy.zeros(0, dtype=numpy.uint32)
            exception_masks = numpy.zeros((0, code_words), dtype=numpy.uint32)
        exception_reads = numpy.ascontiguousarray(exception_reads, dtype=numpy.uint32)
        exception_masks = numpy.ascontiguousarray(exception_masks, dtype=numpy.uint32).reshape(-1, max(code_words, 1))
        raw = _native.PackedReads()
        raw.stream, raw.code_words, raw.first_read, raw.n_reads = int(stream), code_words, int(first_read), n_reads
        raw.read_stride = code_words
        raw.uniform_len = int(lengths[0]) if n_reads and (lengths == lengths[0]).all() else -1
        raw.codes = codes.ctypes.data
        raw.lengths = lengths.ctypes.data
        raw.n_exceptions = exception_reads.size
        raw.exception_reads = exception_reads.ctypes.data if exception_reads.size else None
        raw.exception_masks = exception_masks.ctypes.data if exception_reads.size else None
        return cls(raw, keep=(codes, lengths, exception_reads, exception_masks), paired=paired)

    @classmethod
    def from_ascii(cls, bases, offsets, stream=0, first_read=0, variant=-1, paired=None):
        """Pack reads held as bases back to back + offsets (skm_pack_reads)."""
        offsets = numpy.ascontiguousarray(offsets, dtype=numpy.int64)
        bases = numpy.ascontiguousarray(bases, dtype=numpy.uint8)
        n_reads = offsets.size - 1
        longest = int(numpy.diff(offsets).max()) if n_reads else 0
        code_words = max(1, (longest + 31) // 32)
        codes = numpy.zeros((n_reads, code_words), dtype=numpy.uint64)
        lengths = numpy.zeros(n_reads, dtype=numpy.uint32)
        cap = 1024
        while True:
            exc_reads = numpy.zeros(cap, dtype=numpy.uint32)
            exc_masks = numpy.zeros((cap, code_words), dtype=numpy.uint32)
            n_exc = ctypes.c_int64()
            code = _native.host().skm_pack_reads(
                bases.ctypes.data, _native.ptr(offsets, _native.c_i64p), n_reads, code_words, codes.ctypes.data,
                lengths.ctypes.data, exc_reads.ctypes.data, exc_masks.ctypes.data, cap, ctypes.byref(n_exc), variant)
            if code == _native.SKM_ERR_STATE and n_exc.value > cap:
                cap = n_exc.value
                continue
            _native.check_host(code, 'skm_pack_reads')
            break
        return cls.from_arrays(stream, first_read, codes, lengths, exc_reads[:n_exc.value], exc_masks[:n_exc.value],
                               paired=paired)


class Prefault:
    """The page tables of plain FASTQ files set up ahead of their reader (skm_fastq_prefault_*): helper
    threads map the files and touch their pages while the caller still loads its index.  ``finish()``
    (or leaving the ``with`` block) once the reader has opened the files -- or never needed them."""

    def __init__(self, paths, threads=4):
        self.handle = ctypes.c_void_p()
        paths = [pathlib.Path(p) for p in paths]
        if not paths or not PackedReadFeeder.eligible(paths):
            return
        names = [str(p).encode() for p in paths]
        array = (ctypes.c_char_p * len(names))(*names)
        if _native.host().skm_fastq_prefault_start(array, len(names), max(1, int(threads)), ctypes.byref(self.handle)) != 0:
            self.handle = ctypes.c_void_p()           # (only an optimisation: the reader reports what is wrong)

    def finish(self):
        if self.handle:
            _native.host().skm_fastq_prefault_finish(self.handle)
            self.handle = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.finish()

    def __del__(self):
        try:
            self.finish()
        except Exception:
            pass


class PackedReadFeeder:
    """Plain FASTQ files straight to PackedReads pieces in one pass over the text
    (skm_fastq_packed_*): the reference's feeders (seekmer/common.py:126-197) for the mapper's own
    input format.  Pieces of a paired sample come as two streams (mate 1 files, mate 2 files), each
    numbered by unit; a piece is valid until the next one is asked for (``copy()`` keeps it).
    ``eligible(paths)``: plain files only -- compressed inputs go through NativeReadFeeder.

    ``shard=(rank, world)``: this process reads units [T rank / world, T (rank + 1) / world) of the
    sample's T units, numbered as one process numbers them.  The reference's records are lines
    4u .. 4u + 3 of a file whatever they hold, so finding unit u takes the newlines before it: every
    rank counts the newlines of every world-th 8 MiB chunk of every file, ``sum_over_ranks`` (an
    all-reduce of an int64 array, e.g. parallel.Ranks.sum_int64) adds the tables up, and each rank
    walks the one chunk that holds its first line and the one that holds its last
    (skm_fastq_count_newlines / _locate_line): memchr over 1 / world of the text, which is then
    parsed once, by the rank that owns it."""

    COUNT_CHUNK = 8 << 20

    def __init__(self, paths, paired, threads=0, chunk_bytes=0, pinned=False, want_names=False, shard=None,
                 sum_over_ranks=None):
        paths = [pathlib.Path(p) for p in paths]
        if paired and len(paths) % 2 != 0:
            raise ValueError('cannot process odd numbers of pair-ended files')
        self.paths = paths
        self.paired = bool(paired)
        self.threads = int(threads)
        self.chunk_bytes = int(chunk_bytes)
        self.pinned = bool(pinned)
        self.want_names = bool(want_names)
        self.shard = tuple(shard) if shard is not None and shard[1] > 1 else None
        self.sum_over_ranks = sum_over_ranks
        if self.shard is not None and sum_over_ranks is None:
            raise ValueError('a sharded reader needs the ranks\' sum (sum_over_ranks)')
        self.stats = None
        self.share = None             # (begin bytes, end bytes, first unit, units) of this rank once located

    def locate_share(self):
        """Byte ranges of this rank's units in every file and the number of its first unit."""
        if self.share is not None:
            return self.share
        host = _native.host()
        rank, world = self.shard
        sizes = [p.stat().st_size for p in self.paths]
        chunks = [(size + self.COUNT_CHUNK - 1) // self.COUNT_CHUNK for size in sizes]
        table = numpy.zeros(sum(chunks) + len(self.paths), dtype=numpy.int64)
        views, at = [], 0
        for path, n in zip(self.paths, chunks):
            view = table[at:at + n]
            open_line = ctypes.c_int(0)
            _native.check_host(host.skm_fastq_count_newlines(
                str(path).encode(), self.COUNT_CHUNK, rank, world, max(1, self.threads),
                _native.ptr(view, _native.c_i64p) if n else None, n, ctypes.byref(open_line)), 'skm_fastq_count_newlines')
            views.append((at, n, open_line.value))
            at += n
        table = self.sum_over_ranks(table)
        counts = [numpy.ascontiguousarray(table[a:a + n]) for a, n, _ in views]
        lines = [int(c.sum()) + tail for c, (_, _, tail) in zip(counts, views)]
        reads = [(n + 2) // 4 for n in lines]          # a line 4u + 1 makes a read (a trailing name line alone does not)
        step = 2 if self.paired else 1
        units = [min(reads[f:f + step]) for f in range(0, len(reads), step)]     # zip(file1, file2)
        total = sum(units)
        lo, hi = total * rank // world, total * (rank + 1) // world
        begin = numpy.zeros(len(self.paths), dtype=numpy.int64)
        end = numpy.zeros(len(self.paths), dtype=numpy.int64)
        first = 0
        for pair, n_units in enumerate(units):
            a = min(max(lo - first, 0), n_units)
            b = min(max(hi - first, 0), n_units)
            for f in range(pair * step, pair * step + step):
                for line, out in ((4 * a, begin), (4 * b, end)):
                    where = ctypes.c_int64()
                    _native.check_host(host.skm_fastq_locate_line(
                        str(self.paths[f]).encode(), self.COUNT_CHUNK,
                        _native.ptr(counts[f], _native.c_i64p) if counts[f].size else None, counts[f].size, line,
                        ctypes.byref(where)), 'skm_fastq_locate_line')
                    out[f] = where.value
            first += n_units
        self.share = (begin, end, lo, hi - lo)
        return self.share

    @staticmethod
    def eligible(paths):
        return all(pathlib.Path(p).suffix not in ('.gz', '.bz2', '.xz', '.lzma') and pathlib.Path(p).is_file()
                   for p in paths)

    def open(self):
        """The native reader (an skm_fastq_packed handle owner) for callers that drain it natively."""
        return _PackedReader(self)

    def __iter__(self):
        reader = self.open()
        try:
            while True:
                piece = reader.next()
                if piece is None:
                    break
                yield piece
        finally:
            self.stats = reader.stats()
            reader.close()


class _PackedReader:
    def __init__(self, feeder):
        self.paired = feeder.paired
        host = _native.host()
        names = [str(p).encode() for p in feeder.paths]
        array = (ctypes.c_char_p * len(names))(*names)
        self.handle = ctypes.c_void_p()
        if feeder.shard is None:
            _native.check_host(host.skm_fastq_packed_open(
                array, len(names), int(feeder.paired), feeder.threads, feeder.chunk_bytes, int(feeder.want_names),
                ctypes.byref(self.handle)), 'skm_fastq_packed_open')
        else:
            begin, end, first_unit, _ = feeder.locate_share()
            _native.check_host(host.skm_fastq_packed_open_ranges(
                array, len(names), int(feeder.paired), feeder.threads, feeder.chunk_bytes, int(feeder.want_names),
                _native.ptr(begin, _native.c_i64p), _native.ptr(end, _native.c_i64p), first_unit,
                ctypes.byref(self.handle)), 'skm_fastq_packed_open_ranges')
        if feeder.pinned:
            hip = _native.hip()
            _native.check_host(host.skm_fastq_packed_set_allocator(
                self.handle, ctypes.cast(hip.skm_pinned_alloc, ctypes.c_void_p),
                ctypes.cast(hip.skm_pinned_free, ctypes.c_void_p)), 'skm_fastq_packed_set_allocator')

    def next(self):
        raw = _native.PackedReads()
        _native.check_host(_native.host().skm_fastq_packed_next(self.handle, ctypes.byref(raw)),
                           'skm_fastq_packed_next')
        if raw.n_reads == 0 and raw.code_words != PACKED_CUT:
            return None
        return PackedReads(raw, keep=self, paired=self.paired)       # (a piece or a cut)

    def estimate(self):
        """About how many units the files hold (file size over the first record's extent)."""
        units = ctypes.c_int64()
        _native.check_host(_native.host().skm_fastq_packed_estimate(self.handle, ctypes.byref(units)),
                           'skm_fastq_packed_estimate')
        return units.value

    def stats(self):
        out = (ctypes.c_int64 * 8)()
        _native.check_host(_native.host().skm_fastq_packed_stats(self.handle, out), 'skm_fastq_packed_stats')
        return {'accepted': out[0], 'reparsed': out[1], 'reads': out[2], 'exceptions': out[3], 'variant': out[4],
                'units': out[5]}

    def close(self):
        if self.handle:
            _native.host().skm_fastq_packed_close(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
