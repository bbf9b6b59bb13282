"""Seeded synthetic transcriptomes and reads from the native
generator in libseekmer_host.so.  Stand-in for the ENSEMBL GRCh38 cDNA index
and real FASTQ, which are not available offline."""
import ctypes
import os

import numpy

from . import _native


def transcriptome(seed, n_genes):
    """(ids, pool uint8[n + 1], offsets int64[n_tx + 1])"""
    host = _native.host()
    n_tx = ctypes.c_int64()
    pool_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
    _native.check_host(host.skm_synth_transcriptome(seed, n_genes, ctypes.byref(n_tx),
                                                    ctypes.byref(pool_p), ctypes.byref(off_p)),
                       'skm_synth_transcriptome')
    try:
        offsets = numpy.ctypeslib.as_array(ctypes.cast(off_p, _native.c_i64p),
                                           (n_tx.value + 1,)).copy()
        pool = numpy.ctypeslib.as_array(ctypes.cast(pool_p, ctypes.POINTER(ctypes.c_uint8)),
                                        (int(offsets[-1]) + 1,)).copy()
    finally:
        host.skm_synth_free(pool_p)
        host.skm_synth_free(off_p)
    ids = [b'SYNT%08d' % i for i in range(n_tx.value)]
    return ids, pool, offsets


def reads(seed, pool, offsets, first_unit, n_units, read_len, paired, n_threads=None):
    """(bases uint8[n_reads * read_len + 1], offsets int64[n_reads + 1])"""
    if n_threads is None:
        n_threads = min(16, os.cpu_count() or 1)
    n_reads = n_units * (2 if paired else 1)
    bases = numpy.zeros(n_reads * read_len + 1, dtype=numpy.uint8)
    offsets = numpy.ascontiguousarray(offsets, dtype=numpy.int64)
    _native.check_host(_native.host().skm_synth_reads(
        seed, pool.ctypes.data, _native.ptr(offsets, _native.c_i64p), offsets.size - 1,
        first_unit, n_units, read_len, int(bool(paired)), n_threads, bases.ctypes.data),
        'skm_synth_reads')
    read_offsets = numpy.arange(n_reads + 1, dtype=numpy.int64) * read_len
    return bases, read_offsets


def write_fastq(bases, n_units, read_len, paired, path1, path2=None, first_unit=0, n_threads=None):
    """The batch of `reads()` as FASTQ text (one file per mate)."""
    if n_threads is None:
        n_threads = min(16, os.cpu_count() or 1)
    _native.check_host(_native.host().skm_synth_fastq_write(
        bases.ctypes.data, n_units, read_len, int(bool(paired)), first_unit, os.fsencode(str(path1)),
        os.fsencode(str(path2)) if path2 is not None else None, n_threads), 'skm_synth_fastq_write')


def sequences_of(pool, offsets):
    raw = pool.tobytes()
    return [raw[offsets[i]:offsets[i + 1]] for i in range(offsets.size - 1)]
