"""ctypes binding of include/seekmer_hip.h (libseekmer_hip.so + libseekmer_host.so).

The product path has no CPU fallback: if the HIP library is missing the
package raises at first use, and on a machine without a GPU every device call
fails with ``NativeError`` (SKM_ERR_NO_DEVICE).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get('SKM_HIP_LIB') or os.path.join(_HERE, 'libseekmer_hip.so')   # (override: tuning builds)
HOST_LIB_PATH = os.path.join(_HERE, 'libseekmer_host.so')

SKM_OK = 0
SKM_ERR_ARG = 1
SKM_ERR_HIP = 2
SKM_ERR_NO_DEVICE = 3
SKM_ERR_COLLISION = 4
SKM_ERR_STATE = 5
SKM_ERR_IO = 6
SKM_ERR_UNDEFINED = 7
SKM_ERR_COMM = 8

c_void_pp = ctypes.POINTER(ctypes.c_void_p)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i64 = ctypes.c_int64
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)


class PackedReads(ctypes.Structure):
    """skm_packed_reads (include/seekmer_hip.h)"""
    _fields_ = [('stream', ctypes.c_int32), ('code_words', ctypes.c_int32), ('first_read', ctypes.c_int64),
                ('n_reads', ctypes.c_int64), ('read_stride', ctypes.c_int64), ('uniform_len', ctypes.c_int64),
                ('codes', ctypes.c_void_p), ('lengths', ctypes.c_void_p), ('n_exceptions', ctypes.c_int64),
                ('exception_reads', ctypes.c_void_p), ('exception_masks', ctypes.c_void_p),
                ('names', ctypes.c_void_p), ('name_offsets', ctypes.c_void_p)]


class DeviceTable(ctypes.Structure):
    """skm_device_table (include/seekmer_hip.h): a mapper's class table where it lies in HBM"""
    _fields_ = [('device', ctypes.c_int32), ('n_classes', ctypes.c_int64), ('n_ids', ctypes.c_int64),
                ('class_start', ctypes.c_void_p), ('class_len', ctypes.c_void_p), ('class_count', ctypes.c_void_p),
                ('first_seen', ctypes.c_void_p), ('ids', ctypes.c_void_p), ('unaligned', ctypes.c_int64),
                ('units', ctypes.c_int64), ('first_seen_bound', ctypes.c_int64), ('fld', ctypes.c_void_p)]


# every symbol include/seekmer_hip.h declares, by library
HIP_SYMBOLS = {
    'skm_last_error': (ctypes.c_char_p, []),
    'skm_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'skm_device_malloc': (ctypes.c_int, [ctypes.c_int, c_i64, c_void_pp]),
    'skm_device_free': (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    'skm_device_upload': (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, c_i64]),
    'skm_device_download': (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, c_i64]),
    'skm_device_synchronize': (ctypes.c_int, [ctypes.c_int]),
    'skm_device_gather_ceiling': (ctypes.c_int, [ctypes.c_int, c_i64, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, c_f64p]),
    'skm_pinned_alloc': (ctypes.c_void_p, [ctypes.c_size_t]),
    'skm_pinned_free': (None, [ctypes.c_void_p]),
    'skm_pinned_set_device': (ctypes.c_int, [ctypes.c_int]),
    'skm_index_create': (ctypes.c_int, [ctypes.c_void_p, c_i64, ctypes.c_void_p, c_i64,
                                        ctypes.c_void_p, c_i64, ctypes.c_void_p, c_i64,
                                        ctypes.c_int, c_void_pp]),
    'skm_index_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_index_info': (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    'skm_index_layout': (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    'skm_mapper_create': (ctypes.c_int, [ctypes.c_void_p, c_void_pp]),
    'skm_mapper_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_mapper_map_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i64p, c_i64,
                                            ctypes.c_int]),
    'skm_mapper_map_batch_async': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i64p, c_i64,
                                                  ctypes.c_int, c_i64]),
    'skm_mapper_map_batch_uniform_async': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, c_i64,
                                                          ctypes.c_int, c_i64]),
    'skm_mapper_sync': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_mapper_expect_units': (ctypes.c_int, [ctypes.c_void_p, c_i64]),
    'skm_mapper_push_packed': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(PackedReads), ctypes.c_int]),
    'skm_mapper_map_packed_source': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                    c_i64p]),
    'skm_mapper_map_batch_device': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p, c_i64, ctypes.c_int,
                                                   ctypes.c_int32]),
    'skm_mapper_last_batch': (ctypes.c_int, [ctypes.c_void_p, c_i32p, c_i32p, c_i32p, c_i32p,
                                             c_i32p, c_i32p, c_i64, c_i64p]),
    'skm_mapper_keep_spans': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'skm_mapper_summary': (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    'skm_mapper_export': (ctypes.c_int, [ctypes.c_void_p, c_i64p, c_i32p, c_i64p, c_i64p, c_i64p]),
    'skm_mapper_merge': (ctypes.c_int, [ctypes.c_void_p, c_i64, c_i64p, c_i32p, c_i64p, c_i64p,
                                        c_i64, c_i64p]),
    'skm_mapper_device_table': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(DeviceTable)]),
    'skm_mapper_merge_device': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(DeviceTable)]),
    'skm_mapper_clear': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_mapper_reset': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_mapper_timing': (ctypes.c_int, [ctypes.c_void_p, c_f64p]),
    'skm_mapper_set_stats': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'skm_mapper_access_stats': (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    'skm_effective_lengths': (ctypes.c_int, [ctypes.c_int, c_i64p, c_f64p, c_i64, c_f64p]),
    'skm_quant_create': (ctypes.c_int, [ctypes.c_int, c_i64, c_i64, c_i64p, c_i32p, c_f64p,
                                        c_void_pp]),
    'skm_quant_create_from_mapper': (ctypes.c_int, [ctypes.c_void_p, c_i64, c_void_pp]),
    'skm_quant_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_quant_em': (ctypes.c_int, [ctypes.c_void_p, c_f64p, c_f64p, ctypes.c_double,
                                    ctypes.c_double, c_i64, c_i64, c_i64p]),
    'skm_quant_bootstrap': (ctypes.c_int, [ctypes.c_void_p, c_i64, ctypes.c_uint64, c_f64p,
                                           c_f64p, ctypes.c_double, ctypes.c_double, c_i64,
                                           c_f64p, c_i64p, c_i64p]),
    'skm_quant_bootstrap_tpm': (ctypes.c_int, [ctypes.c_void_p, c_i64, ctypes.c_uint64, c_f64p,
                                               c_f64p, ctypes.c_double, ctypes.c_double, c_i64,
                                               c_f64p, c_i64p]),
    'skm_quant_bootstrap_share_tpm': (ctypes.c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, ctypes.c_uint64, c_f64p,
                                                     c_f64p, ctypes.c_double, ctypes.c_double, c_i64,
                                                     c_f64p, c_i64p]),
    'skm_quant_set_counts': (ctypes.c_int, [ctypes.c_void_p, c_f64p]),
    'skm_quant_timing': (ctypes.c_int, [ctypes.c_void_p, c_f64p]),
    'skm_comm_unique_id': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_comm_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       c_void_pp]),
    'skm_comm_count': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    'skm_comm_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_mapper_exchange_tables': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                  ctypes.c_void_p]),
    'skm_quant_set_comm': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'skm_quant_infer': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_f64p, c_i64, ctypes.c_double,
                                       ctypes.c_double, c_i64, c_f64p, c_f64p, c_i64p]),
}

HOST_SYMBOLS = {
    'skm_build_index': (ctypes.c_int, [ctypes.c_void_p, c_i64p, c_i64, ctypes.c_int, c_void_pp]),
    'skm_built_sizes': (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    'skm_built_copy': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p]),
    'skm_built_free': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_fastq_open': (ctypes.c_int, [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int,
                                      c_i64, c_void_pp]),
    'skm_fastq_set_allocator': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'skm_fastq_set_parallel': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'skm_fastq_set_shard': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    'skm_fastq_batch_read_length': (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    'skm_fastq_batch_index': (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    'skm_fastq_next': (ctypes.c_int, [ctypes.c_void_p, c_i64p, c_void_pp, c_void_pp, c_void_pp,
                                      c_void_pp]),
    'skm_fastq_detach': (ctypes.c_int, [ctypes.c_void_p, c_void_pp]),
    'skm_fastq_recycle': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'skm_fastq_close': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_fastq_cache_bytes': (ctypes.c_int, [c_i64]),
    'skm_fastq_wait_unmapped': (ctypes.c_int, []),
    'skm_fastq_packed_open': (ctypes.c_int, [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, c_i64, ctypes.c_int, c_void_pp]),
    'skm_fastq_packed_open_ranges': (ctypes.c_int, [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, c_i64, ctypes.c_int, c_i64p, c_i64p, c_i64, c_void_pp]),
    'skm_fastq_count_newlines': (ctypes.c_int, [ctypes.c_char_p, c_i64, c_i64, c_i64, ctypes.c_int, c_i64p, c_i64,
                                                ctypes.POINTER(ctypes.c_int)]),
    'skm_fastq_locate_line': (ctypes.c_int, [ctypes.c_char_p, c_i64, c_i64p, c_i64, c_i64, c_i64p]),
    'skm_fastq_packed_set_allocator': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'skm_fastq_packed_next': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(PackedReads)]),
    'skm_fastq_packed_stats': (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    'skm_fastq_prefault_start': (ctypes.c_int, [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int, c_void_pp]),
    'skm_fastq_prefault_finish': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_fastq_packed_estimate': (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    'skm_fastq_packed_close': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_pack_reads': (ctypes.c_int, [ctypes.c_void_p, c_i64p, c_i64, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_i64, c_i64p,
                                      ctypes.c_int]),
    'skm_pack_set_variant': (ctypes.c_int, [ctypes.c_int]),
    'skm_synth_transcriptome': (ctypes.c_int, [ctypes.c_uint64, c_i64, c_i64p, c_void_pp,
                                               c_void_pp]),
    'skm_synth_free': (ctypes.c_int, [ctypes.c_void_p]),
    'skm_synth_fastq_write': (ctypes.c_int, [ctypes.c_void_p, c_i64, ctypes.c_int, ctypes.c_int, c_i64,
                                             ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]),
    'skm_synth_reads': (ctypes.c_int, [ctypes.c_uint64, ctypes.c_void_p, c_i64p, c_i64, c_i64,
                                       c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p]),
}


class NativeError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, code, message):
        super().__init__('seekmer_hip error %d: %s' % (code, message))
        self.code = code


def _bind(path, symbols):
    if not os.path.exists(path):
        raise ImportError(
            '%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C seekmer_amd/csrc` (there is no CPU fallback)' % path)
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in symbols.items():
        fn = getattr(lib, name)            # AttributeError = header and library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


_hip = None
_host = None


def hip():
    global _hip
    if _hip is None:
        _hip = _bind(HIP_LIB_PATH, HIP_SYMBOLS)
    return _hip


def host():
    global _host
    if _host is None:
        _host = _bind(HOST_LIB_PATH, HOST_SYMBOLS)
    return _host


def check(code):
    """Raise the reference's exception type for the reference's error cases."""
    if code == SKM_OK:
        return
    message = hip().skm_last_error().decode(errors='replace')
    if code == SKM_ERR_ARG:
        raise ValueError(message)
    raise NativeError(code, message)


def check_host(code, what):
    if code == SKM_OK:
        return
    if code == SKM_ERR_ARG:
        raise ValueError('%s: bad argument' % what)
    if code == SKM_ERR_IO:
        raise OSError('%s: cannot open input' % what)
    raise NativeError(code, what)


def device_count():
    n = ctypes.c_int(0)
    code = hip().skm_device_count(ctypes.byref(n))
    if code == SKM_ERR_NO_DEVICE:
        return 0
    check(code)
    return n.value


def ptr(array, ctype):
    return array.ctypes.data_as(ctype)
