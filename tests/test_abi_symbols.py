"""The C-ABI libraries load on a machine without a GPU and export every symbol
include/seekmer_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    with open(os.path.join(ROOT, 'include', 'seekmer_hip.h')) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(skm_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_expected_families():
    names = _declared_symbols()
    for family in ('skm_index_create', 'skm_mapper_map_batch', 'skm_mapper_export', 'skm_quant_em',
                   'skm_quant_bootstrap', 'skm_effective_lengths', 'skm_build_index',
                   'skm_fastq_next', 'skm_synth_reads', 'skm_comm_unique_id'):
        assert family in names


def test_every_declared_symbol_is_exported_and_bound(native_libs):
    hip = ctypes.CDLL(native_libs.HIP_LIB_PATH)
    host = ctypes.CDLL(native_libs.HOST_LIB_PATH)
    bound = set(native_libs.HIP_SYMBOLS) | set(native_libs.HOST_SYMBOLS)
    for name in _declared_symbols():
        assert hasattr(hip, name) or hasattr(host, name), name + ' is declared but not exported'
        assert name in bound, name + ' is declared but not bound in seekmer_amd/_native.py'
    for name in bound:
        assert name in _declared_symbols(), name + ' is bound but not declared in the header'
    native_libs.hip()
    native_libs.host()


def test_no_gpu_fails_loudly(native_libs):
    """Without a GPU every device entry point must fail, never fall back."""
    import numpy as np
    import pytest
    if native_libs.device_count() > 0:
        pytest.skip('a GPU is present')
    n = ctypes.c_int(-1)
    assert native_libs.hip().skm_device_count(ctypes.byref(n)) == native_libs.SKM_ERR_NO_DEVICE
    assert b'no HIP device' in native_libs.hip().skm_last_error()
    fld = np.zeros(2000, dtype=np.int64)
    length = np.ones(4)
    out = np.zeros(4)
    code = native_libs.hip().skm_effective_lengths(
        0, native_libs.ptr(fld, native_libs.c_i64p), native_libs.ptr(length, native_libs.c_f64p), 4,
        native_libs.ptr(out, native_libs.c_f64p))
    assert code == native_libs.SKM_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, 'seekmer_amd')
    for base, _, files in os.walk(pkg):
        for name in files:
            if name.endswith(('.py', '.cpp', '.hip', '.h')):
                with open(os.path.join(base, name), errors='replace') as f:
                    text = f.read()
                assert 'skmo_' not in text, name
                assert 'import oracle' not in text and 'from oracle' not in text, name
                assert 'libskm_oracle' not in text, name
