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


def test_the_production_map_kernel_does_not_spill(native_libs):
    """map_units_kernel<false, true> runs at 128 VGPRs (4 waves per SIMD) WITHOUT scratch: 7 spilled registers were
    3 GB of HBM writes per 10 M pairs in the counters of round 4 (WRITE_SIZE x 7), caused by moving one branch of the
    action switch.  The build keeps the compiler's resource remarks beside the object."""
    import re
    import subprocess
    remarks = os.path.join(ROOT, 'seekmer_amd', 'csrc', 'build', 'skm_map.remarks')
    if not os.path.exists(remarks):
        subprocess.check_call(['touch', os.path.join(ROOT, 'seekmer_amd', 'csrc', 'skm_map.hip')])
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'seekmer_amd', 'csrc')], stdout=subprocess.DEVNULL)
    text = open(remarks).read()
    blocks = text.split('Function Name: ')
    mine = [b for b in blocks if b.startswith('_ZN3skm16map_units_kernelILb0ELb1EEE')]
    assert len(mine) == 1
    scratch = int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)', mine[0]).group(1))
    vgprs = int(re.search(r' VGPRs: (\d+)', mine[0]).group(1))
    spilled = int(re.search(r'VGPRs Spill: (\d+)', mine[0]).group(1))
    assert scratch == 0 and spilled == 0 and vgprs <= 128, (scratch, spilled, vgprs)
