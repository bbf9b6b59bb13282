"""Random 16-byte gather ceiling of the GPU (independent and dependent), by table size."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekmer_amd import _native   # noqa: E402

hip = _native.hip()
for mb in (16, 128, 512, 2048, 8192):
    for chain, per_lane, blocks in ((0, 256, 4096), (1, 64, 4096), (1, 64, 1024)):
        rate = ctypes.c_double()
        _native.check(hip.skm_device_gather_ceiling(0, mb << 20, blocks, per_lane, chain, ctypes.byref(rate)))
        lanes = blocks * 256
        extra = ''
        if chain:
            extra = '  -> %.0f ns per dependent gather per lane' % (lanes / rate.value * 1e9)
        print('table %5d MiB  %s  blocks %4d: %7.2f G gathers/s = %7.1f GB/s of 16 B (%.0f GB/s of 64 B sectors)%s'
              % (mb, 'dependent  ' if chain else 'independent', blocks, rate.value / 1e9, rate.value * 16 / 1e9,
                 rate.value * 64 / 1e9, extra), flush=True)
