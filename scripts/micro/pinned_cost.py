"""hipHostMalloc cost by size (skm_pinned_alloc without the arena): what one page-locked arena costs."""
import os
import sys
import time
os.environ['SKM_PINNED_ARENA_MB'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from seekmer_amd import _native   # noqa: E402
hip = _native.hip()
_native.check(hip.skm_pinned_set_device(0))
for mb in (1, 2, 8, 32, 64, 128, 256, 512, 1, 128):
    t0 = time.perf_counter()
    p = hip.skm_pinned_alloc(mb << 20)
    t1 = time.perf_counter()
    hip.skm_pinned_free(p)
    t2 = time.perf_counter()
    print('%4d MB: alloc %.2f ms  free %.2f ms' % (mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
