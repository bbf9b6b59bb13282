"""One rank sends a mapper's table to itself over RCCL (ncclSend / ncclRecv in one group) and merges it into
another mapper: the data path of skm_mapper_exchange_tables on one GPU.  Run under `timeout`."""
import ctypes
import os
import sys

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from seekmer_amd import _native, common, index_builder, mapper, parallel, synth   # noqa: E402

ids, pool, tx_offsets = synth.transcriptome(7, 80)
index = index_builder.build_pooled(ids, pool, tx_offsets)
n = 60000
bases, offsets = synth.reads(7, pool, tx_offsets, 0, n, 100, True)
whole = mapper.MapResult(index)
mapper.ReadMapper(index, whole).map_batch(common.ReadBatch(n, bases, offsets, True))
parts = []
for lo, hi in ((0, 25000), (25000, n)):
    part = mapper.MapResult(index)
    mapper.ReadMapper(index, part).map_batch_async(
        common.ReadBatch(hi - lo, bases, np.ascontiguousarray(offsets[2 * lo:2 * hi + 1]), True, first_unit=lo))
    part.sync()
    parts.append(part)
hip = _native.hip()
raw = ctypes.create_string_buffer(128)
_native.check(hip.skm_comm_unique_id(raw))
comm = parallel.create_comm(0, raw.raw, 0, 1)
try:
    _native.check(hip.skm_mapper_exchange_tables(parts[1]._handle, 0, parts[0]._handle, 0, comm))
finally:
    parallel.destroy_comm(comm)
for got, want in zip(parts[0].export(), whole.export()):
    assert np.array_equal(got, want)
assert parts[0].sizes() == whole.sizes()
print('exchange ok', parts[0].sizes())
