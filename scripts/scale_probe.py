import sys, time, ctypes, os
sys.path.insert(0, os.getcwd())
import numpy as np
from seekmer_amd import synth, index_builder, mapper, common, infer, _native
genes = int(sys.argv[1]); n_units = int(sys.argv[2])
t=time.time(); ids, pool, offs = synth.transcriptome(1, genes); print('transcriptome', len(ids), offs[-1], time.time()-t, flush=True)
t=time.time(); index = index_builder.build_pooled(ids, pool, offs); print('index build s', time.time()-t, 'slots', index.kmers.size, 'contigs', index.contigs.size, 'targets', index.targets.size, 'max tc', index.contigs['target_count'].max(), flush=True)
t=time.time(); bases, ro = synth.reads(1, pool, offs, 0, n_units, 100, True); print('reads gen s', time.time()-t, flush=True)
hip = _native.hip()
t=time.time(); h = index.device_handle(0); print('index upload s', time.time()-t, flush=True)
d_bases = ctypes.c_void_p(); d_off = ctypes.c_void_p()
_native.check(hip.skm_device_malloc(0, bases.size, ctypes.byref(d_bases)))
_native.check(hip.skm_device_malloc(0, ro.size*8, ctypes.byref(d_off)))
_native.check(hip.skm_device_upload(0, d_bases, bases.ctypes.data, bases.size))
_native.check(hip.skm_device_upload(0, d_off, ro.ctypes.data, ro.size*8))
for rep in range(3):
    res = mapper.MapResult(index)
    t=time.time()
    _native.check(hip.skm_mapper_map_batch_device(res._handle, d_bases, d_off, n_units, 1, 100))
    t_map=time.time()-t
    print('map wall s', t_map, 'pairs/s', n_units/t_map, res.timing(), res.sizes(), flush=True)
    t=time.time()
    q = infer._QuantHandle.from_map_result(res, len(ids))
    fld = res.fragment_length_counts
    eff = res._effective_lengths(fld)
    x = np.ones(eff.size)/eff; x/=x.sum()
    x, it = q.em(x, eff)
    tpm = infer._tpm(x)
    t_q=time.time()-t
    print('quant wall s', t_q, 'iters', it, q.timing(), 'total pairs/s', n_units/(t_map+t_q), flush=True)
    q.close()
