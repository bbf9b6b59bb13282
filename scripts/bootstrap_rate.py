"""Rate of the `-b N` branch (BASELINE.json configs[4] shape): N multinomial resamplings of the
class counts + EM from the main estimate, on the resident-size class table.
    python3 scripts/bootstrap_rate.py --pairs 10000000 --boot 20 --cache /tmp/skm_idx.npz
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekmer_amd import _native, common, index_builder, infer, mapper, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--genes', type=int, default=20000)
    ap.add_argument('--pairs', type=int, default=10_000_000)
    ap.add_argument('--boot', type=int, default=20)
    ap.add_argument('--cache', default='')
    args = ap.parse_args()
    ids, pool, tx_offsets = synth.transcriptome(1, args.genes)
    if args.cache and os.path.exists(args.cache):
        index = common.KMerIndex.load(args.cache)
    else:
        index = index_builder.build_pooled(ids, pool, tx_offsets)
        if args.cache:
            index.save(args.cache)
    bases, offsets = synth.reads(1, pool, tx_offsets, 0, args.pairs, 100, True)
    hip = _native.hip()
    index.device_handle(0)
    d_bases, d_off = ctypes.c_void_p(), ctypes.c_void_p()
    _native.check(hip.skm_device_malloc(0, bases.size, ctypes.byref(d_bases)))
    _native.check(hip.skm_device_malloc(0, offsets.size * 8, ctypes.byref(d_off)))
    _native.check(hip.skm_device_upload(0, d_bases, bases.ctypes.data, bases.size))
    _native.check(hip.skm_device_upload(0, d_off, offsets.ctypes.data, offsets.size * 8))
    result = mapper.MapResult(index)
    result.map_resident(d_bases, d_off, args.pairs, True, 100)
    t0 = time.perf_counter()
    summarized = result.summarize()
    t_sum = time.perf_counter() - t0
    t0 = time.perf_counter()
    main_tpm, iters = infer.quantify(summarized, return_iters=True)
    t_main = time.perf_counter() - t0
    print('summarize (export of %d classes) %.1f ms; main quantify %.1f ms (%d EM steps)'
          % (summarized.class_count.size, t_sum * 1e3, t_main * 1e3, iters), flush=True)
    quant = infer._QuantHandle.from_map_result(result, len(ids))
    eff = summarized.effective_lengths.astype('f8')
    x0 = main_tpm / main_tpm.sum()
    for rep in range(3):
        t0 = time.perf_counter()
        out, _, its = quant.bootstrap(args.boot, 7, x0, eff)
        dt = time.perf_counter() - t0
        print('native call alone: %d bootstraps in %.1f ms = %.2f ms each (%d EM steps in all)'
              % (args.boot, dt * 1e3, dt * 1e3 / args.boot, int(its.sum())), flush=True)
    quant.close()
    for rep in range(2):
        t0 = time.perf_counter()
        boots = infer.bootstrap_quantify(summarized, main_tpm, args.boot, seed=7)
        dt = time.perf_counter() - t0
        spread = np.std(np.asarray(boots), axis=0)
        print('%d bootstraps in %.1f ms = %.2f ms each (mean TPM std over expressed transcripts %.3g)'
              % (args.boot, dt * 1e3, dt * 1e3 / args.boot, float(spread[main_tpm > 0].mean())), flush=True)


if __name__ == '__main__':
    main()
