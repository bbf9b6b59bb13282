"""Map-only driver for profiling: builds (or loads a cached) synthetic index,
keeps one batch resident and launches the mapper `--reps` times.
    python3 scripts/profile_map.py --genes 20000 --pairs 10000000 --reps 3 --cache /tmp/skm_idx.npz
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekmer_amd import _native, common, index_builder, mapper, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--genes', type=int, default=20000)
    ap.add_argument('--pairs', type=int, default=10_000_000)
    ap.add_argument('--read-len', type=int, default=100)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--single', action='store_true')
    ap.add_argument('--cache', default='')
    ap.add_argument('--stats', action='store_true')
    ap.add_argument('--census', action='store_true', help='census build: production paths + cycle sums')
    ap.add_argument('--again', action='store_true', help='map the batch once more without a reset')
    ap.add_argument('--sorted', action='store_true',
                    help='experiment: re-run with the units ordered by their anchor contig')
    args = ap.parse_args()
    t0 = time.time()
    ids, pool, tx_offsets = synth.transcriptome(1, args.genes)
    if args.cache and os.path.exists(args.cache):
        index = common.KMerIndex.load(args.cache)
    else:
        index = index_builder.build_pooled(ids, pool, tx_offsets)
        if args.cache:
            index.save(args.cache)
    print('index ready in %.1fs' % (time.time() - t0), flush=True)
    paired = not args.single
    bases, offsets = synth.reads(1, pool, tx_offsets, 0, args.pairs, args.read_len, paired)
    hip = _native.hip()
    index.device_handle(0)
    lengths = index.contigs['target_count'] if 'target_count' in index.contigs.dtype.names else index.contigs[index.contigs.dtype.names[-1]]
    print('device index', index.device_info(), 'contig slices > 16 targets: %.4f, > 8: %.4f, max %d' % (
        float((lengths > 16).mean()), float((lengths > 8).mean()), int(lengths.max())), flush=True)
    d_bases, d_off = ctypes.c_void_p(), ctypes.c_void_p()
    _native.check(hip.skm_device_malloc(0, bases.size, ctypes.byref(d_bases)))
    _native.check(hip.skm_device_malloc(0, offsets.size * 8, ctypes.byref(d_off)))
    _native.check(hip.skm_device_upload(0, d_bases, bases.ctypes.data, bases.size))
    _native.check(hip.skm_device_upload(0, d_off, offsets.ctypes.data, offsets.size * 8))
    result = mapper.MapResult(index, keep_spans=args.sorted)
    if args.stats or args.census:
        result.set_stats(2 if args.census else 1)
    for rep in range(args.reps):
        result.reset()
        before = result.timing()
        t0 = time.time()
        result.map_resident(d_bases, d_off, args.pairs, paired, args.read_len)
        wall = time.time() - t0
        after = result.timing()
        print('rep %d: wall %.2f ms pack %.3f map %.3f classes %.3f ms sizes %s' % (
            rep, wall * 1e3, (after['pack_ns'] - before['pack_ns']) * 1e-6,
            (after['map_ns'] - before['map_ns']) * 1e-6,
            (after['class_ns'] - before['class_ns']) * 1e-6, result.sizes()), flush=True)
    if args.again:          # the same batch once more WITHOUT a reset: every class exists already
        before = result.timing()
        result.map_resident(d_bases, d_off, args.pairs, paired, args.read_len)
        after = result.timing()
        print('again (no reset): pack %.3f map %.3f classes %.3f ms sizes %s' % (
            (after['pack_ns'] - before['pack_ns']) * 1e-6, (after['map_ns'] - before['map_ns']) * 1e-6,
            (after['class_ns'] - before['class_ns']) * 1e-6, result.sizes()), flush=True)
    if args.stats or args.census:
        print(result.access_stats())
    if args.sorted:
        rm = mapper.ReadMapper(index, result)
        units = rm.last_batch(args.pairs)
        entry = units[2].astype(np.int64)
        contig = np.where(entry < 0, ~entry, entry)
        none = units[4] == 0          # spread the unmapped units evenly (block ranges are static)
        contig[none] = np.random.default_rng(1).integers(0, int(contig.max()) + 1, int(none.sum()))
        order = np.argsort(contig, kind='stable')
        width = 2 if paired else 1
        reads2d = bases[:-1].reshape(args.pairs, width * args.read_len)[order]
        sorted_bases = np.concatenate([reads2d.reshape(-1), np.zeros(1, np.uint8)])
        _native.check(hip.skm_device_upload(0, d_bases, sorted_bases.ctypes.data, sorted_bases.size))
        for rep in range(args.reps):
            result.reset()
            before = result.timing()
            result.map_resident(d_bases, d_off, args.pairs, paired, args.read_len)
            after = result.timing()
            print('sorted rep %d: pack %.3f map %.3f classes %.3f ms sizes %s' % (
                rep, (after['pack_ns'] - before['pack_ns']) * 1e-6,
                (after['map_ns'] - before['map_ns']) * 1e-6,
                (after['class_ns'] - before['class_ns']) * 1e-6, result.sizes()), flush=True)
    counts = np.sort(result.export()[2])[::-1]
    print('class counts: top10 %s, top-100 share %.3f, top-1000 share %.3f, singletons %d' % (
        counts[:10].tolist(), counts[:100].sum() / counts.sum(), counts[:1000].sum() / counts.sum(),
        int((counts == 1).sum())))


if __name__ == '__main__':
    main()
