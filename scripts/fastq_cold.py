"""What ONE run of `seekmer infer` sees from FASTQ text: every pass in a process of its own
(nothing mapped, nothing pooled, nothing page-locked yet), against later passes of the same
process.  Numbers for DESIGN.md, not the bench metric.

    python3 scripts/fastq_cold.py --genes 2000 --pairs 10000000 [--threads 14] [--chunk-mb 16]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(args):
    import numpy as np   # noqa: F401
    from seekmer_amd import _native, common, infer, mapper
    t_start = time.perf_counter()
    index = common.KMerIndex.load(args.index)
    index.device_handle(0)
    _native.check(_native.hip().skm_pinned_set_device(0))
    result = mapper.MapResult(index)
    rm = mapper.ReadMapper(index, result)
    t_ready = time.perf_counter()
    out = {'setup_s': t_ready - t_start, 'passes': []}
    paths = [args.r1, args.r2]
    if args.keep:
        _native.host().skm_fastq_cache_bytes(1 << 40)
    for k in range(args.passes):
        result.reset()
        t0 = time.perf_counter()
        if args.mode == 'parse':
            n = sum(piece.n_reads for piece in common.PackedReadFeeder(paths, True, threads=args.threads,
                                                                       chunk_bytes=args.chunk_mb << 20,
                                                                       pinned=not args.pageable))
            out['passes'].append({'parse_s': time.perf_counter() - t0, 'reads': n})
            continue
        feeder = common.PackedReadFeeder(paths, True, threads=args.threads, chunk_bytes=args.chunk_mb << 20,
                                         pinned=not args.pageable)
        rm(feeder)
        t_map = time.perf_counter() - t0
        infer.quantify_resident(result)
        out['passes'].append({'map_s': t_map, 'total_s': time.perf_counter() - t0, 'units': feeder.stats['units']})
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--genes', type=int, default=2000)
    ap.add_argument('--pairs', type=int, default=10_000_000)
    ap.add_argument('--threads', type=int, default=14)
    ap.add_argument('--chunk-mb', type=int, default=16)
    ap.add_argument('--passes', type=int, default=3)
    ap.add_argument('--mode', default='')
    ap.add_argument('--pageable', action='store_true')
    ap.add_argument('--keep', action='store_true')
    ap.add_argument('--index', default='')
    ap.add_argument('--r1', default='')
    ap.add_argument('--r2', default='')
    args = ap.parse_args()
    if args.mode:
        return child(args)
    from seekmer_amd import index_builder, synth
    ids, pool, tx_offsets = synth.transcriptome(1, args.genes)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    ram = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix='skm_cold_', dir=ram)
    try:
        index_path = os.path.join(work, 'index.npz')
        index.save(index_path)
        bases, _ = synth.reads(1, pool, tx_offsets, 0, args.pairs, 100, True)
        r1, r2 = os.path.join(work, 'r_1.fastq'), os.path.join(work, 'r_2.fastq')
        synth.write_fastq(bases, args.pairs, 100, True, r1, r2)
        del bases
        for mode in ('parse', 'run'):
            for extra in ([], ['--keep'], ['--pageable']):
                cmd = [sys.executable, os.path.abspath(__file__), '--mode', mode, '--index', index_path, '--r1', r1,
                       '--r2', r2, '--threads', str(args.threads), '--chunk-mb', str(args.chunk_mb),
                       '--passes', str(args.passes)] + extra
                done = subprocess.run(cmd, stdout=subprocess.PIPE, check=True)
                result = json.loads(done.stdout.decode().strip().splitlines()[-1])
                key = 'parse_s' if mode == 'parse' else 'total_s'
                rates = ['%.0f M/s' % (args.pairs / p[key] / 1e6) for p in result['passes']]
                print('%-5s %-10s setup %.1fs  passes: %s' % (mode, ' '.join(extra) or 'default', result['setup_s'],
                                                             ', '.join(rates)), flush=True)
    finally:
        import shutil
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
