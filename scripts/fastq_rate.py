"""Parse rate of the native FASTQ reader by engine, thread count and slab memory (host only; the
pinned variants need the GPU library).  python3 scripts/fastq_rate.py [--pairs N] [--trace]"""
import argparse
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekmer_amd import common, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=10_000_000)
    ap.add_argument('--batch', type=int, default=2_000_000)
    ap.add_argument('--threads', default='0,1,2,4,8,12,16')
    ap.add_argument('--pinned', action='store_true')
    args = ap.parse_args()
    ids, pool, tx = synth.transcriptome(1, 2000)
    bases, _ = synth.reads(1, pool, tx, 0, args.pairs, 100, True)
    folder = tempfile.mkdtemp(prefix='skm_rate_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        p1, p2 = os.path.join(folder, 'r_1.fastq'), os.path.join(folder, 'r_2.fastq')
        synth.write_fastq(bases, args.pairs, 100, True, p1, p2)
        for threads in [int(t) for t in args.threads.split(',')]:
            for rep in range(3):
                t0 = time.perf_counter()
                n = 0
                for batch in common.NativeReadFeeder([p1, p2], True, batch_units=args.batch, threads=threads,
                                                     pinned=args.pinned):
                    n += batch.count
                dt = time.perf_counter() - t0
                print('threads %2d pinned %d rep %d: %.3f s  %.1f M pairs/s  %.1f GB/s of text'
                      % (threads, args.pinned, rep, dt, n / dt / 1e6, 2 * n * 219 / dt / 1e9), flush=True)
    finally:
        shutil.rmtree(folder, ignore_errors=True)


if __name__ == '__main__':
    main()
