"""End-to-end rates around the hot path (numbers for DESIGN.md, not the bench metric):
  * FASTQ text -> batches (native reader alone),
  * FASTQ text -> TPM through infer.run (reader + H2D + GPU + outputs),
  * one resident-size batch handed over as HOST arrays (skm_mapper_map_batch: the
    PCIe-inclusive rate of the boundary).
    python3 scripts/e2e_ingest.py --genes 20000 --pairs 2000000 --cache /tmp/skm_idx.npz
"""
import argparse
import os
import pathlib
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekmer_amd import common, index_builder, infer, mapper, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--genes', type=int, default=20000)
    ap.add_argument('--pairs', type=int, default=2_000_000)
    ap.add_argument('--host-pairs', type=int, default=10_000_000)
    ap.add_argument('--read-len', type=int, default=100)
    ap.add_argument('--cache', default='')
    ap.add_argument('--jobs', type=int, default=1)
    args = ap.parse_args()
    ids, pool, tx_offsets = synth.transcriptome(1, args.genes)
    if args.cache and os.path.exists(args.cache):
        index = common.KMerIndex.load(args.cache)
    else:
        index = index_builder.build_pooled(ids, pool, tx_offsets)
        if args.cache:
            index.save(args.cache)
    index.device_handle(0)
    work = pathlib.Path(tempfile.mkdtemp(prefix='skm_e2e_'))
    index_path = pathlib.Path(args.cache) if args.cache else work / 'index.npz'
    if not args.cache:
        index.save(index_path)

    # ---- a resident-size batch handed over as host arrays
    bases, offsets = synth.reads(1, pool, tx_offsets, 0, args.host_pairs, args.read_len, True)
    result = mapper.MapResult(index)
    rm = mapper.ReadMapper(index, result)
    batch = common.ReadBatch(args.host_pairs, bases, offsets, True)
    for rep in range(3):
        result.reset()
        t0 = time.perf_counter()
        rm.map_batch(batch)
        dt = time.perf_counter() - t0
        print('host-buffer map_batch rep %d: %d pairs in %.1f ms = %.1f M pairs/s (%.2f GB of bases over PCIe)'
              % (rep, args.host_pairs, dt * 1e3, args.host_pairs / dt / 1e6, bases.size / 1e9), flush=True)
    del result, rm, batch

    # ---- FASTQ text
    r1, r2 = work / 'r1.fastq', work / 'r2.fastq'
    synth.write_fastq(bases, args.pairs, args.read_len, True, r1, r2)
    size = (r1.stat().st_size + r2.stat().st_size) / 1e9
    for rep in range(2):
        t0 = time.perf_counter()
        n = sum(b.count for b in common.NativeReadFeeder([r1, r2], True))
        dt = time.perf_counter() - t0
        print('native reader alone (sequential engine) rep %d: %d pairs in %.2f s = %.2f M pairs/s (%.2f GB/s of FASTQ text)'
              % (rep, n, dt, n / dt / 1e6, size / dt), flush=True)
    for rep in range(2):
        t0 = time.perf_counter()
        infer.run(index_path, work / ('out%d' % rep), [r1, r2], args.jobs, False, False, 0, False)
        dt = time.perf_counter() - t0
        print('infer.run rep %d (index load + FASTQ -> abundance.tsv, -j %d): %d pairs in %.2f s = %.2f M pairs/s'
              % (rep, args.jobs, args.pairs, dt, args.pairs / dt / 1e6), flush=True)
    # ---- the same run, phase by phase
    import datetime
    t = time.perf_counter()
    phases = []

    def lap(name):
        nonlocal t
        now = time.perf_counter()
        phases.append('%s %.2fs' % (name, now - t))
        t = now

    index2 = common.KMerIndex.load(index_path); lap('index load')
    index2.device_handle(0); lap('index upload')
    feeder = infer._feeder([r1, r2], True, None, None, False)
    map_result = mapper.map_reads(index2, feeder, job_count=args.jobs); lap('map_reads')
    summarized = map_result.summarize(); lap('summarize')
    tpm = infer.quantify(summarized); lap('quantify')
    (work / 'out_phases').mkdir()
    infer.output_results(work / 'out_phases', index2, datetime.datetime.utcnow(), summarized, tpm, [])
    lap('output_results')
    print('phases: ' + ', '.join(phases), flush=True)
    for p in (r1, r2):
        p.unlink()


if __name__ == '__main__':
    main()
