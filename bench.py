#!/usr/bin/env python3
"""Headline benchmark: paired reads/s mapped + quantified (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--config 1|3|4]

One "step" = one pass of the infer hot path over one resident batch:
pack -> map -> class counting -> effective lengths -> EM to the reference's
stopping rule -> TPM, against the synthetic ~190k-transcript index (the ENSEMBL
cDNA BASELINE.json names is not available offline; SURVEY.md 8(d) defines the
seeded stand-in).  Inputs are resident in HBM when the timed region starts.

  --config 1 (default)  BASELINE configs[1]: 10 M 2x100 bp pairs per GPU
  --config 3            BASELINE configs[3]: 50 M single-end 150 bp reads (-s)
  --config 4            BASELINE configs[4]: 20 M 2x100 bp pairs + `-b 100` (the step also draws
                        100 multinomial resamples of the class table and runs the EM on each)

For N > 1 (launched by torch.distributed.run, one rank per GPU) reads shard
across ranks with no collective while mapping and one RCCL all-reduce of
f64[T] per EM step (--config 4: the merged table on every rank, the replicates
shared out, no collective while they run); rank 0 prints ONE JSON line.  The
default run (N = 1, --config 1) also carries `e2e`: the same workload timed
from host arrays (PCIe inclusive) and from FASTQ text -- neither is `value` --
and `other_configs`: a few timed steps of configs[3] and configs[4] on the
same index.
"""
import argparse
import ctypes
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PMC_SUMMARY = os.path.join('profiles', 'r04_pmc_map.json')     # rocprofv3 --pmc passes of the map kernel
MAP_KERNEL_SOURCES = ('skm_map.hip', 'skm_device.h', 'skm_kernels.h')


def map_source_hash():
    """Digest of the map kernel's sources: scripts/pmc_finish.py stores it with the PMC summary, and
    `roofline.traffic` is only quoted from a summary whose digest is the tree's (a kernel change
    after the counter passes makes the quote null, not stale)."""
    import hashlib
    h = hashlib.sha256()
    for name in MAP_KERNEL_SOURCES:
        with open(os.path.join(ROOT, 'seekmer_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def table_digest(map_result):
    """sizes() and a digest of export() (class offsets, ids, counts, first-seen units, histogram)."""
    import hashlib
    h = hashlib.sha256()
    for a in map_result.export():
        h.update(np.ascontiguousarray(a).tobytes())
    return tuple(int(v) for v in map_result.sizes()), h.hexdigest()[:16]


def log(*a):
    print('[bench]', *a, file=sys.stderr, flush=True)


def cpu_baseline(index, pool, tx_offsets, seed, read_len, sample_units, paired, bootstraps=0, gpu_table=None,
                 parity=None):
    """The CPU oracle ("port" of the reference algorithm) on a bounded sample of the same
    workload: map + class counting + effective lengths + EM (+ `bootstraps` resamples, each a
    numpy multinomial draw and an EM from the main estimate, as seekmer/infer.py:108-111).
    Timed twice: one thread, and the map phase sharded over all the host cores this process may
    use (the oracle's C mapper releases the GIL; classes are then counted in shard order, EM on
    one core) -- `value` is the faster, the other is quoted in `sample`."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    from seekmer_amd import synth
    oindex = O.OracleIndex(index.kmers, index.contigs, index.sequences, index.targets,
                           lengths=np.diff(tx_offsets))
    bases, offsets = synth.reads(seed, pool, tx_offsets, 0, sample_units, read_len, paired)
    mates = 2 if paired else 1

    def quantify(results, fld):
        classes = O.Classes()
        for r in results:
            classes.update(r)
        class_map, class_count = classes.summarize()
        eff = O.effective_lengths(fld, oindex.lengths)
        tpm, iters = O.quantify(eff, class_map, class_count)
        boot_s = 0.0
        if bootstraps:
            rng = np.random.default_rng(seed)
            x0 = tpm / tpm.sum()
            t0 = time.perf_counter()
            for _ in range(bootstraps):
                draw = rng.multinomial(int(class_count.sum()), class_count / class_count.sum()).astype('f8')
                O.em(x0, eff, class_map, draw)
            boot_s = time.perf_counter() - t0
        return tpm, iters, boot_s

    t0 = time.perf_counter()
    fld = np.zeros(2000, dtype=np.int64)
    result = O.map_batch(oindex, bases, offsets, sample_units, paired, fld)
    t_map1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    tpm, iters, boot_s = quantify([result], fld)
    t_quant = time.perf_counter() - t0
    if gpu_table is not None:
        # the oracle as the checker (untimed): the HIP path's class table, histogram and TPM of the SAME
        # sample against the ones just computed -- a mismatch fails the run
        classes = O.Classes()
        classes.update(result)
        want = classes.export() + (fld,)
        got, got_unaligned, got_tpm, got_iters = gpu_table(bases, offsets, sample_units)
        same = (all(np.array_equal(g, w) for g, w in zip((got[0], got[1], got[2], got[4]), want))
                and got_unaligned == classes.unaligned and got_iters == iters)
        mask = tpm > 0
        rel = float((np.abs(got_tpm[mask] - tpm[mask]) / tpm[mask]).max()) if mask.any() else 0.0
        same = same and np.array_equal(got_tpm > 0, mask) and rel < 1e-4
        parity['cpu_sample_vs_hip'] = {
            'units': sample_units, 'classes': int(want[2].size), 'unaligned': int(classes.unaligned),
            'tables_identical': bool(same), 'em_iterations': int(iters), 'tpm_max_rel_diff': rel,
            'what': 'class offsets, ids, counts (first-seen order), fragment-length histogram and unaligned count of '
                    'the oracle == the HIP path\'s on the cpu_baseline sample, EM step count equal, TPM within 1e-4'}
        if not same:
            raise SystemExit('parity check failed: the HIP path and the oracle disagree on the cpu_baseline sample: %s'
                             % parity['cpu_sample_vs_hip'])
    del result

    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    per = (sample_units + cores - 1) // cores

    def shard(k):
        first, last = min(k * per, sample_units), min((k + 1) * per, sample_units)
        lo, hi = offsets[mates * first], offsets[mates * last]
        sub_offsets = (offsets[mates * first:mates * last + 1] - lo).copy()
        sub_bases = np.concatenate([bases[lo:hi], np.zeros(1, np.uint8)])
        sub_fld = np.zeros(2000, dtype=np.int64)
        return O.map_batch(oindex, sub_bases, sub_offsets, last - first, paired, sub_fld), sub_fld

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(shard, range(cores)))
    t_mapn = time.perf_counter() - t0
    fld_n = np.sum([p[1] for p in parts], axis=0)
    assert np.array_equal(fld_n, fld)           # sharding does not change the integer results
    single = sample_units / (t_map1 + t_quant)
    multi = sample_units / (t_mapn + t_quant)   # (classes + EM timed once: the same work either way)
    unit = 'pairs/s' if paired else 'reads/s'
    out = {
        # the faster of the two runs, with the threads it used (the oracle restates the reference's
        # per-read malloc'd lists, which limits its thread scaling)
        'value': max(single, multi), 'unit': unit, 'cores': cores if multi > single else 1,
        'kind': 'port', 'single_thread_value': single, 'all_core_value': multi, 'host_cores': cores,
        'sample': 'the C port of the reference algorithm (oracle/) on THIS host -- not the reference\'s Cython build, whose '
                  'ratio to the port could not be calibrated here (the reference is not importable in this image); the '
                  'only reference-side figure is SURVEY.md 6: 132.9 k pairs/s at -j1 on an 8-vCPU Xeon.  '
                  '%d %s of the same read set; 1 thread: oracle map %.2fs (%.0f %s) + classes/EM%s %.2fs '
                  '(%d EM steps); %d threads: map %.2fs (%.0f %s) + the same classes/EM'
                  % (sample_units, 'pairs' if paired else 'reads', t_map1, sample_units / t_map1, unit,
                     ' incl. %d bootstraps' % bootstraps if bootstraps else '', t_quant, iters, cores,
                     t_mapn, sample_units / t_mapn, unit),
    }
    if bootstraps:
        out['bootstraps_per_s'] = bootstraps / boot_s
    return out


def e2e_legs(args, hip, _native, index, device, bases, offsets, n_units, n_tx, passes=3):
    """configs[1] once more from outside the GPU (N = 1 only), none of it `value`:
    (a) `pcie_inclusive`: the reads as the mapper takes them from a host -- 2-bit code words, 32 bytes
        per 100-base read -- in page-locked host arrays, pushed in `--e2e-batches` pieces per stream
        (skm_mapper_push_packed: the copy of a piece runs under the kernels of the ones before), then
        the resident quantification; `ascii` beside it: the same reads as ASCII bases (100 bytes per
        read, packed on the device) through skm_mapper_map_batch_uniform_async;
    (b) `fastq_inclusive`: the same reads as FASTQ text on the host's RAM disk, parsed to code words
        in ONE pass by the native reader's workers and drained into the mapper natively
        (skm_mapper_map_packed_source over skm_fastq_packed_next), then the quantification;
        `two_pass_ascii` beside it: last round's path (newline index, ASCII slabs).
    Best of `passes`; the first pass (cold file mapping, cold pools) is reported too."""
    from seekmer_amd import common, infer, mapper, synth
    out = {}
    read_len, pieces = args.read_len, max(1, args.e2e_batches)
    _native.check(hip.skm_pinned_set_device(device))
    result = mapper.MapResult(index, device=device)
    rm = mapper.ReadMapper(index, result)
    host = _native.host()

    def pinned_copy(array):
        raw = hip.skm_pinned_alloc(array.nbytes)
        if not raw:
            raise RuntimeError('cannot page-lock %d bytes of host memory' % array.nbytes)
        view = np.ctypeslib.as_array(ctypes.cast(raw, ctypes.POINTER(ctypes.c_uint8)), (array.nbytes,))
        view[:] = array.reshape(-1).view(np.uint8)
        return raw, view.view(array.dtype).reshape(array.shape)

    # ---- (a) host arrays -> TPM
    cut = [n_units * k // pieces for k in range(pieces + 1)]
    t0 = time.perf_counter()
    lengths = np.full(n_units, read_len, dtype=np.uint32)
    streams, held = [], []
    flat = bases[:2 * n_units * read_len].reshape(n_units, 2, read_len)
    for s in range(2):
        mate = np.ascontiguousarray(flat[:, s, :]).reshape(-1)
        piece = common.PackedReads.from_ascii(np.concatenate([mate, np.zeros(1, np.uint8)]),
                                              np.arange(n_units + 1, dtype=np.int64) * read_len, stream=s, paired=True)
        assert piece.raw.n_exceptions == 0
        raw, codes = pinned_copy(np.array(piece.codes))
        held.append(raw)
        streams.append(codes)
    log('packed the reads for the host-array leg in %.1fs' % (time.perf_counter() - t0))
    packed_bytes = sum(c.nbytes for c in streams)

    def from_host_packed():
        result.reset()
        t0 = time.perf_counter()
        for k in range(pieces):
            for s in range(2):
                rm.push_packed(common.PackedReads.from_arrays(s, cut[k], streams[s][cut[k]:cut[k + 1]],
                                                              lengths[cut[k]:cut[k + 1]], paired=True))
        result.sync()
        t_map = time.perf_counter() - t0
        infer.quantify_resident(result)
        return t_map, time.perf_counter() - t0

    runs = [from_host_packed() for _ in range(passes + 1)]
    best = min(runs, key=lambda t: t[1])
    classes_host = result.sizes()
    out['pcie_inclusive'] = {
        'value': n_units / best[1], 'unit': 'pairs/s', 'through_mapping': n_units / best[0],
        'first_pass': n_units / runs[0][1],
        'host_bytes': int(packed_bytes), 'bytes_per_pair': packed_bytes / n_units,
        'GBps_over_pcie': packed_bytes / best[0] / 1e9,
        'how': '%d pairs as 2-bit code words (%d bytes per pair) in page-locked host arrays, %d pieces per stream '
               'through skm_mapper_push_packed, then skm_quant_infer; best of %d passes'
               % (n_units, packed_bytes // n_units, pieces, passes)}
    for raw in held:
        hip.skm_pinned_free(raw)
    del streams
    # the same from ASCII bases (what round 2 measured)
    n_bytes = bases.size
    p_bases = hip.skm_pinned_alloc(n_bytes)
    if not p_bases:
        raise RuntimeError('cannot page-lock %d bytes of host memory' % n_bytes)
    h_bases = np.ctypeslib.as_array(ctypes.cast(p_bases, ctypes.POINTER(ctypes.c_uint8)), (n_bytes,))
    h_bases[:] = bases

    def from_host_ascii():
        result.reset()
        t0 = time.perf_counter()
        for k in range(pieces):
            sub = common.ReadBatch(cut[k + 1] - cut[k], h_bases, offsets[2 * cut[k]:2 * cut[k + 1] + 1], True,
                                   first_unit=cut[k], uniform_len=read_len)
            rm.map_batch_async(sub)
        result.sync()
        t_map = time.perf_counter() - t0
        infer.quantify_resident(result)
        return t_map, time.perf_counter() - t0

    best = min((from_host_ascii() for _ in range(passes)), key=lambda t: t[1])
    assert result.sizes() == classes_host
    out['pcie_inclusive']['ascii'] = {'value': n_units / best[1], 'through_mapping': n_units / best[0],
                                      'host_bytes': int(n_bytes), 'GBps_over_pcie': n_bytes / best[0] / 1e9}
    hip.skm_pinned_free(p_bases)
    # ---- (b) FASTQ text -> TPM
    ram = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
    need = 2 * n_units * (2 * read_len + 19)
    if shutil.disk_usage(ram).free < need * 1.2:
        ram = tempfile.gettempdir()
    if shutil.disk_usage(ram).free < need * 1.2:
        out['fastq_inclusive'] = None
        return out
    folder = tempfile.mkdtemp(prefix='skm_bench_', dir=ram)
    try:
        p1, p2 = os.path.join(folder, 'r_1.fastq'), os.path.join(folder, 'r_2.fastq')
        t0 = time.perf_counter()
        synth.write_fastq(bases, n_units, read_len, True, p1, p2)
        log('wrote %.1f GB of FASTQ text to %s in %.1fs' % (need / 1e9, folder, time.perf_counter() - t0))
        cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        threads = max(1, min(args.parse_threads, cores))
        chunk = args.e2e_chunk_mb << 20

        def feeder():
            return common.PackedReadFeeder([p1, p2], True, threads=threads, chunk_bytes=chunk, pinned=True)

        def parse_only():
            t0 = time.perf_counter()
            count = sum(piece.n_reads for piece in feeder())
            assert count == 2 * n_units
            return time.perf_counter() - t0

        def from_fastq():
            result.reset()
            t0 = time.perf_counter()
            source = feeder()
            rm(source)                       # the reference's mapping loop: every piece handed over, then sync
            t_map = time.perf_counter() - t0
            infer.quantify_resident(result)
            return t_map, time.perf_counter() - t0, source.stats

        cold = from_fastq()                  # first pass: the files are mapped and the pools filled here
        assert result.sizes() == classes_host                # same classes as from the host arrays
        _native.check_host(host.skm_fastq_cache_bytes(2 * need), 'skm_fastq_cache_bytes')   # later passes re-use the mappings
        from_fastq()
        t_parse = min(parse_only() for _ in range(2))
        runs = [from_fastq() for _ in range(passes)]
        best = min(runs, key=lambda t: t[1])
        assert result.sizes() == classes_host
        out['fastq_inclusive'] = {
            'value': n_units / best[1], 'unit': 'pairs/s', 'through_mapping': n_units / best[0],
            'first_pass': n_units / cold[1],
            'parse_only': n_units / t_parse, 'text_GBps': need / best[0] / 1e9, 'parse_threads': threads,
            'chunk_mb': args.e2e_chunk_mb, 'reader': best[2],
            'how': 'two FASTQ files (%d bytes per record, RAM disk) -> PackedReadFeeder(threads=%d, pinned, one pass '
                   'over the text, 2-bit code words) -> skm_mapper_map_packed_source -> skm_quant_infer; best of %d '
                   'passes with the files\' mappings and the page-locked pieces kept between passes, `first_pass` '
                   'without either' % (2 * read_len + 19, threads, passes)}

        # what ONE `seekmer infer` run sees: the same pass in a process of its own (nothing mapped,
        # pooled or page-locked yet; the index upload is set-up, as it is for `first_pass`)
        try:
            index_path = args.index_cache if args.index_cache and os.path.exists(args.index_cache) \
                else os.path.join(folder, 'index.npz')
            if not os.path.exists(index_path):
                index.save(index_path)
            cmd = [sys.executable, os.path.abspath(__file__), '--cold-child', index_path, p1, p2,
                   '--parse-threads', str(threads), '--e2e-chunk-mb', str(args.e2e_chunk_mb)]
            import subprocess

            def children(env_extra):
                env = dict(os.environ)
                env.update(env_extra)
                runs = []
                for _ in range(2):
                    done = subprocess.run(cmd, stdout=subprocess.PIPE, check=True, timeout=600, env=env)
                    runs.append(json.loads(done.stdout.decode().strip().splitlines()[-1]))
                assert all(tuple(c['sizes']) == tuple(classes_host) for c in runs)
                return runs

            colds = children({})
            bare = children({'SKM_COLD_NO_PREFAULT': '1'})
            best_cold, best_bare = (min(runs, key=lambda c: c['total_s']) for runs in (colds, bare))
            out['fastq_inclusive']['cold_process'] = {
                'value': n_units / best_cold['total_s'], 'unit': 'pairs/s',
                'through_mapping': n_units / best_cold['map_s'],
                'runs': [n_units / c['total_s'] for c in colds], 'setup_s': best_cold['setup_s'],
                'without_prefault': {'value': n_units / best_bare['total_s'], 'through_mapping': n_units / best_bare['map_s'],
                                     'runs': [n_units / c['total_s'] for c in bare]},
                'how': 'the fastq_inclusive pass as the ONLY pass of a fresh process (python bench.py --cold-child), set up '
                       'as seekmer_amd.infer.run sets a run up: the page-locked arena reserved and the page tables of the '
                       'FASTQ text set up by helper threads WHILE the index is loaded and uploaded (all of that is set-up), '
                       'then FASTQ text -> TPM once, timed; the faster of two such processes.  `without_prefault`: the same '
                       'with the text\'s page tables left to the reader (SKM_COLD_NO_PREFAULT=1)'}
        except Exception as e:                      # (the leg is a report, not the metric)
            out['fastq_inclusive']['cold_process'] = {'error': repr(e)}

        # last round's path beside it: newline index + ASCII slabs
        fastq_pieces = max(1, args.e2e_fastq_batches)
        batch_units = (n_units + fastq_pieces - 1) // fastq_pieces

        def from_fastq_ascii():
            result.reset()
            t0 = time.perf_counter()
            source = common.NativeReadFeeder([p1, p2], True, batch_units=batch_units, threads=min(threads, 12), pinned=True)
            rm(source)
            t_map = time.perf_counter() - t0
            infer.quantify_resident(result)
            return t_map, time.perf_counter() - t0

        best = min((from_fastq_ascii() for _ in range(passes)), key=lambda t: t[1])
        assert result.sizes() == classes_host
        out['fastq_inclusive']['two_pass_ascii'] = {'value': n_units / best[1], 'through_mapping': n_units / best[0],
                                                    'batches': fastq_pieces}
    finally:
        host.skm_fastq_cache_bytes(0)
        host.skm_fastq_wait_unmapped()       # (the text's mappings go in the background: not under the next leg)
        shutil.rmtree(folder, ignore_errors=True)
    return out


def cold_child(args):
    """`e2e.fastq_inclusive.cold_process`: what a one-shot `seekmer infer` run does after its index is
    in HBM -- ONE pass from FASTQ text to TPM in a process that has mapped, pooled and page-locked
    nothing yet."""
    from seekmer_amd import _native, common, infer, mapper
    index_path, r1, r2 = args.cold_child
    t_start = time.perf_counter()
    _native.check(_native.hip().skm_pinned_set_device(0))      # (as infer.run: first thing, before the index is loaded)
    ahead = common.Prefault([r1, r2] if os.environ.get('SKM_COLD_NO_PREFAULT') != '1' else [], threads=4)   # (as infer.run)
    index = common.KMerIndex.load(index_path)
    index.device_handle(0)
    result = mapper.MapResult(index)
    rm = mapper.ReadMapper(index, result)
    _native.check(_native.hip().skm_device_synchronize(0))        # (scripts/cold_timeline.py: start of the timed region)
    t0 = time.perf_counter()
    feeder = common.PackedReadFeeder([r1, r2], True, threads=args.parse_threads,
                                     chunk_bytes=args.e2e_chunk_mb << 20, pinned=True)
    if os.environ.get('SKM_COLD_PARSE_ONLY') == '1':          # (tuning aid: what the reader alone costs a fresh process)
        n = sum(piece.n_reads for piece in feeder)
        print(json.dumps({'setup_s': t0 - t_start, 'parse_s': time.perf_counter() - t0, 'reads': n}), flush=True)
        ahead.finish()
        return
    rm(feeder)
    t_map = time.perf_counter() - t0
    infer.quantify_resident(result)
    total = time.perf_counter() - t0
    _native.check(_native.hip().skm_device_synchronize(0))        # (its end)
    ahead.finish()
    print(json.dumps({'setup_s': t0 - t_start, 'map_s': t_map, 'total_s': total,
                      'sizes': [int(v) for v in result.sizes()]}), flush=True)


SHAPES = {1: (10_000_000, 100, True, 0), 3: (50_000_000, 150, False, 0), 4: (20_000_000, 100, True, 100),
          # BASELINE.json north_star's target sentence: 50 M 2x100 pairs, ~200k-transcript index, one GPU
          'north_star': (50_000_000, 100, True, 0)}


def measure(args, ctx, config, steps, warmup, n_units=0, read_len=0, bootstraps=-1, with_cpu=True, with_e2e=False):
    """One configuration on the index that is already in HBM: `warmup` untimed steps, EXACTLY `steps`
    timed ones between barrier + device synchronisation, the roofline of the map kernel from its own
    HIP-event times, and (rank 0) the JSON line's dict."""
    from seekmer_amd import infer, mapper, parallel, synth
    hip, _native, index, device = ctx['hip'], ctx['native'], ctx['index'], ctx['device']
    rank, world, dist, comm = ctx['rank'], ctx['world'], ctx['dist'], ctx['comm']
    pool, tx_offsets, n_tx = ctx['pool'], ctx['tx_offsets'], ctx['n_tx']
    shape = SHAPES[config]
    n_units = n_units or shape[0]
    read_len = read_len or shape[1]
    paired = shape[2]
    bootstraps = shape[3] if bootstraps < 0 else bootstraps
    unit_name = 'pairs' if paired else 'reads'
    ranks = parallel.Ranks(rank, world, ctx['local_rank'], dist)

    bases, offsets = synth.reads(args.seed, pool, tx_offsets, rank * n_units, n_units, read_len, paired)
    d_bases, d_offsets = ctypes.c_void_p(), ctypes.c_void_p()
    _native.check(hip.skm_device_malloc(device, bases.size, ctypes.byref(d_bases)))
    _native.check(hip.skm_device_malloc(device, offsets.size * 8, ctypes.byref(d_offsets)))
    _native.check(hip.skm_device_upload(device, d_bases, bases.ctypes.data, bases.size))
    _native.check(hip.skm_device_upload(device, d_offsets, offsets.ctypes.data, offsets.size * 8))
    result = mapper.MapResult(index, device=device)

    state = {'boot_s': 0.0, 'boot_iters': 0}
    stage = {}
    profile_stages = os.environ.get('SKM_BENCH_PROFILE') in ('1', '2')   # 2: host wall per call, no syncs
    profile_sync = os.environ.get('SKM_BENCH_PROFILE') == '1'

    def mark(name, t0):
        if profile_stages:
            if profile_sync:
                _native.check(hip.skm_device_synchronize(device))
            stage[name] = stage.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    def quantify_ranks(map_result):
        # fragment lengths (RCCL all-reduce over the ranks) -> effective lengths -> start vector -> EM
        # -> TPM: one native call on the resident class table
        out = infer.quantify_resident(map_result, comm=comm if comm else None, return_iters=True,
                                      return_effective_lengths=bool(bootstraps))    # (the `-b N` loop starts from them)
        state['iters'] = out[1]
        state['eff'] = out[2] if bootstraps else None
        return out[0]

    def step():
        t = time.perf_counter()
        result.reset()
        t = mark('reset', t)
        result.map_resident(d_bases, d_offsets, n_units, paired, read_len)
        t = mark('map_batch', t)
        before = result.timing()
        tpm = quantify_ranks(result)
        t = mark('quantify', t)
        after = result.timing()
        state['em'] = {'em_ns': after['em_ns'] - before['em_ns'],
                       'iterations': after['em_iterations'] - before['em_iterations']}
        state['tpm'] = tpm
        if not bootstraps:
            return
        # `-b N` (seekmer/infer.py:79-82): resample the table of the whole sample + EM from the main estimate
        t_b = time.perf_counter()
        if world == 1:
            quant = infer._QuantHandle.from_map_result(result, n_tx)
            if comm:                       # (1-rank RCCL rehearsal: every replicate's steps all-reduced)
                _native.check(hip.skm_quant_set_comm(quant.handle, comm))
            x0 = tpm / tpm.sum()
            out, _, it = quant.bootstrap(bootstraps, args.seed, x0, state['eff'], tpm=True)   # TPM vectors, as run() keeps them
            state['boot_tpm'] = out
            quant.close()
            state['boot_iters'] += int(it.sum())
        else:
            # several ranks: the tables go to rank 0 and are merged on its GPU, the merged table to
            # every rank, rank r runs replicates r, r + G, ... (no collective), rank 0 gathers them
            tables = ranks.gather_arrays_to_root(parallel.rank_table(result) if rank else {})
            summarized = None
            if rank == 0:
                parallel.merge_into(result, tables[1:])
                summarized = result.summarize()
            state['boot_tpm'] = infer.bootstrap_ranks(summarized, tpm, bootstraps, ranks, seed=args.seed, device=device)
        state['boot_s'] += time.perf_counter() - t_b
        mark('bootstrap', t)

    def barrier():
        if dist is not None:
            dist.barrier()
        _native.check(hip.skm_device_synchronize(device))

    for _ in range(warmup):
        step()
    state['boot_s'], state['boot_iters'] = 0.0, 0
    # (handles of earlier legs that only the cycle collector can reach are destroyed NOW, not by a
    # collection that happens to start inside the timed steps: their device memory goes back with
    # synchronous hipFree calls -- seen once as 90 ms inside configs[3]'s EM phase)
    import gc
    gc.collect()
    t_before = result.timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    em_steps_check = None
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        # every rank must have run the same number of EM steps (each holds one all-reduce): a
        # mismatch would mean the redundantly judged stopping rule diverged between ranks
        lo = torch.tensor([state['iters']], dtype=torch.int64)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        em_steps_check = {'min_over_ranks': int(lo[0]), 'max_over_ranks': int(hi[0])}
        if int(lo[0]) != int(hi[0]):
            raise SystemExit('EM step counts differ between ranks: %s' % em_steps_check)
    t_after = result.timing()
    if profile_stages and rank == 0:
        log('stage ms per step: ' + ', '.join('%s %.2f' % (k, 1e3 * v / (steps + warmup))
                                              for k, v in stage.items()))

    # ---------------- roofline of the dominant kernel
    launches = t_after['batches'] - t_before['batches']
    map_ns = (t_after['map_ns'] - t_before['map_ns']) / max(launches, 1)
    pack_ns = (t_after['pack_ns'] - t_before['pack_ns']) / max(launches, 1)
    class_ns = (t_after['class_ns'] - t_before['class_ns']) / max(launches, 1)
    parity = {}
    production = table_digest(result)      # the last timed step's table (bucket layout, production kernel)
    result.set_stats(True)                 # counting build, untimed: access counts per launch
    result.reset()
    result.map_resident(d_bases, d_offsets, n_units, paired, read_len)
    st = result.access_stats()
    result.set_stats(False)
    counting = table_digest(result)        # the same units through the reference's table layout
    parity['counting_vs_production'] = {
        'units': n_units, 'sizes': list(production[0]), 'digest': production[1],
        'identical': production == counting,
        'what': 'sizes() and sha256 of export() of the counting build\'s launch (reference table layout, the '
                'launch whose access counters are the roofline\'s algorithmic bytes) == the production launch\'s'}
    if bootstraps and world > 1:
        # (rank 0's table then holds the other ranks' merged in: not the launch's own any more)
        parity['counting_vs_production']['identical'] = None
    elif production != counting:
        raise SystemExit('parity check failed: counting build %s != production build %s' % (counting, production))
    algorithmic = (st['read_bases'] + 16 * st['slots'] + 48 * st['contig_reads']
                   + 8 * (st['targets_copied'] + st['targets_merged']) + 8 * st['seq_fetches']
                   + 4 * st['tuple_ids'])
    achieved = algorithmic / map_ns if map_ns else 0.0          # bytes/ns = GB/s
    sizes = result.sizes()
    em_bytes = int(sizes[1] * 28 + sizes[0] * 12 + n_tx * 40)     # B_em per EM step
    # HBM traffic of the map launch: NOT measured by this run (a rocprofv3 --pmc pass is not
    # something a benchmark can do to itself) -- quoted from the PMC passes of the same build over
    # the same workload (FETCH_SIZE + WRITE_SIZE, separate --pmc runs, summary committed in
    # profiles/), and only when this run IS that workload
    traffic, traffic_source, miss_rate = None, None, None
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            pmc = json.load(f)
        if pmc.get('source_hash') != map_source_hash():
            traffic_source = 'null: %s was collected from other kernel sources than this tree\'s' % PMC_SUMMARY
        elif config == 1 and args.genes == 20000 and n_units == 10_000_000 and read_len == 100:
            traffic = pmc['derived']['hbm_traffic_bytes']
            traffic_source = 'quoted from %s (rocprofv3 --pmc passes of this build on this workload), ' \
                             'not measured by this run; below the algorithmic bytes where the product ' \
                             'proves k-mers absent without the bucket reads the reference\'s probe makes ' \
                             '(signatures by minimizer, DESIGN.md 4)' % PMC_SUMMARY
            miss_rate = pmc['per_launch']['TCC_MISS_sum'] / (map_ns * 1e-9)
    except (OSError, KeyError, ValueError):
        pass

    line = None
    if rank == 0:
        total_units = world * n_units * steps
        line = {
            'metric': 'paired reads/sec mapped+quantified' if paired else 'reads/sec mapped+quantified (single-end)',
            'value': total_units / elapsed,
            'unit': '%s/s' % unit_name,
            'n_gpus': world,
            'steps': steps,
            'warmup': warmup,
            'ms_per_step': 1000.0 * elapsed / steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'u64+f64',
            'data': 'synthetic',
            'config': {
                'workload': '%s: synthetic ~190k-tx index (stand-in for ENSEMBL GRCh38 cDNA), '
                            '%d %s synthetic %s per GPU, map+classes+EM to the reference stop rule%s'
                            % ('configs[%d]' % config if isinstance(config, int) else config, n_units, ('2x%dbp' if paired else '%dbp single-end') % read_len,
                               unit_name, ' + %d bootstraps (-b)' % bootstraps if bootstraps else ''),
                'transcripts': n_tx, 'kmer_slots': int(index.kmers.size),
                'units_per_gpu': n_units, 'read_len': read_len, 'paired': paired,
                'classes': sizes[0], 'class_map_rows': sizes[1],
                'em_iterations': int(state['iters']),
                'em_iters_per_s': state['em']['iterations'] / (state['em']['em_ns'] * 1e-9),
                'em_bytes_per_step': em_bytes,
                'em_algorithmic_GBps': em_bytes * state['em']['iterations'] / max(state['em']['em_ns'], 1.0),
                'parallelism': 'reads sharded x%d, RCCL all-reduce f64[T] per EM step%s'
                               % (world, '; bootstraps shared out over the ranks' if bootstraps and world > 1 else ''),
                'rccl_ranks': ctx['rccl_ranks'],
                'phase_ms': {'pack': pack_ns * 1e-6, 'map': map_ns * 1e-6, 'classes': class_ns * 1e-6,
                             'em': state['em']['em_ns'] * 1e-6},
            },
            'roofline': {
                'kernel': 'map_units_kernel', 'bound': 'hbm',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'traffic_source': traffic_source,
                'algorithmic_bytes_per_launch': algorithmic,
                # the probes are dependent random 16-B reads: the relevant ceiling is the chip's
                # random-gather rate (52 G/s over a 2 GiB table, profiles/r01_gather_ceiling.log)
                'l2_miss_rate_vs_random_gather_ceiling': (miss_rate / 52.3e9) if miss_rate else None,
                'bytes_per_unit': algorithmic / n_units,
                'launch_ms': map_ns * 1e-6,
            },
        }
        if em_steps_check:
            line['config']['em_steps_over_ranks'] = em_steps_check
        if bootstraps:
            line['config']['bootstraps'] = bootstraps
            line['config']['bootstraps_per_s'] = bootstraps * steps / state['boot_s']
            if world == 1:
                line['config']['bootstrap_em_steps'] = state['boot_iters'] // steps
            line['config']['phase_ms']['bootstraps'] = 1e3 * state['boot_s'] / steps
        if with_cpu and world == 1:
            def gpu_table(sample_bases, sample_offsets, sample_units):
                from seekmer_amd import common
                sample = mapper.MapResult(index, device=device)
                mapper.ReadMapper(index, sample).map_batch(
                    common.ReadBatch(sample_units, sample_bases, sample_offsets, paired))
                sample_tpm, sample_iters = infer.quantify_resident(sample, return_iters=True)
                return sample.export(), sample.sizes()[2], sample_tpm, sample_iters

            line['cpu_baseline'] = cpu_baseline(index, pool, tx_offsets, args.seed, read_len,
                                                min(args.cpu_sample, n_units), paired,
                                                bootstraps=min(bootstraps, 3), gpu_table=gpu_table, parity=parity)
        line['parity_checked'] = parity
        if with_e2e and world == 1:
            del result
            _native.check(hip.skm_device_free(device, d_bases))
            _native.check(hip.skm_device_free(device, d_offsets))
            d_bases = d_offsets = None
            line['e2e'] = e2e_legs(args, hip, _native, index, device, bases, offsets, n_units, n_tx)
    if d_bases:
        del result
        _native.check(hip.skm_device_free(device, d_bases))
        _native.check(hip.skm_device_free(device, d_offsets))
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', type=int, default=1, choices=(1, 3, 4), help='BASELINE.json configs[] index')
    ap.add_argument('--genes', type=int, default=20000, help='synthetic genes (20000 -> ~190k tx)')
    ap.add_argument('--pairs', type=int, default=0, help='units per GPU per step (default: the config\'s size)')
    ap.add_argument('--read-len', type=int, default=0)
    ap.add_argument('--bootstraps', type=int, default=-1, help='-b N inside the step (default: 100 for --config 4)')
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--cpu-sample', type=int, default=1_000_000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the short legs of configs[3] and configs[4] that follow a default run')
    ap.add_argument('--other-steps', type=int, default=3)
    ap.add_argument('--e2e-batches', type=int, default=10, help='pieces per stream of the host-array leg')
    ap.add_argument('--e2e-fastq-batches', type=int, default=40,
                    help='batches of the two-pass ASCII comparison leg')
    ap.add_argument('--parse-threads', type=int, default=14)
    ap.add_argument('--e2e-chunk-mb', type=int, default=8, help='text range one parser thread takes at a time')
    ap.add_argument('--index-cache', default='', help='load/save the index here instead of building it')
    ap.add_argument('--cold-child', nargs=3, metavar=('INDEX', 'R1', 'R2'), default=None,
                    help='(internal) one FASTQ -> TPM pass in this fresh process; prints its times as JSON')
    args = ap.parse_args()
    if args.cold_child:
        return cold_child(args)
    args.read_len = args.read_len or SHAPES[args.config][1]

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            log('--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`' % (args.gpus, args.gpus))
            sys.exit(2)
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)   # rendezvous / barrier / host-side exchanges

    from seekmer_amd import _native, index_builder, parallel, synth
    hip = _native.hip()
    device = local_rank
    if os.environ.get('SKM_BENCH_ONE_DEVICE') == '1':      # rehearsal of the N > 1 path on a 1-GPU box
        device = 0
    if _native.device_count() <= device:
        raise SystemExit('no GPU %d visible: the benchmark has no CPU path' % device)

    # ---------------- setup (untimed): transcriptome, index -> HBM, communicator
    t0 = time.perf_counter()
    ids, pool, tx_offsets = synth.transcriptome(args.seed, args.genes)
    # one rank builds the index, the others map the container it wrote (2.2 GB, memory-mapped)
    index = parallel.shared_index(lambda: index_builder.build_pooled(ids, pool, tx_offsets), rank, world,
                                  barrier=(dist.barrier if dist is not None else None),
                                  cache=args.index_cache or None)
    n_tx = len(ids)
    log('rank %d: %d transcripts, %d k-mer slots, index ready in %.1fs'
        % (rank, n_tx, index.kmers.size, time.perf_counter() - t0))
    index.device_handle(device)
    comm_id = None
    force_comm = world == 1 and os.environ.get('SKM_FORCE_COMM') == '1'   # 1-rank RCCL rehearsal
    if force_comm:
        raw = ctypes.create_string_buffer(128)
        _native.check(hip.skm_comm_unique_id(raw))
        comm_id = raw.raw
    if world > 1:
        comm_id = parallel.broadcast_comm_id(dist, rank)
    comm = ctypes.c_void_p()
    if comm_id is not None:                # one communicator per process, reused by every step
        comm = parallel.create_comm(device, comm_id, rank, world)     # (RCCL's banner goes to stderr)
    rccl_ranks = 0
    if comm:
        n = ctypes.c_int(0)
        _native.check(hip.skm_comm_count(comm, ctypes.byref(n)))
        rccl_ranks = n.value
    ctx = {'hip': hip, 'native': _native, 'index': index, 'device': device, 'rank': rank, 'world': world,
           'local_rank': local_rank, 'dist': dist, 'comm': comm, 'rccl_ranks': rccl_ranks, 'pool': pool,
           'tx_offsets': tx_offsets, 'n_tx': n_tx}

    default_run = world == 1 and args.config == 1 and not args.pairs
    line = measure(args, ctx, args.config, args.steps, args.warmup, n_units=args.pairs, read_len=args.read_len,
                   bootstraps=args.bootstraps, with_cpu=not args.no_cpu_baseline,
                   with_e2e=default_run and not args.no_e2e)
    if default_run and not args.no_other_configs:
        # the other single-GPU configurations of BASELINE.json, a few steps each on the same index
        # (parity at these sizes: tests/test_gpu_parity.py::test_baseline_config{3,4}_properties)
        line['other_configs'] = {}
        for other in (3, 4, 'north_star'):
            t0 = time.perf_counter()
            sub = measure(args, ctx, other, 2 if other == 'north_star' else args.other_steps, 1, with_cpu=False)
            log('%s: %.1f M %s, %.1f ms per step (%.0fs with set-up)'
                % (other, sub['value'] / 1e6, sub['unit'], sub['ms_per_step'], time.perf_counter() - t0))
            keep = {k: sub[k] for k in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step')}
            keep['workload'] = sub['config']['workload']
            keep['phase_ms'] = sub['config']['phase_ms']
            for k in ('classes', 'em_iterations', 'em_iters_per_s', 'bootstraps', 'bootstraps_per_s'):
                if k in sub['config']:
                    keep[k] = sub['config'][k]
            keep['roofline'] = {k: sub['roofline'][k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac',
                                                                'algorithmic_bytes_per_launch', 'launch_ms')}
            keep['parity_checked'] = sub['parity_checked']
            line['other_configs']['configs[%d]' % other if isinstance(other, int)
                                  else 'north_star: 50 M 2x100 pairs'] = keep
    if rank == 0:
        print(json.dumps(line), flush=True)
    if comm:
        _native.check(hip.skm_comm_destroy(comm))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
