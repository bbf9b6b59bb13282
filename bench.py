#!/usr/bin/env python3
"""Headline benchmark: paired reads/s mapped + quantified (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the infer hot path over one resident batch:
pack -> map -> class counting -> effective lengths -> EM to the reference's
stopping rule -> TPM, for `--pairs` synthetic 2x100 read pairs per GPU against
the synthetic ~190k-transcript index (BASELINE.json configs[1]; the ENSEMBL
cDNA named there is not available offline, SURVEY.md 8(d) defines the seeded
stand-in).  Inputs are resident in HBM when the timed region starts.  For
N > 1 (launched by torch.distributed.run, one rank per GPU) reads shard across
ranks with no collective while mapping and one RCCL all-reduce of f64[T] per
EM step; rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print('[bench]', *a, file=sys.stderr, flush=True)


def cpu_baseline(index, pool, tx_offsets, seed, read_len, sample_units):
    """The CPU oracle ("port" of the reference algorithm) on a bounded sample of the same
    workload: map + class counting + effective lengths + EM.  Timed twice: one thread, and
    the map phase sharded over all the host cores this process may use (the oracle's C
    mapper releases the GIL; classes are then counted in shard order, EM on one core) --
    `value` is the all-core figure, the single-thread one is quoted in `sample`."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    from seekmer_amd import synth
    oindex = O.OracleIndex(index.kmers, index.contigs, index.sequences, index.targets,
                           lengths=np.diff(tx_offsets))
    bases, offsets = synth.reads(seed, pool, tx_offsets, 0, sample_units, read_len, True)

    def quantify(results, fld):
        classes = O.Classes()
        for r in results:
            classes.update(r)
        class_map, class_count = classes.summarize()
        eff = O.effective_lengths(fld, oindex.lengths)
        return O.quantify(eff, class_map, class_count)

    t0 = time.perf_counter()
    fld = np.zeros(2000, dtype=np.int64)
    result = O.map_batch(oindex, bases, offsets, sample_units, True, fld)
    t_map1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    tpm, iters = quantify([result], fld)
    t_quant = time.perf_counter() - t0
    del result

    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    per = (sample_units + cores - 1) // cores

    def shard(k):
        first, last = min(k * per, sample_units), min((k + 1) * per, sample_units)
        lo, hi = offsets[2 * first], offsets[2 * last]
        sub_offsets = (offsets[2 * first:2 * last + 1] - lo).copy()
        sub_bases = np.concatenate([bases[lo:hi], np.zeros(1, np.uint8)])
        sub_fld = np.zeros(2000, dtype=np.int64)
        return O.map_batch(oindex, sub_bases, sub_offsets, last - first, True, sub_fld), sub_fld

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(shard, range(cores)))
    t_mapn = time.perf_counter() - t0
    fld_n = np.sum([p[1] for p in parts], axis=0)
    assert np.array_equal(fld_n, fld)           # sharding does not change the integer results
    t0 = time.perf_counter()
    quantify([p[0] for p in parts], fld_n)
    t_quantn = time.perf_counter() - t0
    single = sample_units / (t_map1 + t_quant)
    multi = sample_units / (t_mapn + t_quantn)
    return {
        # the faster of the two runs, with the threads it used (the oracle restates the reference's
        # per-read malloc'd lists, which limits its thread scaling)
        'value': max(single, multi), 'unit': 'pairs/s', 'cores': cores if multi > single else 1,
        'kind': 'port', 'single_thread_value': single, 'all_core_value': multi, 'host_cores': cores,
        'sample': '%d pairs of the same read set; 1 thread: oracle map %.2fs (%.0f pairs/s) + classes/EM %.2fs '
                  '(%d EM steps); %d threads: map %.2fs (%.0f pairs/s) + classes/EM %.2fs'
                  % (sample_units, t_map1, sample_units / t_map1, t_quant, iters, cores, t_mapn,
                     sample_units / t_mapn, t_quantn),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--genes', type=int, default=20000, help='synthetic genes (20000 -> ~190k tx)')
    ap.add_argument('--pairs', type=int, default=10_000_000, help='read pairs per GPU per step')
    ap.add_argument('--read-len', type=int, default=100)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--cpu-sample', type=int, default=1_000_000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

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
        dist.init_process_group('gloo', rank=rank, world_size=world)   # rendezvous / barrier only

    from seekmer_amd import _native, index_builder, infer, mapper, synth
    hip = _native.hip()
    device = local_rank
    if os.environ.get('SKM_BENCH_ONE_DEVICE') == '1':      # rehearsal of the N > 1 path on a 1-GPU box
        device = 0
    if _native.device_count() <= device:
        raise SystemExit('no GPU %d visible: the benchmark has no CPU path' % device)

    # ---------------- setup (untimed): transcriptome, index, reads -> HBM
    t0 = time.perf_counter()
    ids, pool, tx_offsets = synth.transcriptome(args.seed, args.genes)
    index = index_builder.build_pooled(ids, pool, tx_offsets)
    n_tx = len(ids)
    log('rank %d: %d transcripts, %d k-mer slots, index built in %.1fs'
        % (rank, n_tx, index.kmers.size, time.perf_counter() - t0))
    n_units = args.pairs
    bases, offsets = synth.reads(args.seed, pool, tx_offsets, rank * n_units, n_units,
                                 args.read_len, True)
    index.device_handle(device)
    d_bases, d_offsets = ctypes.c_void_p(), ctypes.c_void_p()
    _native.check(hip.skm_device_malloc(device, bases.size, ctypes.byref(d_bases)))
    _native.check(hip.skm_device_malloc(device, offsets.size * 8, ctypes.byref(d_offsets)))
    _native.check(hip.skm_device_upload(device, d_bases, bases.ctypes.data, bases.size))
    _native.check(hip.skm_device_upload(device, d_offsets, offsets.ctypes.data, offsets.size * 8))
    lengths = np.ascontiguousarray(index.transcripts['length'], dtype='f8')
    result = mapper.MapResult(index, device=device)
    comm_id = None
    force_comm = world == 1 and os.environ.get('SKM_FORCE_COMM') == '1'   # 1-rank RCCL rehearsal
    if force_comm:
        raw = ctypes.create_string_buffer(128)
        _native.check(hip.skm_comm_unique_id(raw))
        comm_id = raw.raw
    if world > 1:
        import torch
        buf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            raw = ctypes.create_string_buffer(128)
            _native.check(hip.skm_comm_unique_id(raw))
            buf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
        dist.broadcast(buf, 0)
        comm_id = bytes(buf.numpy().tobytes())
    comm = ctypes.c_void_p()
    if comm_id is not None:                # one communicator per process, reused by every step
        _native.check(hip.skm_comm_create(device, comm_id, rank, world, ctypes.byref(comm)))

    state = {}

    stage = {}
    profile_stages = os.environ.get('SKM_BENCH_PROFILE') in ('1', '2')   # 2: host wall per call, no syncs
    profile_sync = os.environ.get('SKM_BENCH_PROFILE') == '1'

    def mark(name, t0):
        if profile_stages:
            if profile_sync:
                _native.check(hip.skm_device_synchronize(device))
            stage[name] = stage.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    def step():
        t = time.perf_counter()
        result.reset()
        t = mark('reset', t)
        result.map_resident(d_bases, d_offsets, n_units, True, args.read_len)
        t = mark('map_batch', t)
        before = result.timing()
        # fragment lengths (RCCL all-reduce over the ranks) -> effective lengths -> start
        # vector -> EM -> TPM: one native call on the resident class table
        tpm, iters = infer.quantify_resident(result, comm=comm if comm else None, return_iters=True)
        t = mark('quantify', t)
        after = result.timing()
        state['em'] = {'em_ns': after['em_ns'] - before['em_ns'],
                       'iterations': after['em_iterations'] - before['em_iterations']}
        state['tpm'] = tpm
        state['iters'] = iters

    def barrier():
        if dist is not None:
            dist.barrier()
        _native.check(hip.skm_device_synchronize(device))

    for _ in range(args.warmup):
        step()
    t_before = result.timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    t_after = result.timing()
    if profile_stages and rank == 0:
        log('stage ms per step: ' + ', '.join('%s %.2f' % (k, 1e3 * v / (args.steps + args.warmup))
                                              for k, v in stage.items()))

    # ---------------- roofline of the dominant kernel (map_units_kernel)
    launches = t_after['batches'] - t_before['batches']
    map_ns = (t_after['map_ns'] - t_before['map_ns']) / max(launches, 1)
    pack_ns = (t_after['pack_ns'] - t_before['pack_ns']) / max(launches, 1)
    class_ns = (t_after['class_ns'] - t_before['class_ns']) / max(launches, 1)
    result.set_stats(True)                 # instrumented build, untimed: access counts per launch
    result.reset()
    result.map_resident(d_bases, d_offsets, n_units, True, args.read_len)
    st = result.access_stats()
    result.set_stats(False)
    algorithmic = (st['read_bases'] + 16 * st['slots'] + 48 * st['contig_reads']
                   + 8 * (st['targets_copied'] + st['targets_merged']) + 8 * st['seq_fetches']
                   + 4 * st['tuple_ids'])
    achieved = algorithmic / map_ns if map_ns else 0.0          # bytes/ns = GB/s
    sizes = result.sizes()
    # HBM traffic of the same launch from the PMC passes (FETCH_SIZE + WRITE_SIZE, rocprofv3,
    # separate --pmc runs, profiles/r01_i_pmc_map.json); only quoted for the workload it was taken on
    traffic, miss_rate = None, None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_i_pmc_map.json')) as f:
            pmc = json.load(f)
        if args.genes == 20000 and n_units == 10_000_000 and args.read_len == 100:
            traffic = pmc['derived']['hbm_traffic_bytes']
            miss_rate = pmc['per_launch']['TCC_MISS_sum'] / (map_ns * 1e-9)
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        line = {
            'metric': 'paired reads/sec mapped+quantified',
            'value': world * n_units * args.steps / elapsed,
            'unit': 'pairs/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1000.0 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'u64+f64',
            'data': 'synthetic',
            'config': {
                'workload': 'configs[1]: synthetic ~190k-tx index (stand-in for ENSEMBL GRCh38 cDNA), '
                            '%d 2x%dbp synthetic pairs per GPU, map+classes+EM to the reference stop rule'
                            % (n_units, args.read_len),
                'transcripts': n_tx, 'kmer_slots': int(index.kmers.size),
                'pairs_per_gpu': n_units, 'read_len': args.read_len,
                'classes': sizes[0], 'class_map_rows': sizes[1],
                'em_iterations': int(state['iters']),
                'em_iters_per_s': state['em']['iterations'] / (state['em']['em_ns'] * 1e-9),
                # B_em = M*(4+8+16) + C*(4+8) + T*40 algorithmic bytes per EM step
                'em_bytes_per_step': int(sizes[1] * 28 + sizes[0] * 12 + n_tx * 40),
                'em_algorithmic_GBps': (sizes[1] * 28 + sizes[0] * 12 + n_tx * 40)
                                       * state['em']['iterations'] / max(state['em']['em_ns'], 1.0),
                'parallelism': 'reads sharded x%d, RCCL all-reduce f64[T] per EM step' % world,
                'phase_ms': {'pack': pack_ns * 1e-6, 'map': map_ns * 1e-6, 'classes': class_ns * 1e-6,
                             'em': state['em']['em_ns'] * 1e-6},
            },
            'roofline': {
                'kernel': 'map_units_kernel', 'bound': 'hbm',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'algorithmic_bytes_per_launch': algorithmic,
                # the probes are dependent random 16-B reads: the relevant ceiling is the chip's
                # random-gather rate (52 G/s over a 2 GiB table, profiles/r01_gather_ceiling.log)
                'l2_miss_rate_vs_random_gather_ceiling': (miss_rate / 52.3e9) if miss_rate else None,
                'bytes_per_pair': algorithmic / n_units,
                'launch_ms': map_ns * 1e-6,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(index, pool, tx_offsets, args.seed, args.read_len,
                                                min(args.cpu_sample, n_units))
        print(json.dumps(line), flush=True)
    if comm:
        _native.check(hip.skm_comm_destroy(comm))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
