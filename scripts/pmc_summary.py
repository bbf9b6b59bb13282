"""Average per-launch PMC values of one kernel from rocprofv3 counter CSVs.
    python3 scripts/pmc_summary.py <dir with pass*/> <kernel name substring>
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; the HBM figures below
apply the unit only (random 16-B gathers: one 64-B request per miss, so the
streaming-read half-count correction of the micro-architecture guide does not
apply -- FETCH_SIZE*1024 is cross-checked against TCC_MISS*64 B instead)."""
import csv
import glob
import json
import os
import sys


def main():
    root, name = sys.argv[1], sys.argv[2]
    sums, counts = {}, {}
    durations, map_durations = [], []
    for path in glob.glob(os.path.join(root, 'pass*', '**', '*_counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if name not in row['Kernel_Name']:
                    continue
                key = row['Counter_Name']
                sums[key] = sums.get(key, 0.0) + float(row['Counter_Value'])
                counts[key] = counts.get(key, 0) + 1
    for path in glob.glob(os.path.join(root, 'pass*', '**', '*_kernel_trace.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if name in row['Kernel_Name']:
                    durations.append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6)
                if 'map_units_kernel' in row['Kernel_Name']:
                    map_durations.append(0.0)
    per_launch = {k: sums[k] / counts[k] for k in sorted(sums)}
    out = {'kernel': name, 'per_launch': per_launch, 'launches_seen': max(counts.values()) if counts else 0}
    if durations:
        out['launch_ms_under_pmc'] = sum(durations) / len(durations)
    # a kernel that runs several times per batch (the class kernels: one launch per wave of records):
    # counters and time summed over the launches of one batch
    per_batch = len(durations) / max(len(map_durations), 1) if map_durations else 1.0
    if per_batch > 1.01:
        out['launches_per_batch'] = per_batch
        out['per_batch'] = {k: v * per_batch for k, v in per_launch.items()}
        out['batch_ms_under_pmc'] = sum(durations) / len(map_durations)
    # the kernel's own launch time: for the map kernel the HIP-event time of an unprofiled run of the
    # same driver (plain.log); for every other kernel its duration in the kernel traces of the
    # counter passes (a counter pass slows a launch by a few per cent)
    plain = os.path.join(root, 'plain.log')
    if name.startswith('map_units') and os.path.exists(plain):
        ms = [float(line.split(' map ')[1].split()[0]) for line in open(plain) if line.startswith('rep ')]
        if ms:
            out['launch_ms'] = min(ms)
            out['launch_ms_source'] = 'HIP events, unprofiled run (plain.log)'
    elif durations:
        out['launch_ms'] = sum(durations) / len(durations)
        out['launch_ms_source'] = 'kernel trace of the counter passes (mean over the launches)' 
    d = {}
    if 'FETCH_SIZE' in per_launch and 'WRITE_SIZE' in per_launch:
        d['hbm_traffic_bytes'] = (per_launch['FETCH_SIZE'] + per_launch['WRITE_SIZE']) * 1024
        if 'per_batch' in out:
            d['hbm_traffic_bytes_per_batch'] = d['hbm_traffic_bytes'] * per_batch
    if 'TCC_MISS_sum' in per_launch:
        d['tcc_miss_bytes_at_64B'] = per_launch['TCC_MISS_sum'] * 64
        if 'launch_ms' in out:
            d['l2_miss_rate_per_s'] = per_launch['TCC_MISS_sum'] / (out['launch_ms'] * 1e-3)
    out['derived'] = d
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
