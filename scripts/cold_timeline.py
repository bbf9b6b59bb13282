"""Where the time of ONE cold FASTQ -> TPM pass goes: HIP API calls of the timed region of
`python bench.py --cold-child` from a rocprofv3 --hip-trace CSV (the child brackets its timed region
with two hipDeviceSynchronize calls).
    python3 scripts/cold_timeline.py <dir with *_hip_api_trace.csv> [min_ms]"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    paths = glob.glob(os.path.join(root, '**', '*hip_api_trace.csv'), recursive=True)
    rows = []
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function'], r.get('Thread_Id', '')))
    rows.sort()
    syncs = [r for r in rows if r[2] == 'hipDeviceSynchronize']
    if len(syncs) < 2:
        print('no bracket found (%d hipDeviceSynchronize calls)' % len(syncs))
        return
    t0, t1 = syncs[-2][1], syncs[-1][0]
    print('timed region: %.1f ms' % ((t1 - t0) * 1e-6))
    inside = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    totals = {}
    for s, e, name, tid in inside:
        n, t = totals.get(name, (0, 0))
        totals[name] = (n + 1, t + e - s)
    print('-- per API (calls, total ms over all threads)')
    for name, (n, t) in sorted(totals.items(), key=lambda kv: -kv[1][1])[:14]:
        print('%-32s %6d %9.2f' % (name, n, t * 1e-6))
    print('-- calls of at least %.1f ms (start ms, duration ms, thread)' % floor)
    for s, e, name, tid in inside:
        if (e - s) * 1e-6 >= floor:
            print('%8.2f %8.2f  %-28s %s' % ((s - t0) * 1e-6, (e - s) * 1e-6, name, tid))


if __name__ == '__main__':
    main()
