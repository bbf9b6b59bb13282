#!/bin/bash
# One extra PMC pass over the map kernel:  bash scripts/pmc_extra.sh <tag> COUNTER [COUNTER ...]
# (own run, --kernel-trace only, as the pool requires); prints the per-launch averages.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pass --output-format csv -- \
    python3 $ROOT/scripts/profile_map.py --reps 2 --cache /tmp/skm_idx.npz > $OUT/pass.log 2>&1
python3 - $OUT <<'P'
import csv, glob, sys, collections
sums, n = collections.Counter(), collections.Counter()
for path in glob.glob(sys.argv[1] + '/pass/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        if 'map_units_kernel' in row['Kernel_Name']:
            sums[row['Counter_Name']] += float(row['Counter_Value'])
            n[row['Counter_Name']] += 1
for k in sorted(sums):
    print('%-32s %.4g per launch (%d rows)' % (k, sums[k] / max(1, n[k]) * (n[k] / max(1, n[k])), n[k]))
P
tail -3 $OUT/pass.log
find $OUT -name "*.csv" -size +2M -delete
