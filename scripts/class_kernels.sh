#!/bin/bash
# Per-kernel times of the class stage (binned and plain) under rocprofv3 --kernel-trace --stats:
#   bash scripts/class_kernels.sh OUTDIR
OUT=$(realpath -m $1)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/scripts/profile_map.py --reps 1 --cache /tmp/skm_idx.npz > $OUT/warm.log 2>&1
for mode in bins plain; do
  if [ $mode = plain ]; then export SKM_NO_CLASS_BINS=1; else unset SKM_NO_CLASS_BINS; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/$mode --output-format csv -- \
      python3 $ROOT/scripts/profile_map.py --reps 3 --again --cache /tmp/skm_idx.npz > $OUT/$mode.log 2>&1 || { echo "$mode failed"; tail -3 $OUT/$mode.log; }
  grep "^rep\|^again" $OUT/$mode.log
  python3 - $OUT/$mode $mode <<'P'
import csv, glob, sys
rows = {}
for path in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        if 'class_' in n:
            name = n.split('(')[0].replace('skm::', '')
            rows.setdefault(name, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
for k, v in sorted(rows.items()):
    print('%-6s %-28s launches %3d  us: %s' % (sys.argv[2], k, len(v), ' '.join('%.0f' % x for x in v[-8:])), flush=True)
P
  find $OUT/$mode -name "*.csv" -size +1M -delete
done
