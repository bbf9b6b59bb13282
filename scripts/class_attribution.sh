#!/bin/bash
# Where class_insert_kernel's time goes: the product build against timing-only builds without its
# atomics, without its probe, and without both (scripts/build_variant.sh cxN "-DSKM_CLASS_EXPERIMENT=N"
# skm_classes.hip), each under rocprofv3 --kernel-trace --stats over scripts/profile_map.py (one
# 10 M-pair batch mapped on a fresh table, then once more on the table that holds every class).
#   bash scripts/class_attribution.sh OUTDIR
OUT=$(realpath -m $1)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/scripts/profile_map.py --reps 1 --cache /tmp/skm_idx.npz > $OUT/warm.log 2>&1
for name in ${VARIANTS:-base cx1}; do
  lib=$ROOT/seekmer_amd/libseekmer_hip_$name.so
  [ "$name" = base ] && lib=$ROOT/seekmer_amd/libseekmer_hip.so
  export SKM_HIP_LIB=$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/$name --output-format csv -- \
      python3 $ROOT/scripts/profile_map.py --reps 3 --again --cache /tmp/skm_idx.npz > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -3 $OUT/$name.log; }
  python3 - $OUT/$name $name <<'P'
import csv, glob, sys
rows = {}
for path in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        for key in ('class_insert', 'class_verify', 'map_units'):
            if key in n:
                rows.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
# three batches on a fresh table (one or two launches each: a large batch on an empty table goes in two
# waves of records), then the same batch once more on the table that holds every class (one launch)
print(sys.argv[2], ' '.join('%s: fresh table %.3f ms per batch (%d launches), every class known %.3f ms'
                            % (k, sum(v[:-1]) / 3, (len(v) - 1) // 3, v[-1]) for k, v in sorted(rows.items()) if len(v) > 3), flush=True)
P
  find $OUT/$name -name "*.csv" -size +1M -delete
done
