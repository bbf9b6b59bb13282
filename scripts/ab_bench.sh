#!/bin/bash
# A/B of tuning builds on ONE box: scripts/ab_bench.sh OUTDIR REPEATS name1 name2 ...  ("base" = the product library)
# One index is built and cached; every build then runs the resident configs[1] step REPEATS times.
out=$1; reps=$2; shift 2
mkdir -p $out
cache=/tmp/skm_ab_index.npz
for r in $(seq 1 $reps); do
  for name in "$@"; do
    lib=seekmer_amd/libseekmer_hip_$name.so
    [ "$name" = base ] && lib=seekmer_amd/libseekmer_hip.so
    SKM_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-e2e --no-other-configs --no-cpu-baseline --steps 10 --index-cache $cache \
      > $out/${name}_$r.json 2> $out/${name}_$r.err || { echo "$name run $r failed"; tail -5 $out/${name}_$r.err; exit 1; }
    python - "$out/${name}_$r.json" "$name" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d['config']['phase_ms']
print('%-10s %.1f M/s  step %.2f ms  map %.3f  classes %.3f  pack %.3f  em %.3f' % (sys.argv[2], d['value'] / 1e6, d['ms_per_step'], p['map'], p['classes'], p['pack'], p['em']), flush=True)
P
  done
done
