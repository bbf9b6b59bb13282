#!/bin/bash
# A/B of an environment setting on ONE box: scripts/ab_env.sh OUTDIR REPEATS VAR value1 value2 ... ("-" = unset)
out=$1; reps=$2; var=$3; shift 3
mkdir -p $out
cache=/tmp/skm_ab_index.npz
for r in $(seq 1 $reps); do
  for value in "$@"; do
    if [ "$value" = "-" ]; then unset $var; else export $var="$value"; fi
    timeout -k 10 300 python bench.py --no-e2e --no-other-configs --no-cpu-baseline --steps 10 --index-cache $cache \
      > $out/run.json 2> $out/run.err || { echo "$var=$value run $r failed"; tail -5 $out/run.err; exit 1; }
    python - "$out/run.json" "$var=$value" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d['config']['phase_ms']
print('%-24s %.1f M/s  step %.2f ms  map %.3f  classes %.3f  pack %.3f  em %.3f' % (sys.argv[2], d['value'] / 1e6, d['ms_per_step'], p['map'], p['classes'], p['pack'], p['em']), flush=True)
P
  done
done
