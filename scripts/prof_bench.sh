#!/bin/bash
# rocprofv3 kernel trace + stats of the bench (run on the GPU box): bash scripts/prof_bench.sh <tag>
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- \
    python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/prof_${TAG}_kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -exec cp {} $ROOT/gpurun_out/prof_${TAG}_kernel_trace.csv \;
