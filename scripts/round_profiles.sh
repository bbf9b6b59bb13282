#!/bin/bash
# The round's measured evidence in one call on the GPU box:  bash scripts/round_profiles.sh <tag>
#   gpurun_out/<tag>_bench.json            default bench line (configs[1] with cpu_baseline, e2e and other_configs)
#   gpurun_out/prof_<tag>_kernel_stats.csv rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/pmc_<tag>*.json             PMC passes over the map and class kernels (scripts/pmc_map.sh)
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
CACHE=/tmp/skm_idx.npz
cd $ROOT
# PMC passes first: bench.py quotes roofline.traffic from the committed summary of THIS build
bash scripts/pmc_map.sh $TAG class_insert_kernel class_verify_kernel pack_reads_kernel > $OUT/pmc_$TAG.log 2>&1
ROUND=${ROUND:-r04}
python3 scripts/pmc_finish.py $OUT/pmc_$TAG.json profiles/${ROUND}_pmc_map.json
cp profiles/${ROUND}_pmc_map.json $OUT/${TAG}_pmc_map.json
python3 bench.py --index-cache $CACHE > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
mkdir -p $OUT/prof_$TAG
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG --output-format csv -- \
    python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --index-cache $CACHE > $OUT/prof_$TAG/bench.json 2> $OUT/prof_$TAG/bench.err )
find $OUT/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/prof_${TAG}_kernel_stats.csv \;
find $OUT/prof_$TAG -name "*.csv" -size +2M -delete
