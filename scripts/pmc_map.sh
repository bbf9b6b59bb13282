#!/bin/bash
# PMC passes over the map kernel (one counter set per pass; --kernel-trace only,
# as the pool requires).  Run on the GPU box:  bash scripts/pmc_map.sh <tag>
# Output: gpurun_out/pmc_<tag>/pass*/ (raw CSV) and gpurun_out/pmc_<tag>.json (summary).
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/scripts/profile_map.py --reps 2 --cache /tmp/skm_idx.npz > $OUT/plain.log 2>&1
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "FETCH_SIZE TCC_HIT_sum" \
           "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pass$i --output-format csv -- \
      python3 $ROOT/scripts/profile_map.py --reps 2 --cache /tmp/skm_idx.npz > $OUT/pass$i.log 2>&1
done
python3 $ROOT/scripts/pmc_summary.py $OUT map_units_kernel > $ROOT/gpurun_out/pmc_$TAG.json
# further kernels of the same runs: bash scripts/pmc_map.sh <tag> class_insert_kernel class_verify_kernel ...
shift || true
for EXTRA in "$@"; do
  python3 $ROOT/scripts/pmc_summary.py $OUT $EXTRA > $ROOT/gpurun_out/pmc_${TAG}_$EXTRA.json
done
# raw per-dispatch CSVs are large; keep the summary and the logs
find $OUT -name "*.csv" -size +2M -delete
