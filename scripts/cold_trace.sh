#!/bin/bash
# HIP API timeline + kernel statistics of ONE cold pass of the FASTQ path (a process of its own, the
# bench's `cold_process` leg):   bash scripts/cold_trace.sh OUTDIR [GENES]
OUT=$(realpath -m $1)
GENES=${2:-2000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
python3 - <<P
import os, sys
sys.path.insert(0, "$ROOT")
from seekmer_amd import index_builder, synth
ids, pool, tx = synth.transcriptome(1, $GENES)
index_builder.build_pooled(ids, pool, tx).save("/dev/shm/skm_cold_index.npz")
bases, _ = synth.reads(1, pool, tx, 0, 10_000_000, 100, True)
synth.write_fastq(bases, 10_000_000, 100, True, "/dev/shm/skm_cold_1.fastq", "/dev/shm/skm_cold_2.fastq")
P
cd /tmp && export TMPDIR=/tmp
for k in 1 2; do
  python3 $ROOT/bench.py --cold-child /dev/shm/skm_cold_index.npz /dev/shm/skm_cold_1.fastq /dev/shm/skm_cold_2.fastq 2>> $OUT/plain.err | tail -1
done | tee $OUT/plain.log
for threads in 14 8; do
  SKM_COLD_PARSE_ONLY=1 python3 $ROOT/bench.py --cold-child /dev/shm/skm_cold_index.npz /dev/shm/skm_cold_1.fastq /dev/shm/skm_cold_2.fastq --parse-threads $threads 2>> $OUT/plain.err | tail -1
done | tee $OUT/parse_only.log
for mb in 4 16; do
  python3 $ROOT/bench.py --cold-child /dev/shm/skm_cold_index.npz /dev/shm/skm_cold_1.fastq /dev/shm/skm_cold_2.fastq --e2e-chunk-mb $mb 2>> $OUT/plain.err | tail -1
done | tee $OUT/chunks.log
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --stats -d $OUT/prof --output-format csv -- \
    python3 $ROOT/bench.py --cold-child /dev/shm/skm_cold_index.npz /dev/shm/skm_cold_1.fastq /dev/shm/skm_cold_2.fastq > $OUT/run.log 2>&1
find $OUT/prof -name "*hip_api_stats.csv" -exec cp {} $OUT/hip_api_stats.csv \;
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 $ROOT/scripts/cold_timeline.py $OUT/prof 0.5 > $OUT/timeline.txt 2>&1
find $OUT/prof -name "*.csv" -size +1M -delete
rm -f /dev/shm/skm_cold_index.npz /dev/shm/skm_cold_1.fastq /dev/shm/skm_cold_2.fastq
tail -1 $OUT/run.log
head -60 $OUT/timeline.txt
