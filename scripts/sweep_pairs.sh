set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 scripts/profile_map.py --reps 1 --pairs 1000 --cache /tmp/skm_idx.npz > gpurun_out/pairs_sweep.log 2>&1
for n in 1250000 2500000 5000000 10000000 20000000; do
  echo "== pairs $n" >> gpurun_out/pairs_sweep.log
  timeout -k 10 200 python3 scripts/profile_map.py --reps 3 --pairs $n --cache /tmp/skm_idx.npz 2>&1 | grep -E "rep [12]|rror" >> gpurun_out/pairs_sweep.log
done
