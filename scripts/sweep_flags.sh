set -e
cd $GRAFT_REPO_ROOT
python3 scripts/profile_map.py --reps 1 --cache /tmp/skm_idx.npz > gpurun_out/flags_sweep.log 2>&1
for f in 0 1 2 3 4 5 7; do
  echo "== flags $f" >> gpurun_out/flags_sweep.log
  SKM_MAP_VOTE=1,1,1,1,1,1,$f timeout -k 10 120 python3 scripts/profile_map.py --reps 3 --cache /tmp/skm_idx.npz 2>&1 | grep -E "rep [12]" >> gpurun_out/flags_sweep.log
done
