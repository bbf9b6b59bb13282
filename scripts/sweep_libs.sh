set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 scripts/profile_map.py --reps 1 --cache /tmp/skm_idx.npz > gpurun_out/libs_sweep.log 2>&1
for f in seekmer_amd/libseekmer_hip.so seekmer_amd/libseekmer_hip_t*.so; do
  echo "== $f" >> gpurun_out/libs_sweep.log
  SKM_HIP_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 120 python3 scripts/profile_map.py --reps 3 --cache /tmp/skm_idx.npz 2>&1 | grep -E "rep [12]|rror" >> gpurun_out/libs_sweep.log
done
