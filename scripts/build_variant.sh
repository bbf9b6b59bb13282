#!/bin/bash
# A tuning build of the GPU library beside the product one:
#   scripts/build_variant.sh NAME "-DSKM_MAP_PREFETCH=1" [file.hip ...]   (default: skm_map.hip)
# compiles the named sources with the extra flags, links them with the product's other objects into
# seekmer_amd/libseekmer_hip_NAME.so; run with SKM_HIP_LIB=seekmer_amd/libseekmer_hip_NAME.so.
set -e
name=$1; flags=$2; shift 2
files=${@:-skm_map.hip}
cd "$(dirname "$0")/../seekmer_amd/csrc"
make -s ../libseekmer_hip.so
mkdir -p build_$name
objs=""
for src in skm_abi.hip skm_map.hip skm_classes.hip skm_em.hip skm_em_batch.hip skm_quant_setup.hip skm_pool.hip; do
  if [[ " $files " == *" $src "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function $flags -c $src -o build_$name/${src%.hip}.o
    objs="$objs build_$name/${src%.hip}.o"
  else
    objs="$objs build/${src%.hip}.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libseekmer_hip_$name.so $objs -ldl
echo built seekmer_amd/libseekmer_hip_$name.so
