#!/bin/bash
# The write-back slot cache in front of the global table (SNAPPY_HIP_GT_CACHE = slots; 256 and 1024 exist in the ablation
# build only, so the sweep runs on it: python tools/build_ablation.py): the global-table kernel alone and mixes with 0 / 1 / 2
# LDS-table wavefronts per CU, bulk form (SNAPPY_HIP_K1_STREAM=1) and stream form (3) on the global-table kernel.
# 2 GiB container, one box.
args=("SNAPPY_HIP_K1_STREAM=1,SNAPPY_HIP_LDS_WAVES=0,X=no_cache_alone" "SNAPPY_HIP_K1_STREAM=1,X=no_cache_mix")
for s in 1 3; do
  for c in 256 512 1024; do
    for lds in 0 256 512; do args+=("SNAPPY_HIP_K1_STREAM=$s,SNAPPY_HIP_GT_CACHE=$c,SNAPPY_HIP_LDS_WAVES=$lds"); done
  done
done
SNAPPY_PROF_LIB=$PWD/pim-compression_amd/libsnappy_hip_ablation.so timeout -k 10 800 python3 tools/exp_variants.py 2048 "${args[@]}" 2>&1 | grep "GB/s" | grep -v decompress
