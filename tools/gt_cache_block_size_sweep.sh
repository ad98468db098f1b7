#!/bin/bash
# The slot cache against the block size (the LDS-table wavefronts' default count depends on it): product default of round 2
# (no cache) against cache 512 / 256 + stream form on the global-table kernel, with the default LDS-table wave count and fixed ones.
for bs in 4096 8192 16384 32768 65535; do
  echo "== block size $bs"
  args=("SNAPPY_HIP_K1_STREAM=1,X=no_cache")
  for c in 256 512; do
    args+=("SNAPPY_HIP_GT_CACHE=$c,SNAPPY_HIP_K1_STREAM=3")
    for lds in 0 256 512 1024; do args+=("SNAPPY_HIP_GT_CACHE=$c,SNAPPY_HIP_K1_STREAM=3,SNAPPY_HIP_LDS_WAVES=$lds"); done
  done
  SNAPPY_PROF_LIB=$PWD/pim-compression_amd/libsnappy_hip_ablation.so EXP_BLOCK_SIZE=$bs timeout -k 10 300 python3 tools/exp_variants.py 2048 "${args[@]}" 2>&1 | grep "GB/s" | grep -v decompress
done
