#!/bin/bash
# Re-fit of K1's wave mix with the stream form on the LDS-table wavefronts (36 KiB of LDS each instead of 33): LDS-table
# wavefronts x total K1 wavefronts (SNAPPY_HIP_GT_WAVES is the total of both kinds); 2 GiB container, one box.
args=()
for lds in 512 768 1024; do
  for tot in 4608 5120 5632 5888 6400; do
    args+=("SNAPPY_HIP_LDS_WAVES=$lds,SNAPPY_HIP_GT_WAVES=$tot")
  done
done
timeout -k 10 800 python3 tools/exp_variants.py 2048 "X=0" "${args[@]}" "X=1" 2>&1 | grep "GB/s" | grep -v decompress
