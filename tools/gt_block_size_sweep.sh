#!/bin/bash
# The global-table kernel alone (and the LDS-table kernel alone) against the block size: the hash table of a block of n bytes
# has min(16384, 2^ceil(log2 n)) slots, so smaller blocks mean a smaller random-access footprint per wavefront.
for bs in 2048 4096 8192 16384 32768 65536; do
  echo "== block size $bs"
  EXP_BLOCK_SIZE=$bs timeout -k 10 300 python3 tools/exp_variants.py 2048 "SNAPPY_HIP_LDS_WAVES=0" "SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_GT_WAVES=4096" "SNAPPY_HIP_COMPRESS_VARIANT=1" "X=mix" 2>&1 | grep "GB/s" | grep -v decompress
done
