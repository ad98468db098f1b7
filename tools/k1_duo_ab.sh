#!/bin/bash
# K1's LDS-table forms side by side on ONE box (tools/exp_variants.py, 2 GiB container): bulk (0), stream (1), duo (5);
# in the product mix and alone.
MIB=${1:-2048}
timeout -k 10 600 python3 tools/exp_variants.py $MIB \
  "SNAPPY_HIP_K1_STREAM=1" "SNAPPY_HIP_K1_STREAM=5" "SNAPPY_HIP_K1_STREAM=5,SNAPPY_HIP_LDS_WAVES=1024" \
  "SNAPPY_HIP_K1_STREAM=1,SNAPPY_HIP_COMPRESS_VARIANT=1" "SNAPPY_HIP_K1_STREAM=5,SNAPPY_HIP_COMPRESS_VARIANT=1" \
  "SNAPPY_HIP_K1_STREAM=5,SNAPPY_HIP_LDS_WAVES=512" "SNAPPY_HIP_K1_STREAM=5,SNAPPY_HIP_LDS_WAVES=768,SNAPPY_HIP_GT_WAVES=4864" 2>&1 | grep "GB/s"
