#!/bin/bash
# K1 forms side by side on ONE box (tools/exp_variants.py, 2 GiB container): the product mix, the LDS-table kernel alone and the
# global-table kernel alone, each with the bulk form (SNAPPY_HIP_K1_STREAM=0) and the stream form (3 = both kernels).
MIB=${1:-2048}
timeout -k 10 900 python3 tools/exp_variants.py $MIB \
  "SNAPPY_HIP_K1_STREAM=0" "SNAPPY_HIP_K1_STREAM=3" "SNAPPY_HIP_K1_STREAM=1" "SNAPPY_HIP_K1_STREAM=2" \
  "SNAPPY_HIP_K1_STREAM=0,SNAPPY_HIP_COMPRESS_VARIANT=1" "SNAPPY_HIP_K1_STREAM=3,SNAPPY_HIP_COMPRESS_VARIANT=1" \
  "SNAPPY_HIP_K1_STREAM=0,SNAPPY_HIP_LDS_WAVES=0" "SNAPPY_HIP_K1_STREAM=3,SNAPPY_HIP_LDS_WAVES=0" \
  "SNAPPY_HIP_K1_STREAM=3,SNAPPY_HIP_LDS_WAVES=1024,SNAPPY_HIP_GT_WAVES=3328" \
  "SNAPPY_HIP_K1_STREAM=1,SNAPPY_HIP_LDS_WAVES=1024,SNAPPY_HIP_GT_WAVES=3328" 2>&1 | grep "GB/s"
