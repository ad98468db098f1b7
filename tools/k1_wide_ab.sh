#!/bin/bash
# A/B of the content-carrying global table (SNAPPY_HIP_GT_WIDE=1: 32-byte slots, no candidate fetches) against the tagged
# 4-byte slots: the global-table kernel alone in both forms, then the product mix.  2 GiB container, one box.
timeout -k 10 600 python3 tools/exp_variants.py 2048 \
  "SNAPPY_HIP_LDS_WAVES=0,X=narrow_bulk" \
  "SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_GT_WIDE=1,X=wide_bulk" \
  "SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_K1_STREAM=3,X=narrow_stream" \
  "SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_K1_STREAM=3,SNAPPY_HIP_GT_WIDE=1,X=wide_stream" \
  "SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_GT_WIDE=1,SNAPPY_HIP_GT_WAVES=4096,X=wide_bulk_4096" \
  "SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_GT_WIDE=1,SNAPPY_HIP_GT_WAVES=2048,X=wide_bulk_2048" \
  "SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_K1_STREAM=3,SNAPPY_HIP_GT_WIDE=1,SNAPPY_HIP_GT_WAVES=4096,X=wide_stream_4096" \
  "X=mix_narrow" \
  "SNAPPY_HIP_GT_WIDE=1,X=mix_wide" \
  "SNAPPY_HIP_GT_WIDE=1,SNAPPY_HIP_K1_STREAM=3,X=mix_wide_stream" 2>&1 | grep "GB/s"
