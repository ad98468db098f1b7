#!/bin/bash
# The global-table kernel alone against the number of its wavefronts (= tables of 64 KiB in the scratch): does a table
# footprint under the 256 MiB of Infinity Cache pay for the lost wavefronts?
args=()
for w in 512 1024 1536 2048 2560 3072 3584 4096 5120 6144 8192; do args+=("SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_GT_WAVES=$w"); done
timeout -k 10 600 python3 tools/exp_variants.py 2048 "${args[@]}" 2>&1 | grep "GB/s" | grep -v decompress
