#!/bin/bash
# The final K1 mix (cached global-table wavefronts + 256 LDS-table ones) against the total wavefront count, and the cached
# kernel alone: where does adding wavefronts stop paying?  (SNAPPY_HIP_GT_WAVES = total of both kinds.)
args=()
for w in 2304 2816 3328 3840 4352 4864 5376; do args+=("SNAPPY_HIP_LDS_WAVES=256,SNAPPY_HIP_GT_WAVES=$w"); done
for w in 2048 3072 4096 5120 6656; do args+=("SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_GT_WAVES=$w"); done
timeout -k 10 600 python3 tools/exp_variants.py 2048 "${args[@]}" 2>&1 | grep "GB/s" | grep -v decompress
