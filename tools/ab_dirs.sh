#!/bin/bash
# A/B of two source trees of the library on ONE box: tools/ab_dirs.sh <csrc dir A> <csrc dir B> ; each built and timed on the
# product's mix, the LDS-table kernel alone and the global-table kernel alone (tools/exp_variants.py, 2 GiB), twice, alternating.
ROOT=$PWD
i=0
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -I$ROOT/include $d/snappy_hip.hip -o pim-compression_amd/libab_$i.so || exit 1
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for d in "$@"; do
    echo "== [$d]"
    SNAPPY_PROF_LIB=$ROOT/pim-compression_amd/libab_$i.so timeout -k 10 300 python3 tools/exp_variants.py 2048 "X=0" "SNAPPY_HIP_COMPRESS_VARIANT=1" "SNAPPY_HIP_LDS_WAVES=0" "SNAPPY_HIP_K1_STREAM=0" 2>&1 | grep "GB/s" | grep -v decompress || exit 1
    i=$((i+1))
  done
done
rm -f pim-compression_amd/libab_*.so
