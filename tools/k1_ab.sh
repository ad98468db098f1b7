#!/bin/bash
# A/B of library builds on ONE box for K1: tools/k1_ab.sh "<hipcc flags>" ... ; each build timed on the product's mix, the
# LDS-table kernel alone and the global-table kernel alone (tools/exp_variants.py, 2 GiB container), twice, alternating.
# SNAPPY_K1_AB_ENV="A=1,B=2" adds environment settings to every run.
ROOT=$PWD
i=0
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden $v pim-compression_amd/csrc/snappy_hip.hip -o pim-compression_amd/libk1ab_$i.so || exit 1
  i=$((i+1))
done
E=${SNAPPY_K1_AB_ENV:-X=0}
for rep in 1 2; do
  i=0
  for v in "$@"; do
    echo "== [$v]"
    SNAPPY_PROF_LIB=$ROOT/pim-compression_amd/libk1ab_$i.so timeout -k 10 300 python3 tools/exp_variants.py 2048 "$E" "$E,SNAPPY_HIP_COMPRESS_VARIANT=1" "$E,SNAPPY_HIP_LDS_WAVES=0" 2>&1 | grep "GB/s" | grep -v decompress || exit 1
    i=$((i+1))
  done
done
rm -f pim-compression_amd/libk1ab_*.so
