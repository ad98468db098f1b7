#!/bin/bash
# A/B of library builds on ONE box (box-to-box differences reach 10 % for K2): tools/k2_ab.sh <csrc dir or flags> ...
# An argument that is a directory is built from <dir>/snappy_hip.hip, anything else is passed to hipcc as flags for the
# working tree's csrc.  Each build is timed twice, alternating (tools/exp_variants.py, 2 GiB container).
ROOT=$PWD
i=0
for v in "$@"; do
  if [ -d "$v" ]; then srcf=$v/snappy_hip.hip; fl=""; else srcf=pim-compression_amd/csrc/snappy_hip.hip; fl=$v; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $fl $srcf -o pim-compression_amd/libk2ab_$i.so || exit 1
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for v in "$@"; do
    echo "== [$v]"
    SNAPPY_PROF_LIB=$ROOT/pim-compression_amd/libk2ab_$i.so timeout -k 10 200 python3 tools/exp_variants.py 2048 "3:0" 2>&1 | grep "decompress\|GB/s" || exit 1
    i=$((i+1))
  done
done
rm -f pim-compression_amd/libk2ab_*.so
