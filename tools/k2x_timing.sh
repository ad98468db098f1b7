#!/bin/bash
# Where does K2's time go?  Builds the library with one phase of the per-window decoder removed at a time -- the global
# loads of far copies (-DK2X_NO_FAR), the LDS rounds of near copies (-DK2X_NO_NEAR), the flush of the stage
# (-DK2X_NO_FLUSH) -- and times K2 on a 2 GiB container.  The experimental builds produce WRONG bytes on purpose; they
# exist only inside this script.
ROOT=$PWD
for v in "" "-DK2X_NO_FAR" "-DK2X_NO_NEAR" "-DK2X_NO_FLUSH" "-DK2X_NO_FAR -DK2X_NO_NEAR -DK2X_NO_FLUSH"; do
  t=$(echo "$v" | tr -d ' -')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v pim-compression_amd/csrc/snappy_hip.hip -o pim-compression_amd/libk2x_$t.so || exit 1
done
for l in pim-compression_amd/libk2x_*.so; do
  echo "== $l"
  SNAPPY_PROF_LIB=$ROOT/$l timeout -k 10 200 python3 tools/exp_variants.py 2048 "3:0" 2>&1 | grep decompress || exit 1
done
rm -f pim-compression_amd/libk2x_*.so
