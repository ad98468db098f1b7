#!/bin/bash
# What does one more table read / table store / dependent read cost the global-table kernel?  Experiment builds
# (EXP_* macros, removed from the tree after the measurement): every slot read is doubled or tripled by a read of the same
# slot in a second table, every table store likewise, or a dependent read is chained behind every slot read.
ROOT=$PWD
i=0
for v in ${K1_MC_BUILDS:-"-DX0" "-DEXP_DUP_READ=1" "-DEXP_DUP_READ=2" "-DEXP_DUP_STORE=1" "-DEXP_DUP_STORE=2" "-DEXP_CHAIN_READ" "-DEXP_HOT_READ" "-DEXP_HOT_STORE"}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden $v pim-compression_amd/csrc/snappy_hip.hip -o pim-compression_amd/libk1mc_$i.so || exit 1
  echo "== [$v]"
  SNAPPY_PROF_LIB=$ROOT/pim-compression_amd/libk1mc_$i.so timeout -k 10 200 python3 tools/exp_variants.py 2048 "SNAPPY_HIP_LDS_WAVES=0" "SNAPPY_HIP_LDS_WAVES=0,SNAPPY_HIP_GT_WAVES=4096" "X=mix" 2>&1 | grep "GB/s" | grep -v decompress || exit 1
  i=$((i+1))
done
rm -f pim-compression_amd/libk1mc_*.so
