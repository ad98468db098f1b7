#!/bin/bash
# Round-3 soak of the final build on one box: soak (co-running kernels, big containers against the oracle), fuzz slices of
# the three launch shapes, the drop-in pair under stress, the ablation build's configurations.
OUT=gpurun_out/r03_soak
mkdir -p $OUT
timeout -k 10 500 python3 tools/soak.py 18 512 > $OUT/soak.txt 2>&1; echo "soak rc=$?"; tail -n 2 $OUT/soak.txt
timeout -k 10 300 python3 tools/fuzz_gpu.py 500 778 > $OUT/fuzz_default.txt 2>&1; echo "fuzz default rc=$?"; tail -n 1 $OUT/fuzz_default.txt
SNAPPY_HIP_LDS_WAVES=0 timeout -k 10 300 python3 tools/fuzz_gpu.py 500 779 > $OUT/fuzz_global_table_cached.txt 2>&1; echo "fuzz global-table (cached) rc=$?"; tail -n 1 $OUT/fuzz_global_table_cached.txt
SNAPPY_HIP_LDS_WAVES=5 SNAPPY_HIP_GT_WAVES=11 SNAPPY_HIP_HYBRID_MIN_BLOCKS=1 timeout -k 10 300 python3 tools/fuzz_gpu.py 500 780 > $OUT/fuzz_tiny_hybrid.txt 2>&1; echo "fuzz tiny hybrid rc=$?"; tail -n 1 $OUT/fuzz_tiny_hybrid.txt
SNAPPY_HIP_LDS_WAVES=0 SNAPPY_HIP_K1_STREAM=1 timeout -k 10 300 python3 tools/fuzz_gpu.py 300 781 > $OUT/fuzz_global_table_cached_bulk.txt 2>&1; echo "fuzz global-table (cached, bulk form) rc=$?"; tail -n 1 $OUT/fuzz_global_table_cached_bulk.txt
timeout -k 10 400 python3 tools/dropin_stress.py 40 3 > $OUT/dropin_stress.txt 2>&1; echo "dropin stress rc=$?"; tail -n 2 $OUT/dropin_stress.txt
timeout -k 10 500 python3 tools/ablation_check.py > $OUT/ablation_check.txt 2>&1; echo "ablation check rc=$?"; tail -n 2 $OUT/ablation_check.txt
