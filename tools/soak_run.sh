#!/bin/bash
# Soak of the final build on one box: bash tools/soak_run.sh <tag>.  Big containers against the oracle under co-running
# kernels, fuzz slices of the launch shapes that ship, the drop-in pair under stress.
OUT=gpurun_out/${1:-r04}_soak
mkdir -p $OUT
timeout -k 10 500 python3 tools/soak.py 18 512 > $OUT/soak.txt 2>&1; echo "soak rc=$?"; tail -n 2 $OUT/soak.txt
timeout -k 10 300 python3 tools/fuzz_gpu.py 500 878 > $OUT/fuzz_default.txt 2>&1; echo "fuzz default rc=$?"; tail -n 1 $OUT/fuzz_default.txt
SNAPPY_HIP_LDS_WAVES=0 timeout -k 10 300 python3 tools/fuzz_gpu.py 500 879 > $OUT/fuzz_global_table_cached.txt 2>&1; echo "fuzz global-table (cached) rc=$?"; tail -n 1 $OUT/fuzz_global_table_cached.txt
SNAPPY_HIP_LDS_WAVES=5 SNAPPY_HIP_GT_WAVES=11 SNAPPY_HIP_HYBRID_MIN_BLOCKS=1 timeout -k 10 300 python3 tools/fuzz_gpu.py 500 880 > $OUT/fuzz_tiny_hybrid.txt 2>&1; echo "fuzz tiny hybrid rc=$?"; tail -n 1 $OUT/fuzz_tiny_hybrid.txt
SNAPPY_HIP_LDS_WAVES=0 SNAPPY_HIP_K1_STREAM=1 timeout -k 10 300 python3 tools/fuzz_gpu.py 300 881 > $OUT/fuzz_global_table_cached_bulk.txt 2>&1; echo "fuzz global-table (cached, bulk form) rc=$?"; tail -n 1 $OUT/fuzz_global_table_cached_bulk.txt
SNAPPY_HIP_COMPRESS_VARIANT=1 timeout -k 10 300 python3 tools/fuzz_gpu.py 300 882 > $OUT/fuzz_lds_table.txt 2>&1; echo "fuzz LDS-table kernel alone rc=$?"; tail -n 1 $OUT/fuzz_lds_table.txt
timeout -k 10 400 python3 tools/dropin_stress.py 40 3 > $OUT/dropin_stress.txt 2>&1; echo "dropin stress rc=$?"; tail -n 2 $OUT/dropin_stress.txt
