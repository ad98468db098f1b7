set -e
ROOT=$PWD
timeout -k 10 500 python -m pytest tests -x -q -m gpu > gpurun_out/r01k_gpu_tests.log 2>&1
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r01k_bench.json 2> gpurun_out/r01k_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r01k_prof -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/r01k_bench_under_rocprof.json 2> $ROOT/gpurun_out/r01k_prof.err
cd $ROOT
if [ -n "$WITH_K1_PMC" ]; then
SNAPPY_HIP_LDS_WAVES=0 bash tools/pmc_k1.sh gpurun_out/r01k_pmc_global > gpurun_out/r01k_pmc_global.txt 2>&1
SNAPPY_HIP_COMPRESS_VARIANT=1 bash tools/pmc_k1.sh gpurun_out/r01k_pmc_lds > gpurun_out/r01k_pmc_lds.txt 2>&1
fi
bash tools/pmc_k2.sh > gpurun_out/r01k_pmc_k2.txt 2>&1
tail -2 gpurun_out/r01k_gpu_tests.log; cat gpurun_out/r01k_bench.json
