#!/bin/bash
# Round-3 evidence run on the GPU box (tools/r03_profile_run.sh [part]): everything lands under gpurun_out/r03_final/ and the
# summaries are copied to profiles/r03_* afterwards (see profiles/README.md).  Parts: bench | workloads | pmc_k1 | pmc_k2
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/r03_final
mkdir -p $OUT
PART=${1:-all}
if [ "$PART" = all ] || [ "$PART" = bench ]; then
  timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_prof -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-preverify --no-stream-alone > $OUT/bench_under_rocprof.json 2> $OUT/bench_prof.err )
  tail -c 600 $OUT/bench.json
fi
if [ "$PART" = all ] || [ "$PART" = workloads ]; then
  for w in dickens_like mozilla_like spamfile_like; do
    timeout -k 10 300 python bench.py --workload $w --steps 50 --warmup 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
    ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python3 $ROOT/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-preverify --no-stream-alone > $OUT/bench_${w}_under_rocprof.json 2> $OUT/prof_$w.err )
    python3 -c "import json;d=json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1]);print('$w',d['value'],d['ms_per_step'],d['compress_kernel_GBps'],d['decompress_kernel_GBps'])"
  done
fi
if [ "$PART" = all ] || [ "$PART" = pmc_k1 ]; then
  SNAPPY_HIP_LDS_WAVES=0 bash tools/pmc_k1.sh gpurun_out/r03_final/pmc_k1_global_table > $OUT/pmc_k1_global_table.txt 2>&1
  SNAPPY_HIP_COMPRESS_VARIANT=1 bash tools/pmc_k1.sh gpurun_out/r03_final/pmc_k1_lds_table > $OUT/pmc_k1_lds_table.txt 2>&1
  tail -n 30 $OUT/pmc_k1_lds_table.txt
fi
if [ "$PART" = all ] || [ "$PART" = pmc_k2 ]; then
  bash tools/pmc_k2.sh > $OUT/pmc_k2.txt 2>&1
  cat $OUT/pmc_k2.txt
fi
