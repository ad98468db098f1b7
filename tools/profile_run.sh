#!/bin/bash
# The round's evidence run on the GPU box: bash tools/profile_run.sh <tag> [part ...]   (tag e.g. r04; parts: bench workloads
# one_container pmc_k1 pmc_k2; default: all).  Everything lands under gpurun_out/<tag>_final/; the summaries are copied to
# profiles/<tag>_* afterwards (profiles/README.md names each file's command).
set -e
ROOT=$PWD
TAG=${1:-r04}
shift || true
PARTS=${@:-bench workloads one_container pmc_k1 pmc_k2}
OUT=$ROOT/gpurun_out/${TAG}_final
mkdir -p $OUT
for PART in $PARTS; do
  case $PART in
  bench)
    timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
    ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_prof -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-preverify --no-stream-alone > $OUT/bench_under_rocprof.json 2> $OUT/bench_prof.err )
    tail -c 400 $OUT/bench.json ;;
  workloads)
    for w in dickens_like mozilla_like spamfile_like; do
      timeout -k 10 300 python bench.py --workload $w --steps 50 --warmup 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
      ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python3 $ROOT/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-preverify --no-stream-alone > $OUT/bench_${w}_under_rocprof.json 2> $OUT/prof_$w.err )
      python3 -c "import json;d=json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1]);print('$w',d['value'],d['ms_per_step'],d['compress_kernel_GBps'],d['decompress_kernel_GBps'])"
    done ;;
  one_container)      # what one rank of the N = 2 / 4 / 8 strong-scaling runs does, on one GPU
    for c in 8 4 2 1; do
      timeout -k 10 300 python3 bench.py --containers $c --steps 6 --warmup 2 --no-cpu-baseline --no-preverify --no-stream-alone 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('containers $c: ms/step %.3f  value %.2f GB/s  K1 %.3f ms  K2 %.3f ms' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['decompress_kernel']['avg_launch_ms']))"
    done | tee $OUT/one_container_step.txt ;;
  pmc_k1)             # rocprofv3 --pmc serialises kernels: each K1 kernel alone (five passes each: instruction mix, waits, TCC requests, FETCH_SIZE, WRITE_SIZE)
    SNAPPY_HIP_LDS_WAVES=0 bash tools/pmc_k1.sh gpurun_out/${TAG}_final/pmc_k1_global_table > $OUT/pmc_k1_global_table.txt 2>&1
    SNAPPY_HIP_COMPRESS_VARIANT=1 bash tools/pmc_k1.sh gpurun_out/${TAG}_final/pmc_k1_lds_table > $OUT/pmc_k1_lds_table.txt 2>&1
    tail -n 24 $OUT/pmc_k1_global_table.txt ;;
  pmc_k2)
    bash tools/pmc_k2_walk_levels.sh 2 > $OUT/pmc_k2.txt 2>&1
    cat $OUT/pmc_k2.txt ;;
  esac
done
