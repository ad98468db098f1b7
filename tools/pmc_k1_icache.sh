#!/bin/bash
# Instruction-cache behaviour of K1 (large kernels, wavefronts at different places of the code): SQC_ICACHE_* and SQ_IFETCH*,
# the product mix is not measurable (--pmc serialises kernels), so each kernel alone.  usage: bash tools/pmc_k1_icache.sh OUTDIR
OUT=${1:-gpurun_out/pmc_k1_icache}
ROOT=$PWD
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "SNAPPY_HIP_LDS_WAVES=0" "SNAPPY_HIP_COMPRESS_VARIANT=1" "SNAPPY_HIP_LDS_WAVES=0 SNAPPY_HIP_K1_STREAM=1"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  export $cfg
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/$tag/p$i" -- python3 "$ROOT/tools/prof_once.py" 2048 2 > "$ROOT/$OUT/$tag.p$i.log" 2>&1 || echo "set $i failed ($cfg)"
  done
  echo "#### $cfg"
  python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT/$tag"
  for v in $cfg; do unset ${v%%=*}; done
done 2>&1 | tee "$ROOT/$OUT/summary.txt"
