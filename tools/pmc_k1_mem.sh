#!/bin/bash
# L2 <-> fabric requests of the global-table kernel alone (SNAPPY_HIP_LDS_WAVES=0), tagged 4-byte slots against the
# content-carrying 32-byte slots; one rocprofv3 run per counter set.  usage: bash tools/pmc_k1_mem.sh OUTDIR [MiB]
OUT=${1:-gpurun_out/pmc_k1_mem}
MIB=${2:-2048}
ROOT=$PWD
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
export SNAPPY_HIP_LDS_WAVES=0
rocprofv3 --list-avail > "$ROOT/$OUT/avail.txt" 2>&1
for wide in 0 1; do
  export SNAPPY_HIP_GT_WIDE=$wide
  i=0
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/w$wide/p$i" -- python3 "$ROOT/tools/prof_once.py" "$MIB" 2 > "$ROOT/$OUT/w$wide.p$i.log" 2>&1 || echo "set $i failed (wide=$wide)"
  done
  echo "#### SNAPPY_HIP_GT_WIDE=$wide"
  python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT/w$wide"
done 2>&1 | tee "$ROOT/$OUT/summary.txt"
