#!/bin/bash
# PMC passes for K1 on the GPU box: instruction mix, wait states and L2<->fabric requests of the compress kernel,
# one rocprofv3 run per counter set (never combined with other trace domains).  Environment selects the K1
# configuration, e.g.
#   SNAPPY_HIP_LDS_WAVES=0 SNAPPY_HIP_K1_FORM=2 SNAPPY_HIP_K1_FILTER=1 bash tools/pmc_k1.sh gpurun_out/pmc_bulk
# (--pmc serialises kernels, so measure the global-table and the LDS-table form separately: SNAPPY_HIP_LDS_WAVES=0
# or SNAPPY_HIP_COMPRESS_VARIANT=1).  Prints the per-kernel summary (tools/pmc_summary.py).
set -e
OUT=${1:-gpurun_out/pmc_k1}
MIB=${2:-2048}
ROOT=$PWD
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/p$i" -- python3 "$ROOT/tools/prof_once.py" "$MIB" 2 > "$ROOT/$OUT/p$i.log" 2>&1
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" | tee "$ROOT/$OUT/summary.txt"
