#!/bin/bash
# Instruction mix of the free-table K1 (csrc/ablation/k1_oracle_table.hpp) beside the product's cached global-table kernel:
# two rocprofv3 --pmc passes over tools/gate_b_ceiling.py <MiB> pmc.   bash tools/gate_b_pmc.sh gpurun_out/r04_gate_b_pmc [MiB]
set -e
OUT=${1:-gpurun_out/r04_gate_b_pmc}
MIB=${2:-2048}
ROOT=$PWD
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$ROOT/$OUT/p$i" -- python3 "$ROOT/tools/gate_b_ceiling.py" "$MIB" pmc > "$ROOT/$OUT/p$i.log" 2>&1
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" | tee "$ROOT/$OUT/summary.txt"
