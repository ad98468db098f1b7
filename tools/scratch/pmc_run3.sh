set -e
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
export SNAPPY_HIP_LDS_WAVES=0 SNAPPY_HIP_K1_AHEAD=64 SNAPPY_HIP_K1_FORM=2
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_b64/p$i -- python3 $ROOT/tools/prof_once.py 2048 2 > $ROOT/gpurun_out/pmc_b64_p$i.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_b64 > $ROOT/gpurun_out/pmc_b64_summary.txt
grep -A20 compress_blocks_global $ROOT/gpurun_out/pmc_b64_summary.txt
