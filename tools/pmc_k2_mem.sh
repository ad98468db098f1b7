#!/bin/bash
# PMC of K2 (2 GiB container): vector-memory / LDS instruction counts, wait states, LDS stalls (the TA_* / TCP_* sets abort rocprofv3 on this image).
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/${1:-pmc_k2_mem}
mkdir -p $OUT
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/prof_once.py 2048 2 > $OUT/p$i.log 2>&1 ) || echo "pass $i failed"
done
python3 - <<PY
import collections, csv, glob
agg=collections.defaultdict(list); dur=[]
for f in glob.glob('$OUT/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decompress' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$OUT/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decompress' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
dur.sort(); print("median ms", dur[len(dur)//2] if dur else None)
for c,v in sorted(agg.items()): print(f"{c:40s} {sum(v)/len(v):18.0f}")
PY
