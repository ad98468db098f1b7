#!/bin/bash
# PMC of K2 with / without the per-window batch (SNAPPY_HIP_K2_BATCH=1/0): instruction mix and wait states, 2 GiB container.
set -e
ROOT=$PWD
for b in 1 0; do
  mkdir -p $ROOT/gpurun_out/pmc_k2_b$b
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    ( cd /tmp && export TMPDIR=/tmp && SNAPPY_HIP_K2_BATCH=$b timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_k2_b$b/p$i -- python3 $ROOT/tools/prof_once.py 2048 2 > $ROOT/gpurun_out/pmc_k2_b$b/p$i.log 2>&1 )
  done
  echo "== SNAPPY_HIP_K2_BATCH=$b"
  python3 - <<PY
import collections, csv, glob
agg=collections.defaultdict(list); dur=[]
for f in glob.glob('$ROOT/gpurun_out/pmc_k2_b$b/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decompress' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$ROOT/gpurun_out/pmc_k2_b$b/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decompress' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
dur.sort(); print("   median ms", dur[len(dur)//2] if dur else None)
for c,v in sorted(agg.items()): print(f"   {c:24s} {sum(v)/len(v):16.0f}")
PY
done
