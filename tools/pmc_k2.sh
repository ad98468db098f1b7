set -e
ROOT=$PWD
mkdir -p $ROOT/gpurun_out/pmc_k2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_k2/p$i -- python3 $ROOT/tools/prof_once.py 2048 2 > $ROOT/gpurun_out/pmc_k2/p$i.log 2>&1
done
python3 - <<'PY'
import collections, csv, glob
agg=collections.defaultdict(list)
for f in glob.glob('/root/repo/gpurun_out/pmc_k2/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decompress' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c,v in sorted(agg.items()): print(f"{c:24s} {sum(v)/len(v):16.0f}")
PY
