#!/bin/bash
# Where does K1's time go?  Builds the library four times -- as shipped, without emission (-DK1X_NO_EMIT), without settling
# doubtful lanes / long matches (-DK1X_NO_STOPS), without both -- and takes instruction counts + kernel time of the
# global-table form alone on a 2 GiB container.  The experimental builds produce WRONG bytes on purpose; they exist only
# inside this script.  Result (profiles/r02_k1_timing_experiments.txt): a third of the scalar instructions gone, same time.
for v in "" "-DK1X_NO_EMIT" "-DK1X_NO_STOPS" "-DK1X_NO_EMIT -DK1X_NO_STOPS"; do
  t=$(echo "$v" | tr -d ' -')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v pim-compression_amd/csrc/snappy_hip.hip -o pim-compression_amd/libk1x_$t.so
done
set -e
ROOT=$PWD
for l in libk1x_.so libk1x_DK1X_NO_EMIT.so libk1x_DK1X_NO_STOPS.so libk1x_DK1X_NO_EMITDK1X_NO_STOPS.so; do
  mkdir -p $ROOT/gpurun_out/k1x/$l
  ( cd /tmp && export TMPDIR=/tmp && SNAPPY_HIP_LDS_WAVES=0 SNAPPY_PROF_LIB=$ROOT/pim-compression_amd/$l timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $ROOT/gpurun_out/k1x/$l -- python3 $ROOT/tools/prof_once.py 2048 2 > $ROOT/gpurun_out/k1x/$l/log 2>&1 )
  echo "== $l"
  python3 - <<PY
import collections, csv, glob
agg=collections.defaultdict(list); dur=[]
for f in glob.glob('$ROOT/gpurun_out/k1x/$l/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'global_table' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$ROOT/gpurun_out/k1x/$l/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'global_table' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
dur.sort(); print("   median ms", dur[len(dur)//2] if dur else None)
for c,v in sorted(agg.items()): print(f"   {c:24s} {sum(v)/len(v):16.0f}")
PY
done
