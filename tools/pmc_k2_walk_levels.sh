#!/bin/bash
# Instruction counts of K2 against the doubling levels in front of its chain walk (SNAPPY_K2_WALK_LEVELS = L: the scalar walk
# visits every 2^L-th element): two rocprofv3 --pmc passes per build, 2 GiB container.   bash tools/pmc_k2_walk_levels.sh 1 2 4
ROOT=$PWD
for L in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSNAPPY_K2_WALK_LEVELS=$L pim-compression_amd/csrc/snappy_hip.hip -o pim-compression_amd/libk2walk_$L.so || exit 1
  OUT=$ROOT/gpurun_out/pmc_k2_walk_$L
  mkdir -p $OUT
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    ( cd /tmp && export TMPDIR=/tmp && SNAPPY_PROF_LIB=$ROOT/pim-compression_amd/libk2walk_$L.so timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/prof_once.py 2048 2 > $OUT/p$i.log 2>&1 ) || echo "pass $i failed"
  done
  echo "== SNAPPY_K2_WALK_LEVELS=$L"
  python3 - <<PY
import collections, csv, glob
agg=collections.defaultdict(list); dur=[]
for f in glob.glob('$OUT/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decompress' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$OUT/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decompress' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
dur.sort(); print("median ms under the profiler", dur[len(dur)//2] if dur else None)
for c,v in sorted(agg.items()): print(f"   {c:24s} {max(v):18.0f}")
PY
  rm -f pim-compression_amd/libk2walk_$L.so
done
