#!/bin/bash
# A/B of two source trees of the library on ONE box for the small and mid-size workloads (bench.py --workload ..., K1 / K2 launch
# times from the JSON line) and the 8 GiB batch: tools/ab_dirs_workloads.sh <csrc dir A> <csrc dir B>.  Run it on the GPU box
# only: it overwrites pim-compression_amd/libsnappy_hip.so of the (scratch) copy it runs in.
ROOT=$PWD
i=0
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -I$ROOT/include $d/snappy_hip.hip -o pim-compression_amd/libab_$i.so || exit 1
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for d in "$@"; do
    cp pim-compression_amd/libab_$i.so pim-compression_amd/libsnappy_hip.so
    for w in dickens_like mozilla_like spamfile_like silesia_mix; do
      echo -n "[$d] $w "
      timeout -k 10 300 python3 bench.py --workload $w --steps $([ $w = silesia_mix ] && echo 4 || echo 40) --warmup $([ $w = silesia_mix ] && echo 1 || echo 5) --no-cpu-baseline --no-preverify --no-stream-alone 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.3f  K1 %.3f ms  K2 %.3f ms ok %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['decompress_kernel']['avg_launch_ms'], d['roundtrip_bit_exact']))"
    done
    i=$((i+1))
  done
done
rm -f pim-compression_amd/libab_*.so
