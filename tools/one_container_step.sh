#!/bin/bash
# What one rank of the N = 2 / 4 / 8 strong-scaling runs does, on one GPU: a step over 4 / 2 / 1 of the 8 x 1 GiB containers
# (the per-launch constant that the predicted scaling efficiency comes from).
for c in 8 4 2 1; do
  timeout -k 10 300 python3 bench.py --containers $c --steps 6 --warmup 2 --no-cpu-baseline --no-preverify --no-stream-alone 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('containers $c: ms/step %.3f  value %.2f GB/s  K1 %.3f ms  K2 %.3f ms' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['decompress_kernel']['avg_launch_ms']))"
done
