#!/bin/bash
# BASELINE configs[2]/[3] stand-ins (one file each, a few hundred to a few thousand blocks): which K1 launch shape is quickest
# when the blocks do not fill the chip?  default = global-table kernel alone below SNAPPY_HIP_HYBRID_MIN_BLOCKS (4096).
for w in dickens_like mozilla_like spamfile_like; do
  for e in "X=0" "SNAPPY_HIP_COMPRESS_VARIANT=1" "SNAPPY_HIP_HYBRID_MIN_BLOCKS=1" "SNAPPY_HIP_HYBRID_MIN_BLOCKS=1 SNAPPY_HIP_LDS_WAVES=1024"; do
    echo -n "$w [$e] "
    env $e timeout -k 10 200 python3 bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-preverify --no-stream-alone 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.3f  e2e %.2f GB/s  K1 %.2f GB/s (%.3f ms)  K2 %.2f GB/s (%.3f ms) ok %s' % (d['ms_per_step'], d['value'], d['compress_kernel_GBps'], d['roofline']['avg_launch_ms'], d['decompress_kernel_GBps'], d['roofline']['decompress_kernel']['avg_launch_ms'], d['roundtrip_bit_exact']))"
  done
done
