#!/bin/bash
# bench.py's launch groups: the decode of group g on a second stream underneath the compression of group g+1.  K1's default
# launch fills every CU's LDS (3 x 33 KiB + 20 x 3 KiB), so K2's wavefronts (1.5 KiB each) only find room once K1 drains;
# here K1 leaves room (SNAPPY_HIP_GT_WAVES = total K1 wavefronts) and K2 is capped to what fits (SNAPPY_HIP_K2_WAVES).
# One box, one process per point: "groups  K1 wavefronts  K2 wavefronts"
for cfg in "1 5888 8192" "4 4864 2048" "4 4352 3072" "2 4864 2048" "4 4864 1024" "8 4864 2048" "4 5376 1024"; do
  set -- $cfg
  echo "== groups $1 K1 waves $2 K2 waves $3"
  SNAPPY_HIP_GT_WAVES=$2 SNAPPY_HIP_K2_WAVES=$3 timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --groups $1 --no-cpu-baseline --no-preverify --no-stream-alone 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f GB/s  ms/step %.2f  K1 avg launch %.2f ms  K2 avg launch %.2f ms  ok %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['decompress_kernel']['avg_launch_ms'], d['roundtrip_bit_exact']))"
done
