#!/bin/bash
# `pre` of snappy_decompress_gpu with the blocks split over 1 / 2 / 8 shards (all on this box's one device): the host pre-scan of the size chain,
# serial (SNAPPY_HIP_HOST_WALK_THREADS=1) against parallel shares (csrc/host_chain.hpp).   bash tools/host_walk_pre.sh  -> profiles/r04_host_walk.txt
for shards in 1 2 8; do
  for t in 1 16; do
    echo "shards $shards host walk threads $t"
    DROPIN_BLOCKS=0,auto SNAPPY_HIP_NUM_GPUS=$shards SNAPPY_HIP_OVERSUBSCRIBE=1 SNAPPY_HIP_HOST_WALK_THREADS=$t timeout -k 10 300 python3 tools/dropin_rate.py 2048 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    d=json.loads(ln); p=d['decompress_phases_ms']
    print('   pipeline_blocks', d['pipeline_blocks'], 'decompress pre %.2f ms  copy_in+run+copy_out %.2f ms  wall %.2f ms' % (p['pre'], d['decompress_ms'], p['wall']))"
  done
done
