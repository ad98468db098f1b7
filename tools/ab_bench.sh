#!/bin/bash
# A/B two builds of libsnappy_hip.so on ONE box (boxes differ by 1-2 %): bash tools/ab_bench.sh old.so new.so [reps]
# Both files must lie inside the repository snapshot (e.g. under tools/ab/, which is git-ignored).
set -e
OLD=$1; NEW=$2; REPS=${3:-2}
mkdir -p gpurun_out
for rep in $(seq 1 $REPS); do
  for v in OLD NEW; do
    cp "${!v}" pim-compression_amd/libsnappy_hip.so
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    python - "$v" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "K1 ms", r["avg_launch_ms"], "K2 ms", r["decompress_kernel"]["avg_launch_ms"],
      "bit-exact", d["roundtrip_bit_exact"], flush=True)
PY
  done
done
