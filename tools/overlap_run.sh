set -e
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 2 > gpurun_out/ov_base.json 2> gpurun_out/ov_base.err
for cap in 3072 4096 6144 8192; do
  SNAPPY_BENCH_OVERLAP=1 SNAPPY_BENCH_K2_CAP=$cap timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/ov_$cap.json 2> gpurun_out/ov_$cap.err
done
for f in gpurun_out/ov_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d.get('kernels'))
"; done
