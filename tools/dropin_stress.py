#!/usr/bin/env python3
"""Stress of the drop-in pair's overlapped form (GPU): many calls in one process -- the streams, page-locked scratch and
hardware queues are reused from call to call -- with random sizes, block sizes, chunkings and shard counts; every stream is
compared with the oracle's and every round trip with the input.  Usage: python tools/dropin_stress.py [calls] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import datagen
import oracle_lib as oracle
import snappy_hip_binding as shb

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = np.random.default_rng(seed)
with open(os.path.join(ROOT, "tests", "golden", "world192.txt"), "rb") as f:
    text = f.read()
pool = datagen.text_random_interleave(text, 96 << 20, seed=seed) + datagen.records(32 << 20, seed=seed + 1)
os.environ["SNAPPY_HIP_OVERSUBSCRIBE"] = "1"
bad = 0
for it in range(calls):
    n = int(r.integers(1, 120 << 20))
    lo = int(r.integers(0, len(pool) - n))
    data = pool[lo:lo + n]
    bs = int(r.choice([32768, 32768, 65535, 4096, 1000, 16384]))
    chunk = int(r.choice([0, 16, 64, 256, 1024, 4096]))
    shards = int(r.choice([1, 1, 2, 3]))
    os.environ["SNAPPY_HIP_PIPELINE_BLOCKS"] = str(chunk)
    os.environ["SNAPPY_HIP_NUM_GPUS"] = str(shards)
    ref = oracle.compress(data, bs, threads=16)
    st, stream, _ = shb.compress_host(data, bs)
    ok_c = st == 0 and stream == ref
    st, plain, _ = shb.decompress_host(ref)
    ok_d = st == 0 and plain == data
    print(f"call {it}: n={n} bs={bs} chunk={chunk} shards={shards} compress={ok_c} decompress={ok_d}", flush=True)
    bad += (not ok_c) + (not ok_d)
print("drop-in stress done, failures:", bad)
sys.exit(1 if bad else 0)
