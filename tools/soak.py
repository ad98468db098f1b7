#!/usr/bin/env python3
"""Soak test (GPU): many compress -> decompress round trips on differently seeded Silesia-mix containers and
LZ-structured inputs, every one checked bit for bit (round trip on device, and the compressed stream against the
multi-threaded oracle).  Looks for rare races in the co-running / persistent kernels.
Usage: python tools/soak.py [iterations] [MiB]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import datagen
import oracle_lib as oracle
import silesia_mix
import snappy_hip_binding as shb

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 512
with open(os.path.join(ROOT, "tests", "golden", "xml.snappy"), "rb") as f:
    xs = np.frombuffer(f.read(), dtype=np.uint8).copy()
st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
xml = d_xml.cpu().numpy()
bad = 0
for it in range(iters):
    n = (mib << 20) - 12345 * it
    bs = [32768, 32768, 65535, 4096, 32768, 10000][it % 6]
    unit = silesia_mix.build_unit(xml, seed=100 + it)
    d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
    d_stream = shb.compress_resident(d_in, bs, n=n)
    st, d_out = shb.decompress_resident(d_stream)
    ok_rt = st == 0 and torch.equal(d_out[:n], d_in[:n])
    ref = oracle.compress(d_in[:n].cpu().numpy(), bs, threads=32)
    ok_or = hashlib.sha256(d_stream.cpu().numpy().tobytes()).digest() == hashlib.sha256(ref).digest()
    print(f"iter {it}: n={n} bs={bs} roundtrip={ok_rt} oracle={ok_or}", flush=True)
    bad += (not ok_rt) + (not ok_or)
    del d_in, d_stream, d_out
for seed in range(40):
    data = datagen.lz_structured(3_000_000 + 7919 * seed, 500 + seed)
    t = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
    t[:len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    bs = [32768, 65535, 1000][seed % 3]
    d_stream = shb.compress_resident(t, bs, n=len(data))
    ref = oracle.compress(data, bs, threads=16)
    st, d_out = shb.decompress_resident(d_stream)
    ok = bytes(d_stream.cpu().numpy()) == ref and st == 0 and bytes(d_out.cpu().numpy()) == data
    if not ok:
        print("LZ seed", seed, "FAILED")
        bad += 1
print("soak done, failures:", bad)
sys.exit(1 if bad else 0)
