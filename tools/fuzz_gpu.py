#!/usr/bin/env python3
"""GPU fuzz: many small and medium inputs of different structure, random block sizes and lengths; every compressed stream
must equal the oracle's byte for byte and decode back to the input.  Usage: python tools/fuzz_gpu.py [cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import datagen
import oracle_lib as oracle
import snappy_hip_binding as shb


def run(cases, seed, verbose=True):
    """Returns the number of failing cases."""
    rnd = random.Random(seed)
    with open(os.path.join(ROOT, "tests", "golden", "plrabn12.txt"), "rb") as f:
        text = f.read()
    with open(os.path.join(ROOT, "tests", "golden", "world192.txt"), "rb") as f:
        world = f.read()
    bad = 0
    for c in range(cases):
        kind = rnd.randrange(9)
        n = rnd.choice([rnd.randrange(1, 200), rnd.randrange(200, 70_000), rnd.randrange(70_000, 600_000)])
        seed_c = rnd.randrange(1 << 30)
        if kind == 0:
            data = datagen.lz_structured(n, seed_c)
        elif kind == 1:
            data = datagen.records(n, seed_c)
        elif kind == 2:
            data = datagen.text_random_interleave(text, n, seed_c, chunk=rnd.choice([300, 800, 3000]))
        elif kind == 3:
            data = datagen.periodic(n, rnd.randrange(1, 200), seed_c)
        elif kind == 4:
            data = datagen.low_entropy(n, rnd.choice([2, 3, 4, 16]), seed_c)
        elif kind == 5:
            o = rnd.randrange(0, max(1, len(world) - n))
            data = world[o:o + n]
        elif kind == 6:
            data = datagen.random_bytes(n, seed_c)
        elif kind == 7:
            data = (datagen.zeros(n // 2) + datagen.lz_structured(n - n // 2, seed_c))
        else:
            piece = datagen.lz_structured(rnd.randrange(5, 400), seed_c)
            data = (piece * (n // len(piece) + 1))[:n]
        bs = rnd.choice([rnd.randrange(16, 300), rnd.randrange(300, 5000), rnd.randrange(5000, 65536), 32768, 65535, 4096])
        t = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
        t[:len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        d_stream = shb.compress_resident(t, bs, n=len(data))
        ref = oracle.compress(data, bs, threads=4)
        st, d_out = shb.decompress_resident(d_stream)
        ok = bytes(d_stream.cpu().numpy()) == ref and st == 0 and bytes(d_out.cpu().numpy()) == data
        if not ok:
            print(f"case {c}: kind {kind} n {n} seed {seed_c} bs {bs} FAILED", flush=True)
            bad += 1
        elif verbose and c % 50 == 0:
            print(f"case {c} ok", flush=True)
    return bad


def run_element_streams(cases, seed, verbose=True):
    """Decoder only: random valid element streams (datagen.element_stream -- what no greedy compressor writes) must decode as
    the oracle decodes them.  Returns the number of failing cases."""
    rnd = random.Random(seed)
    bad = 0
    for c in range(cases):
        bs = rnd.choice([rnd.randrange(16, 300), rnd.randrange(300, 5000), rnd.randrange(5000, 65536), 32768, 65535, 4096])
        n = rnd.randrange(1, 20_000) if bs < 300 else rnd.randrange(1, 300_000)
        stream, plain = datagen.element_stream(n, bs, rnd.randrange(1 << 30), c % 4)
        st_ref, ref = oracle.decompress(stream)
        t = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
        st, d_out = shb.decompress_resident(t, stream_len=len(stream))
        ok = st_ref == 0 and ref == plain and st == 0 and bytes(d_out.cpu().numpy()) == plain
        if not ok:
            print(f"element stream case {c}: n {n} bs {bs} flavour {c % 4} FAILED", flush=True)
            bad += 1
        elif verbose and c % 50 == 0:
            print(f"element stream case {c} ok", flush=True)
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    failures = run(n_cases, seed)
    failures += run_element_streams(n_cases // 4, seed + 1)
    print("fuzz done, failures:", failures)
    sys.exit(1 if failures else 0)
