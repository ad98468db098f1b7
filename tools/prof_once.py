#!/usr/bin/env python3
"""Run K1 + compact + index + K2 a few times on one Silesia-mix container (for rocprofv3 --pmc passes).
Usage: python3 tools/prof_once.py [MiB] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import silesia_mix
import snappy_hip_binding as shb

if os.environ.get("SNAPPY_PROF_LIB"):            # an experimental build of the library (timing experiments)
    shb.LIB_PATH = os.environ["SNAPPY_PROF_LIB"]

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = mib << 20
with open(os.path.join(ROOT, "tests", "golden", "xml.snappy"), "rb") as f:
    xs = np.frombuffer(f.read(), dtype=np.uint8).copy()
st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
ws = shb.CompressWorkspace(n, 32768)
d_stream = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device="cuda")
nb = shb.num_blocks(n, 32768)
status = torch.empty(nb, dtype=torch.int32, device="cuda")
out = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
for _ in range(reps):
    shb.compress_blocks(d_in, n, ws)
    shb.compact(n, ws, d_stream)
    slen = int(ws.stream_len.item())
    shb.decompress_blocks(d_stream, slen, ws.offsets[:nb].contiguous(), n, 32768, out, status)
torch.cuda.synchronize()
print("ok", torch.equal(out[:n], d_in[:n]), slen, "lds_form_blocks", ws.lds_form_blocks(), "of", nb)
