#!/usr/bin/env python3
"""Repeat K1 on one container and print time + the share of blocks the LDS-table wavefronts took (placement of the two
co-running kernels varies from launch to launch).  Usage: python tools/lds_share.py MiB reps ["ENV=..,ENV=.." ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import silesia_mix
import snappy_hip_binding as shb
mib, reps = int(sys.argv[1]), int(sys.argv[2])
n = mib << 20
xs = np.frombuffer(open(os.path.join(ROOT, "tests/golden/xml.snappy"), "rb").read(), dtype=np.uint8).copy()
st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
ws = shb.CompressWorkspace(n, 32768)
nb = shb.num_blocks(n, 32768)
for cfg in sys.argv[3:] or [""]:
    kv = dict(x.split("=") for x in cfg.split(",") if "=" in x)
    for k, v in kv.items(): os.environ[k] = v
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); shb.compress_blocks(d_in, n, ws); e1.record(); torch.cuda.synchronize()
        out.append((e0.elapsed_time(e1), ws.lds_form_blocks() / nb))
    print(f"{cfg or 'default':50s} " + "  ".join(f"{t:6.2f}ms/{100*s:4.1f}%" for t, s in out), flush=True)
    for k in kv: os.environ.pop(k, None)
