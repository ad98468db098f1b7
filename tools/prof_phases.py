#!/usr/bin/env python3
"""Cycle-counter phase profile of the bulk K1 (needs `python tools/make_probe_build.py --k1` first: an instrumented
libsnappy_hip_prof.so with s_memtime probes; not part of the product).  Usage: python tools/prof_phases.py MiB "ENV=.." ..."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import silesia_mix
import snappy_hip_binding as shb
shb.LIB_PATH = os.path.join(ROOT, "pim-compression_amd", "libsnappy_hip_prof.so")
L = shb.lib()
L.snappy_hip_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
mib = int(sys.argv[1]); n = mib << 20
xs = np.frombuffer(open(os.path.join(ROOT, "tests/golden/xml.snappy"), "rb").read(), dtype=np.uint8).copy()
st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
ws = shb.CompressWorkspace(n, 32768)
names = ["t_total", "blocks", "t_gather_issue", "t_gather_wait", "n_gather", "t_walk", "n_seg", "t_commit_emit", "t_single", "n_single", "t_win", "n_win", "t_drain", "t_resolve", "n_resolve", "_15", "_16", "t_first_literal", "n_first_literal"]
for cfg in sys.argv[2:]:
    kv = dict(x.split("=") for x in cfg.split(","))
    for k, v in kv.items(): os.environ[k] = v
    shb.compress_blocks(d_in, n, ws); torch.cuda.synchronize()
    L.snappy_hip_debug_prof(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); shb.compress_blocks(d_in, n, ws); e1.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    L.snappy_hip_debug_prof(out, 0)
    p = dict(zip(names, list(out)))
    tot = max(p["t_total"], 1)
    print(f"== {cfg}: {e0.elapsed_time(e1):.2f} ms; blocks {p['blocks']}; cycles/block {tot/max(p['blocks'],1):.0f}")
    for k in ("t_drain", "t_gather_issue", "t_gather_wait", "t_walk", "t_resolve", "t_commit_emit", "t_first_literal", "t_single", "t_win"):
        cnt = {"t_drain": p["n_gather"], "t_gather_issue": p["n_gather"], "t_gather_wait": p["n_gather"], "t_walk": p["n_seg"], "t_resolve": p["n_resolve"], "t_commit_emit": p["n_seg"], "t_first_literal": p["n_first_literal"], "t_single": p["n_single"], "t_win": p["n_win"]}[k]
        print(f"   {k:16s} {100.0*p[k]/tot:5.1f}%   n={cnt:10d}  {p[k]/max(cnt,1):8.0f} cycles each")
    print("   (t_walk includes t_resolve; t_commit_emit includes t_first_literal)")
    rest = tot - sum(p[k] for k in ("t_drain", "t_gather_issue", "t_gather_wait", "t_walk", "t_commit_emit", "t_single", "t_win"))
    print(f"   {'rest':16s} {100.0*rest/tot:5.1f}%", flush=True)
    for k in kv: os.environ.pop(k, None)
