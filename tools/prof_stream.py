#!/usr/bin/env python3
"""Lap-timer phase profile of K1's stream form (snappy_k1_stream.hpp).  Builds pim-compression_amd/libsnappy_hip_prof.so with
-DSNAPPY_PROF (s_memtime probes; not a product build) and runs one container through it.
Usage: python tools/prof_stream.py MiB "ENV=..,ENV=.." ..."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
LIB = os.path.join(ROOT, "pim-compression_amd", "libsnappy_hip_prof.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSNAPPY_PROF",
                       os.path.join(ROOT, "pim-compression_amd", "csrc", "snappy_hip.hip"), "-o", LIB])
import numpy as np, torch
import silesia_mix
import snappy_hip_binding as shb
shb.LIB_PATH = LIB
L = shb.lib()
L.snappy_hip_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
mib = int(sys.argv[1]); n = mib << 20
xs = np.frombuffer(open(os.path.join(ROOT, "tests/golden/xml.snappy"), "rb").read(), dtype=np.uint8).copy()
st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
ws = shb.CompressWorkspace(n, 32768)
names = ["prime", "wait_loads", "finalize", "walk", "settle", "masks_commit", "issue_next", "emit", "dup_analysis", "long_copies", "-", "bulk", "glue"]
for cfg in sys.argv[2:]:
    kv = dict(x.split("=") for x in cfg.split(","))
    for k, v in kv.items(): os.environ[k] = v
    shb.compress_blocks(d_in, n, ws); torch.cuda.synchronize()
    L.snappy_hip_debug_prof(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); shb.compress_blocks(d_in, n, ws); e1.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    L.snappy_hip_debug_prof(out, 0)
    t, c = list(out)[:16], list(out)[16:]
    total = sum(t[:13])
    nb = n // 32768
    windows = max(c[2], 1)
    print(f"== {cfg}: {e0.elapsed_time(e1):.2f} ms; cycles/block {total / nb:.0f}; windows taken by the stream form {c[2]} of {n // 64}"
          f" ({100.0 * c[2] * 64 / n:.1f} %); stream runs {c[0]}, bulk runs {c[11]}")
    for i, name in enumerate(names):
        print(f"   {name:14s} {100.0 * t[i] / max(total, 1):5.1f}%   laps {c[i]:10d}  {t[i] / max(c[i], 1):8.0f} cycles each   {t[i] / windows:8.0f} per window", flush=True)
    for k in kv: os.environ.pop(k, None)
