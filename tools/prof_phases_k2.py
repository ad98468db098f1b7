#!/usr/bin/env python3
"""Cycle-counter phase profile of K2 (needs `python tools/make_probe_build.py --k2` first: an instrumented
libsnappy_hip_prof.so with s_memtime probes; not part of the product).  Usage: python tools/prof_phases_k2.py MiB"""
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
d_stream = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device="cuda")
shb.compress_blocks(d_in, n, ws); shb.compact(n, ws, d_stream)
slen = int(ws.stream_len.item()); nb = shb.num_blocks(n, 32768)
boff = ws.offsets[:nb].contiguous()
status = torch.empty(nb, dtype=torch.int32, device="cuda"); out = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
shb.decompress_blocks(d_stream, slen, boff, n, 32768, out, status); torch.cuda.synchronize()
L.snappy_hip_debug_prof(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); shb.decompress_blocks(d_stream, slen, boff, n, 32768, out, status); e1.record(); torch.cuda.synchronize()
o = (ctypes.c_ulonglong * 32)(); L.snappy_hip_debug_prof(o, 0)
tot = max(o[0], 1)
print(f"K2 {e0.elapsed_time(e1):.2f} ms, blocks {o[1]}, cycles/block {tot/max(o[1],1):.0f}, ok={torch.equal(out[:n], d_in[:n])}")
for name, t, c in (("window+predecode", o[2], o[3]), ("fast loop (asm)", o[11], o[3]), ("literal (C++ path)", o[4], o[5]), ("copy (C++ path)", o[6], o[7]), ("long literal", o[9], o[10])):
    print(f"   {name:18s} {100.0*t/tot:5.1f}%  n={c:11d}  {t/max(c,1):7.0f} cycles each")
print(f"   overlapping copies {o[8]} of {o[7]}")
rest = tot - o[2] - o[4] - o[6] - o[9] - o[11]
print(f"   rest {100.0*rest/tot:5.1f}%")
