#!/usr/bin/env python3
"""Does K2 fit beside K1?  Times K1 alone, K2 alone, and both launched together on two streams (K2 optionally with a reduced
grid via SNAPPY_HIP_K2_WAVES).  Usage: python tools/corun_k1_k2.py [MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import silesia_mix
import snappy_hip_binding as shb
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = mib << 20
xs = np.frombuffer(open(os.path.join(ROOT, "tests/golden/xml.snappy"), "rb").read(), dtype=np.uint8).copy()
st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
ws = shb.CompressWorkspace(n, 32768)
ws2 = shb.CompressWorkspace(n, 32768)
d_stream = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device="cuda")
shb.compress_blocks(d_in, n, ws); shb.compact(n, ws, d_stream)
slen = int(ws.stream_len.item()); nb = shb.num_blocks(n, 32768)
boff = ws.offsets[:nb].clone()
status = torch.empty(nb, dtype=torch.int32, device="cuda"); out = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
s2 = torch.cuda.Stream()
def timed(fn):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
k1 = lambda: shb.compress_blocks(d_in, n, ws2)
k2 = lambda: shb.decompress_blocks(d_stream, slen, boff, n, 32768, out, status)
def both():
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        k2()
    k1()
    main.wait_stream(s2)
print(f"K1 alone {timed(k1):.2f} ms, K2 alone {timed(k2):.2f} ms")
for w in sys.argv[2:] or ["8192", "2304", "1024"]:
    os.environ["SNAPPY_HIP_K2_WAVES"] = w
    print(f"K2 grid {w:>5s}: K2 alone {timed(k2):.2f} ms, K1 || K2 {timed(both):.2f} ms", flush=True)
