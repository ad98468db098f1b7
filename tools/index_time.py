#!/usr/bin/env python3
"""Time the size-chain walk (index_streams) of one compressed Silesia-mix container, alone.  Usage: python tools/index_time.py MiB"""
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
d_stream = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device="cuda")
shb.compress_blocks(d_in, n, ws); shb.compact(n, ws, d_stream)
slen = int(ws.stream_len.item()); nb = shb.num_blocks(n, 32768)
hdr = len(shb.write_header(n, 32768))
boff = torch.zeros(nb, dtype=torch.int64, device="cuda"); res = torch.zeros(2, dtype=torch.int32, device="cuda")
descs = shb.make_stream_descs([dict(stream=d_stream, stream_len=slen, block_offsets=boff, result=res, total_len=n, block_size=32768,
                                    header_len=hdr, num_blocks=nb)])
for waves in sys.argv[2:] or ["0", "1"]:
    os.environ["SNAPPY_HIP_INDEX_READERS"] = waves
    ts = []
    for _ in range(4):
        boff.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); shb.index_streams(descs, 1); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ok = torch.equal(boff, ws.offsets[:nb]) and res.cpu().tolist() == [0, nb]
    print(f"index readers {waves:>2s}: " + " ".join(f"{t:6.2f}" for t in ts) + f" ms   offsets_ok={ok}", flush=True)
