#!/usr/bin/env python3
"""Ablation harness (GPU): time K1 variants / occupancy knobs on one Silesia-mix container.
Usage: python tools/exp_variants.py [MiB] ["variant:extra_lds" ...]"""
import hashlib
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


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    configs = sys.argv[2:] or ["0:0", "1:0", "2:0"]
    n = mib << 20
    with open(os.path.join(ROOT, "tests", "golden", "xml.snappy"), "rb") as f:
        xs = np.frombuffer(f.read(), dtype=np.uint8).copy()
    st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
    assert st == 0
    unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
    d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
    bs = int(os.environ.get("EXP_BLOCK_SIZE", "32768"))
    ws = shb.CompressWorkspace(n, bs)
    d_stream = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device="cuda")
    ref = None
    for cfg in configs:
        kv = dict(x.split("=") for x in cfg.split(",") if "=" in x)
        if not kv:
            v, e = cfg.split(":")
            kv = {"SNAPPY_HIP_COMPRESS_VARIANT": v}
        for k, val in kv.items():
            os.environ[k] = val
        times = []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            shb.compress_blocks(d_in, n, ws)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        shb.compact(n, ws, d_stream)
        slen = int(ws.stream_len.item())
        digest = hashlib.sha256(d_stream[:slen].cpu().numpy().tobytes()).hexdigest()[:16]
        if ref is None:
            ref = digest
        best = min(times[1:])
        print(f"{cfg:40s} best {best:8.3f} ms  {n / best / 1e6:8.2f} GB/s  stream {slen}  sha {digest} {'OK' if digest == ref else 'MISMATCH'}",
              flush=True)
        for k in kv:
            os.environ.pop(k, None)
    if bs != 32768:
        return
    # decompress timing for reference
    st, d_out = shb.decompress_resident(d_stream[:slen])
    total, bs, hdr = shb.parse_header(bytes(d_stream[:10].cpu().numpy()))
    nb = shb.num_blocks(total, bs)
    boff = ws.offsets[:nb].contiguous()
    status = torch.empty(nb, dtype=torch.int32, device="cuda")
    out = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    for dv in ("1",):                            # "0" (round 1's element loop) exists in the ablation build only
        out.zero_()
        times = []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            shb.decompress_blocks(d_stream, slen, boff, n, 32768, out, status)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        print(f"decompress batch={dv} best {min(times[1:]):8.3f} ms {n / min(times[1:]) / 1e6:8.2f} GB/s "
              f"ok={torch.equal(out[:n], d_in[:n])} bad_blocks={int((status != 0).sum())}")


if __name__ == "__main__":
    main()
