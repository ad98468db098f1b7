#!/usr/bin/env python3
"""K1 rate on a resident 1 GiB container: one launch, against launches of `chunk` blocks back to back on one stream,
against the same launches alternating between two streams (two hash-table scratches) -- what bounds the compress
side of the overlapped drop-in pair.  Usage: python tools/exp_chunked_k1.py [chunk_blocks ...]"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pim-compression_amd"))
import silesia_mix  # noqa: E402
import snappy_hip_binding as shb  # noqa: E402

BS = 32768


def main():
    chunks = [int(a) for a in sys.argv[1:]] or [2048, 4096, 5888, 8192, 16384]
    with open(os.path.join(silesia_mix.GOLDEN, "xml.snappy"), "rb") as f:
        st, xml, _ = shb.decompress_host(f.read())
    assert st == 0 and hashlib.sha256(xml).hexdigest() == silesia_mix.XML_TXT_SHA256
    unit = silesia_mix.build_unit(np.frombuffer(xml, dtype=np.uint8), seed=0)
    n = 1 << 30
    d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
    nb = n // BS

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    ws = shb.CompressWorkspace(n, BS)
    t = timed(lambda: shb.compress_blocks(d_in, n, ws))
    ref = ws.block_bytes.clone()
    print(f"one launch of {nb} blocks: {t * 1e3:.2f} ms  {n / t / 1e9:.1f} GB/s", flush=True)

    class View:      # a workspace window over blocks [b0, b0+k) of ws, with its own scratch
        def __init__(self, b0, k, scratch_ws):
            self.block_size, self.stride = BS, ws.stride
            self.slots = ws.slots[b0 * ws.stride:(b0 + k) * ws.stride]
            self.block_bytes = ws.block_bytes[b0:b0 + k]
            self.scratch_ptr, self.scratch_bytes = scratch_ws.scratch_ptr, scratch_ws.scratch_bytes

    ws2 = shb.CompressWorkspace(BS, BS)
    ws3 = shb.CompressWorkspace(BS, BS)
    streams = [torch.cuda.Stream() for _ in range(3)]
    scr = [ws, ws2, ws3]
    for chunk in chunks:
        ranges = [(b, min(chunk, nb - b)) for b in range(0, nb, chunk)]

        def run(width):
            main_s = torch.cuda.current_stream()
            for s in streams[:width]:
                s.wait_stream(main_s)
            for i, (b0, k) in enumerate(ranges):
                with torch.cuda.stream(streams[i % width]):
                    shb.compress_blocks(d_in[b0 * BS:], k * BS, View(b0, k, scr[i % width]))
            for s in streams[:width]:
                main_s.wait_stream(s)

        for width in (1, 2, 3):
            ws.block_bytes.zero_()
            t = timed(lambda: run(width))
            ok = bool(torch.equal(ws.block_bytes, ref))
            print(f"chunks of {chunk:5d} blocks, {width} in flight: {t * 1e3:7.2f} ms  {n / t / 1e9:5.1f} GB/s  "
                  f"{t * 1e3 / len(ranges):.2f} ms/launch  same bytes per block: {ok}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
