#!/usr/bin/env python3
"""Where does the LDS-table kernel alone stop being the quicker K1 launch for a small input?  (the cut-over of
launch_shape::small_input_takes_lds_kernel_alone -- one LDS-table wavefront per SIMD, 1,024 blocks on a whole MI355X -- was
measured at 312 blocks only.)  Times K1 alone on prose inputs of N blocks for the two launch shapes, at two block sizes.
  python tools/small_inputs_threshold.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import snappy_hip_binding as shb
import standins


def k1_ms(d_in, n, ws, env):
    for k, v in env.items():
        os.environ[k] = v
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        shb.compress_blocks(d_in, n, ws)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    for k in env:
        os.environ.pop(k, None)
    ts.sort()
    return ts[len(ts) // 4], ws.lds_form_blocks()


def main():
    prose = standins.dickens_like(standins.prose_texts())
    data = np.frombuffer(prose * 9, dtype=np.uint8)
    for bs in (32768, 8192):
        print(f"block size {bs}")
        for blocks in (312, 512, 768, 900, 1024, 1100, 1280, 1563, 2048, 2571):
            n = blocks * bs
            d_in = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
            d_in[:n] = torch.from_numpy(data[:n].copy()).cuda()
            ws = shb.CompressWorkspace(n, bs)
            default, dshare = k1_ms(d_in, n, ws, {})
            lds, _ = k1_ms(d_in, n, ws, {"SNAPPY_HIP_COMPRESS_VARIANT": "1"})
            gt, _ = k1_ms(d_in, n, ws, {"SNAPPY_HIP_LDS_WAVES": "0"})
            best = "LDS-table kernel alone" if lds < gt else "global-table kernel alone"
            print(f"  {blocks:5d} blocks: default {default:6.3f} ms (LDS-table share {dshare / blocks:4.2f})   LDS-table kernel alone {lds:6.3f} ms   "
                  f"global-table kernel alone {gt:6.3f} ms   -> {best}", flush=True)


if __name__ == "__main__":
    main()
