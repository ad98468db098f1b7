#!/usr/bin/env python3
"""K1 / K2 throughput by data class of the Silesia-mix (text, xml, structured binary, noise): a 1 GiB container of each class
alone, product launch.  Shows where the parse forms spend the batch's time.   python tools/k1_by_data_class.py [MiB]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import silesia_mix
import snappy_hip_binding as shb

if os.environ.get("SNAPPY_PROF_LIB"):
    shb.LIB_PATH = os.environ["SNAPPY_PROF_LIB"]


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = mib << 20
    with open(os.path.join(ROOT, "tests", "golden", "xml.snappy"), "rb") as f:
        xs = np.frombuffer(f.read(), dtype=np.uint8).copy()
    st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
    xml = d_xml.cpu().numpy()
    r = np.random.default_rng(1000)
    texts = [silesia_mix._read(t) for t in ("world192.txt", "plrabn12.txt", "terror2.txt")]
    text = np.concatenate([np.roll(texts[i % 3], int(r.integers(0, texts[i % 3].size))) for i in range(12)])
    classes = {"text": text, "xml": xml, "records": silesia_mix._records(8 << 20, 2000),
               "noise": r.integers(0, 256, size=8 << 20, dtype=np.uint8), "mix unit": silesia_mix.build_unit(xml, seed=0)}
    ws = shb.CompressWorkspace(n, 32768)
    d_stream = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device="cuda")
    nb = shb.num_blocks(n, 32768)
    status = torch.empty(nb, dtype=torch.int32, device="cuda")
    out = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    for name, unit in classes.items():
        d_in = silesia_mix.container_from_unit(torch.from_numpy(np.ascontiguousarray(unit)).cuda(), n)
        tk1, tk2 = [], []
        for _ in range(4):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            shb.compress_blocks(d_in, n, ws)
            e[1].record()
            shb.compact(n, ws, d_stream)
            torch.cuda.synchronize()
            slen = int(ws.stream_len.item())
            e[2].record()
            shb.decompress_blocks(d_stream, slen, ws.offsets[:nb].contiguous(), n, 32768, out, status)
            e[3].record()
            torch.cuda.synchronize()
            tk1.append(e[0].elapsed_time(e[1]))
            tk2.append(e[2].elapsed_time(e[3]))
        ok = bool(torch.equal(out[:n], d_in[:n]))
        k1, k2 = min(tk1[1:]), min(tk2[1:])
        print(f"{name:10s} saving {1 - slen / n:6.3f}  K1 {k1:8.3f} ms {n / k1 / 1e6:7.1f} GB/s   K2 {k2:7.3f} ms {n / k2 / 1e6:7.1f} GB/s   "
              f"LDS-table share {ws.lds_form_blocks() / nb:5.3f}  round trip {ok}", flush=True)


if __name__ == "__main__":
    main()
