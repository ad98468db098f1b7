#!/usr/bin/env python3
"""Gate (b) of the two-pass K1: the ceiling of a table-free parse kernel.

K1's stream form with its hash table answered for free from records a CPU run of the reference parse made
(csrc/ablation/k1_oracle_table.hpp, tools/gate_b_records.c), timed at several wavefront counts beside the product's launch on
the same container.  Every configuration must produce the product's bytes (stream digest).  Needs the ablation build
(python tools/build_ablation.py) and gcc with OpenMP.
Usage: python tools/gate_b_ceiling.py [MiB] [pmc]   (default 1024; "pmc": only the two kernels compared under rocprofv3 --pmc)"""
import ctypes
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import silesia_mix
import snappy_hip_binding as shb

shb.LIB_PATH = os.path.join(ROOT, "pim-compression_amd", "libsnappy_hip_ablation.so")


def host_lib():
    out = os.path.join(ROOT, "tools", "ab", "libgate_b_records.so")
    src = os.path.join(ROOT, "tools", "gate_b_records.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", out, src])
    L = ctypes.CDLL(out)
    L.gate_b_records.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p]
    return L


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = mib << 20
    bs = 32768
    with open(os.path.join(ROOT, "tests", "golden", "xml.snappy"), "rb") as f:
        xs = np.frombuffer(f.read(), dtype=np.uint8).copy()
    st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
    assert st == 0
    unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
    d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
    host = np.zeros(n + 64, dtype=np.uint8)
    host[:n] = d_in[:n].cpu().numpy()
    rec = np.zeros(n + 64, dtype=np.uint32)
    t0 = time.time()
    host_lib().gate_b_records(host.ctypes.data, n, bs, rec.ctypes.data)
    print(f"host records for {mib} MiB: {time.time() - t0:.1f} s", flush=True)
    d_rec = torch.from_numpy(rec).cuda()
    prevw = np.zeros(n + 64, dtype=np.uint16)
    HL = host_lib()
    HL.gate_b_prevw.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p]
    HL.gate_b_prevw(host.ctypes.data, n, bs, prevw.ctypes.data)
    d_prevw = torch.from_numpy(prevw).cuda()
    L = shb.lib()
    L.snappy_hip_debug_set_oracle_records.argtypes = [ctypes.c_void_p]
    L.snappy_hip_debug_set_oracle_records.restype = None
    L.snappy_hip_debug_set_oracle_records(d_rec.data_ptr())
    L.snappy_hip_debug_set_oracle_prevw.argtypes = [ctypes.c_void_p]
    L.snappy_hip_debug_set_oracle_prevw.restype = None
    L.snappy_hip_debug_set_oracle_prevw(d_prevw.data_ptr())

    ws = shb.CompressWorkspace(n, bs)
    d_stream = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device="cuda")
    product = "SNAPPY_HIP_GT_CACHE=512,SNAPPY_HIP_K1_STREAM=3"
    configs = [("product launch (cached global-table + 1 LDS-table wavefront per CU)", product + ",SNAPPY_HIP_LDS_WAVES=256"),
               ("cached global-table kernel alone, 20 per CU", product + ",SNAPPY_HIP_LDS_WAVES=0"),
               ("LDS-table kernel alone, 4 per CU", "SNAPPY_HIP_K1_STREAM=3,SNAPPY_HIP_COMPRESS_VARIANT=1")]
    for per_cu in (8, 12, 16, 20, 24, 28, 32):
        configs.append((f"FREE TABLE (oracle records), {per_cu} wavefronts per CU", f"SNAPPY_HIP_COMPRESS_VARIANT=6,SNAPPY_HIP_GT_WAVES={per_cu * 256}"))
    for per_cu in (16, 20, 24):
        configs.append((f"FREE ANSWERS, PAID LOOK-UPS (bitmap + memo array), {per_cu} wavefronts per CU", f"SNAPPY_HIP_COMPRESS_VARIANT=7,SNAPPY_HIP_GT_WAVES={per_cu * 256}"))
    if len(sys.argv) > 2 and sys.argv[2] == "pmc":      # under rocprofv3 --pmc: the cached global-table kernel and the free-table kernel only
        configs = [configs[1]] + [c for c in configs if "24 wavefronts" in c[0]]
    ref = None
    for name, cfg in configs:
        kv = dict(x.split("=") for x in cfg.split(","))
        os.environ.update(kv)
        times = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            shb.compress_blocks(d_in, n, ws)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        shb.compact(n, ws, d_stream)
        slen = int(ws.stream_len.item())
        digest = hashlib.sha256(d_stream[:slen].cpu().numpy().tobytes()).hexdigest()[:16]
        ref = ref or digest
        best = min(times[1:])
        print(f"{name:72s} {best:8.3f} ms  {n / best / 1e6:7.2f} GB/s of input  -> {8192.0 / mib * best:7.1f} ms per 8 GiB  "
              f"stream {slen} {digest} {'bit-exact' if digest == ref else 'MISMATCH'}", flush=True)
        for k in kv:
            os.environ.pop(k, None)


if __name__ == "__main__":
    main()
