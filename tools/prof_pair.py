#!/usr/bin/env python3
"""Phase profile of the two-wavefront K1 (diagnostic build with s_memtime probes, -DSNAPPY_PAIR_PROBE; not a product build).
Usage: python tools/prof_pair.py [MiB] ["ENV=V,ENV=V" ...]   (builds pim-compression_amd/libsnappy_hip_prof.so first)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
PROF = os.path.join(ROOT, "pim-compression_amd", "libsnappy_hip_prof.so")


def build():
    src = os.path.join(ROOT, "pim-compression_amd", "csrc", "snappy_hip.hip")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-shared", "-DSNAPPY_PAIR_PROBE", src, "-o", PROF])


def main():
    if "--build-only" in sys.argv:
        build()
        return
    if not os.path.exists(PROF):
        build()
    import numpy as np
    import torch
    import silesia_mix
    import snappy_hip_binding as shb
    shb.LIB_PATH = PROF
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mib = int(args[0]) if args else 1024
    configs = args[1:] or ["SNAPPY_HIP_PAIR_PER_CU=4,SNAPPY_HIP_GT_WAVES=0"]
    n = mib << 20
    with open(os.path.join(ROOT, "tests", "golden", "xml.snappy"), "rb") as f:
        xs = np.frombuffer(f.read(), dtype=np.uint8).copy()
    st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
    unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
    d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
    ws = shb.CompressWorkspace(n, 32768)
    L = shb.lib()
    L.snappy_hip_debug_pair_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    names = ["prep", "wait", "critical", "token_write", "post", "windows"]
    for cfg in configs:
        kv = dict(x.split("=") for x in cfg.split(","))
        os.environ.update(kv)
        shb.compress_blocks(d_in, n, ws)
        torch.cuda.synchronize()
        L.snappy_hip_debug_pair_prof(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        shb.compress_blocks(d_in, n, ws)
        e1.record()
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 16)()
        L.snappy_hip_debug_pair_prof(out, 0)
        ms = e0.elapsed_time(e1)
        wins = max(1, out[5])
        tot = sum(out[i] for i in range(5))
        print(f"== {cfg}: {ms:.2f} ms, {n / ms / 1e6:.1f} GB/s; windows with a critical section {wins}; "
              f"cycles per window (one wavefront) {tot / wins:.0f}")
        for i in range(5):
            print(f"   {names[i]:12s} {100.0 * out[i] / tot:5.1f}%  {out[i] / wins:8.0f} cycles per window")
        print(f"   inside critical: segments {out[7] / wins:.2f}/window, walk+resolves {out[6] / wins:.0f} cycles/window; resolves "
              f"{out[8] / wins:.2f}/window at {out[9] / max(1, out[8]):.0f} cycles each ({out[10] / wins:.2f} match_extend calls/window); "
              f"commit+accounting {out[11] / wins:.0f} cycles/window; single steps {out[12] / wins:.3f}/window")
        for k in kv:
            os.environ.pop(k, None)


if __name__ == "__main__":
    main()
