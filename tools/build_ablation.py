#!/usr/bin/env python3
"""Build pim-compression_amd/libsnappy_hip_ablation.so: the product sources + the non-default kernel forms under
csrc/ablation/ (-DSNAPPY_ABLATION).  The SNAPPY_HIP_K1_* / SNAPPY_HIP_COMPRESS_VARIANT=4,5 / SNAPPY_HIP_DECOMPRESS_VARIANT
knobs documented in DESIGN.md only exist in this build; the product library rejects them.  Not a product build."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "pim-compression_amd", "libsnappy_hip_ablation.so")


def build():
    src = os.path.join(ROOT, "pim-compression_amd", "csrc", "snappy_hip.hip")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-shared", "-DSNAPPY_ABLATION", src, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build())
