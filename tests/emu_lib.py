"""ctypes binding of the CPU wave emulator build of the kernels (tests/emu). Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_LIB = None


def build():
    src = os.path.join(HERE, "emu", "emu_runtime.cpp")
    # EMU_CXXFLAGS (e.g. -DSNAPPY_K2_WALK_LEVELS=2): extra flags for an experimental build of the kernels, kept in a library of its own
    extra = os.environ.get("EMU_CXXFLAGS", "").split()
    tag = ("_" + "".join(c if c.isalnum() else "_" for c in "".join(extra))) if extra else ""
    out = os.path.join(HERE, "emu", f"libsnappy_emu{tag}.so")
    csrc = os.path.join(ROOT, "pim-compression_amd", "csrc")
    deps = [src, os.path.join(HERE, "emu", "hip", "hip_runtime.h"), os.path.join(csrc, "snappy_kernels.hpp"),
            os.path.join(csrc, "snappy_k1_stream.hpp")] + [os.path.join(csrc, "ablation", f) for f in os.listdir(os.path.join(csrc, "ablation"))]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        # -DSNAPPY_ABLATION: the emulator also compiles the non-default kernel forms under csrc/ablation/
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DSNAPPY_ABLATION", "-I" + os.path.join(HERE, "emu"),
                               "-I" + os.path.join(ROOT, "pim-compression_amd", "csrc")] + extra + [src, "-o", out])
    return out


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.emu_compress.restype = ctypes.c_uint64
        L.emu_compress.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64]
        L.emu_compress_variant.restype = ctypes.c_uint64
        L.emu_compress_variant.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                           ctypes.c_int]
        L.emu_decompress.restype = ctypes.c_int
        L.emu_decompress.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                     ctypes.c_void_p]
        L.emu_decompress_variant.restype = ctypes.c_int
        L.emu_decompress_variant.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.c_void_p, ctypes.c_int]
        L.emu_verify_index.restype = None
        L.emu_verify_index.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_void_p]
        L.emu_index_parallel.restype = ctypes.c_int
        L.emu_index_parallel.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_void_p]
        _LIB = L
    return _LIB


def compress(data, block_size=32768, variant=43503):   # the product default for blocks of more than 8 KiB: global table behind the slot cache, stream form
    a = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
    n = len(data)
    nb = (n + block_size - 1) // block_size
    stride = (4 + 32 + block_size + block_size // 6 + 15) & ~15
    cap = 10 + nb * stride
    out = np.zeros(cap + 16, dtype=np.uint8)
    got = lib().emu_compress_variant(a.ctypes.data, n, block_size, out.ctypes.data, cap, variant)
    assert got > 0
    return out[:got].tobytes()


def decompress(stream, total_len, block_size, header_len, variant=3):   # 3 = the product's per-window batch decoder
    a = np.frombuffer(stream, dtype=np.uint8).copy()
    out = np.zeros(max(total_len, 1) + 16, dtype=np.uint8)
    st = lib().emu_decompress_variant(a.ctypes.data, a.size, total_len, block_size, header_len, out.ctypes.data, variant)
    return st, out[:total_len].tobytes()


def verify_index(stream, offsets, total_len, block_size, header_len):
    """verify_index kernels on a candidate index (num_blocks + 1 offsets) -> (status, links that hold)."""
    a = np.frombuffer(stream, dtype=np.uint8).copy()
    offs = np.ascontiguousarray(np.asarray(offsets, dtype=np.uint64))
    res = np.zeros(2, dtype=np.uint32)
    lib().emu_verify_index(a.ctypes.data, a.size, offs.ctypes.data, total_len, block_size, header_len, res.ctypes.data)
    return int(res[0]), int(res[1])


def index_parallel(stream, total_len, block_size, header_len):
    """chain_*_kernel + the serial walk for what they leave -> (resolved in parallel?, status, blocks, offsets)."""
    a = np.frombuffer(stream, dtype=np.uint8).copy()
    nb = (total_len + block_size - 1) // block_size if block_size else 0
    offs = np.zeros(max(nb, 1), dtype=np.uint64)
    res = np.zeros(2, dtype=np.uint32)
    resolved = lib().emu_index_parallel(a.ctypes.data, a.size, offs.ctypes.data, total_len, block_size, header_len, res.ctypes.data)
    return bool(resolved), int(res[0]), int(res[1]), offs[:nb].copy()
