"""ctypes binding of the CPU oracle (oracle/liboracle_snappy.so). Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle_snappy.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        L = ctypes.CDLL(path)
        u8p = ctypes.c_void_p
        L.oracle_max_compressed_length.restype = ctypes.c_uint64
        L.oracle_max_compressed_length.argtypes = [ctypes.c_uint64]
        L.oracle_table_size.restype = ctypes.c_uint32
        L.oracle_table_size.argtypes = [ctypes.c_uint32]
        L.oracle_compress_bound.restype = ctypes.c_uint64
        L.oracle_compress_bound.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
        L.oracle_compress.restype = ctypes.c_uint64
        L.oracle_compress.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint32, u8p, ctypes.c_uint64]
        L.oracle_compress_mt.restype = ctypes.c_uint64
        L.oracle_compress_mt.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint32, u8p, ctypes.c_uint64, ctypes.c_int]
        L.oracle_read_header.restype = ctypes.c_uint32
        L.oracle_read_header.argtypes = [u8p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        L.oracle_decompress.restype = ctypes.c_int
        L.oracle_decompress.argtypes = [u8p, ctypes.c_uint64, u8p, ctypes.c_uint64]
        L.oracle_decompress_mt.restype = ctypes.c_int
        L.oracle_decompress_mt.argtypes = [u8p, ctypes.c_uint64, u8p, ctypes.c_uint64, ctypes.c_int]
        L.oracle_index_blocks.restype = ctypes.c_int64
        L.oracle_index_blocks.argtypes = [u8p, ctypes.c_uint64, u8p, ctypes.c_uint64]
        _LIB = L
    return _LIB


def _arr(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a)


def compress(data, block_size=32768, threads=0):
    """Oracle compress -> bytes of the framed stream."""
    a = _arr(data)
    cap = int(lib().oracle_compress_bound(a.size, block_size))
    out = np.empty(cap, dtype=np.uint8)
    if threads and threads > 0:
        n = lib().oracle_compress_mt(a.ctypes.data, a.size, block_size, out.ctypes.data, cap, threads)
    else:
        n = lib().oracle_compress(a.ctypes.data, a.size, block_size, out.ctypes.data, cap)
    if n == 0:
        raise RuntimeError("oracle_compress failed")
    return out[:n].tobytes()


def read_header(stream):
    a = _arr(stream)
    total = ctypes.c_uint32()
    bs = ctypes.c_uint32()
    h = lib().oracle_read_header(a.ctypes.data, a.size, ctypes.byref(total), ctypes.byref(bs))
    if h == 0:
        raise ValueError("bad header")
    return total.value, bs.value, h


def decompress(stream, threads=0):
    """Oracle decompress -> (status, bytes)."""
    a = _arr(stream)
    total, _, _ = read_header(a)
    out = np.zeros(max(total, 1), dtype=np.uint8)
    if threads and threads > 0:
        st = lib().oracle_decompress_mt(a.ctypes.data, a.size, out.ctypes.data, total, threads)
    else:
        st = lib().oracle_decompress(a.ctypes.data, a.size, out.ctypes.data, total)
    return st, out[:total].tobytes()


def index_blocks(stream):
    a = _arr(stream)
    total, bs, _ = read_header(a)
    nb = (total + bs - 1) // bs
    offs = np.zeros(max(nb, 1), dtype=np.uint64)
    got = lib().oracle_index_blocks(a.ctypes.data, a.size, offs.ctypes.data, nb)
    if got < 0:
        raise ValueError("bad chain")
    return offs[:nb]
