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
        L.oracle_mt_create.restype = ctypes.c_void_p
        L.oracle_mt_create.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int]
        L.oracle_mt_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_mt_compress.restype = ctypes.c_uint64
        L.oracle_mt_compress.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64, u8p, ctypes.c_uint64]
        L.oracle_mt_decompress.restype = ctypes.c_int
        L.oracle_mt_decompress.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64, u8p, ctypes.c_uint64]
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


class MtContext:
    """All-cores oracle with a persistent, pre-faulted workspace (bench.py's cpu_baseline leg): compress() / decompress()
    work on caller-owned numpy buffers that were allocated and touched beforehand, so a timed call measures the codec
    threads and their parallel concat only."""

    def __init__(self, max_n, block_size, threads):
        self.block_size, self.threads = block_size, threads
        self.ctx = lib().oracle_mt_create(max_n, block_size, threads)
        if not self.ctx:
            raise MemoryError("oracle_mt_create failed")
        self.bound = int(lib().oracle_compress_bound(max_n, block_size))

    def compress_into(self, src, dst):
        """src, dst: contiguous uint8 numpy arrays (dst of at least self.bound bytes); returns the stream length."""
        n = lib().oracle_mt_compress(self.ctx, src.ctypes.data, src.size, dst.ctypes.data, dst.size)
        if n == 0:
            raise RuntimeError("oracle_mt_compress failed")
        return int(n)

    def decompress_into(self, stream, stream_len, out):
        return int(lib().oracle_mt_decompress(self.ctx, stream.ctypes.data, stream_len, out.ctypes.data, out.size))

    def close(self):
        if self.ctx:
            lib().oracle_mt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()
