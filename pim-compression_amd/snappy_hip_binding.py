"""ctypes binding of libsnappy_hip.so (include/snappy_hip.h) for the tests, bench.py and smoke().

Plumbing only: torch provides device memory and streams, every byte of codec work happens in
the HIP library.  There is no fallback: if the library or a GPU is missing, calls raise.
"""
import ctypes
import os

import numpy as np

# The drop-in pair's overlapped pipeline keeps six HIP streams busy; HIP maps streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4).  The host program owns its environment (the library never calls setenv), so this host-side module
# asks for 8 unless the variable is set already; it takes effect if HIP has not been initialised yet.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsnappy_hip.so")

SNAPPY_OK, SNAPPY_INVALID_INPUT, SNAPPY_BUFFER_TOO_SMALL = 0, 1, 2


class HostBufferContext(ctypes.Structure):
    """struct host_buffer_context, reference snappy/dpu_snappy.h:37-44."""
    _fields_ = [("file_name", ctypes.c_char_p), ("buffer", ctypes.c_void_p), ("curr", ctypes.c_void_p),
                ("length", ctypes.c_ulong), ("max", ctypes.c_ulong)]


class ProgramRuntime(ctypes.Structure):
    """struct program_runtime, reference snappy/dpu_snappy.h:47-55."""
    _fields_ = [(k, ctypes.c_double) for k in ("pre", "d_alloc", "load", "copy_in", "run", "copy_out", "d_free")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class StreamDesc(ctypes.Structure):
    """snappy_hip_stream_desc."""
    _fields_ = [("stream", ctypes.c_void_p), ("stream_len", ctypes.c_uint64), ("block_offsets", ctypes.c_void_p),
                ("result", ctypes.c_void_p), ("total_len", ctypes.c_uint32), ("block_size", ctypes.c_uint32),
                ("header_len", ctypes.c_uint32), ("num_blocks", ctypes.c_uint32)]


STREAM_DESC_DTYPE = np.dtype([("stream", "<u8"), ("stream_len", "<u8"), ("block_offsets", "<u8"), ("result", "<u8"),
                              ("total_len", "<u4"), ("block_size", "<u4"), ("header_len", "<u4"), ("num_blocks", "<u4")])

_LIB = None
_LIBC = None


class SnappyHipError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SnappyHipError(f"{LIB_PATH} is missing: build it (python -c 'import __graft_entry__ as g; g.build()')")
        L = ctypes.CDLL(LIB_PATH)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.snappy_hip_device_count.restype = ctypes.c_int
        L.snappy_hip_set_device.restype = ctypes.c_int
        L.snappy_hip_set_device.argtypes = [ctypes.c_int]
        L.snappy_hip_last_error.restype = ctypes.c_char_p
        L.snappy_hip_arch.restype = ctypes.c_char_p
        L.snappy_hip_slot_stride.restype = u32
        L.snappy_hip_slot_stride.argtypes = [u32]
        L.snappy_hip_num_blocks.restype = u64
        L.snappy_hip_num_blocks.argtypes = [u64, u32]
        L.snappy_hip_stream_bound.restype = u64
        L.snappy_hip_stream_bound.argtypes = [u64, u32]
        L.snappy_hip_write_header.restype = u32
        L.snappy_hip_write_header.argtypes = [vp, u32, u32]
        L.snappy_hip_parse_header.restype = u32
        L.snappy_hip_parse_header.argtypes = [vp, u64, ctypes.POINTER(u32), ctypes.POINTER(u32)]
        L.snappy_hip_compress_blocks.restype = ctypes.c_int
        L.snappy_hip_compress_blocks.argtypes = [vp, u64, u32, vp, u32, vp, vp, u64, vp]
        L.snappy_hip_compress_blocks_batch.restype = ctypes.c_int
        L.snappy_hip_compress_blocks_batch.argtypes = [vp, u32, u32, u32, vp, u64, vp]
        L.snappy_hip_compress_scratch_bytes.restype = u64
        L.snappy_hip_k1_lds_waves_per_cu.restype = u32
        L.snappy_hip_k1_lds_waves_per_cu.argtypes = [u32]
        L.snappy_hip_compact.restype = ctypes.c_int
        L.snappy_hip_compact.argtypes = [vp, u32, vp, u64, u32, vp, vp, vp, vp]
        L.snappy_hip_index_streams.restype = ctypes.c_int
        L.snappy_hip_index_streams.argtypes = [vp, u32, vp]
        L.snappy_hip_decompress_blocks_batch.restype = ctypes.c_int
        L.snappy_hip_decompress_blocks_batch.argtypes = [vp, u32, u32, vp]
        L.snappy_hip_verify_index.restype = ctypes.c_int
        L.snappy_hip_verify_index.argtypes = [vp, u32, vp]
        L.snappy_hip_decompress_blocks.restype = ctypes.c_int
        L.snappy_hip_decompress_blocks.argtypes = [vp, u64, vp, u64, u32, vp, vp, vp]
        L.snappy_compress_gpu.restype = ctypes.c_int
        L.snappy_compress_gpu.argtypes = [ctypes.POINTER(HostBufferContext), ctypes.POINTER(HostBufferContext), u32,
                                          ctypes.POINTER(ProgramRuntime)]
        L.snappy_decompress_gpu.restype = ctypes.c_int
        L.snappy_decompress_gpu.argtypes = [ctypes.POINTER(HostBufferContext), ctypes.POINTER(HostBufferContext),
                                            ctypes.POINTER(ProgramRuntime)]
        _LIB = L
    return _LIB


def libc():
    global _LIBC
    if _LIBC is None:
        _LIBC = ctypes.CDLL(None)
        _LIBC.free.argtypes = [ctypes.c_void_p]
        _LIBC.malloc.restype = ctypes.c_void_p
        _LIBC.malloc.argtypes = [ctypes.c_size_t]
    return _LIBC


def _check(rc, what):
    if rc != 0:
        raise SnappyHipError(f"{what} failed ({rc}): {lib().snappy_hip_last_error().decode()}")


def slot_stride(block_size):
    return int(lib().snappy_hip_slot_stride(block_size))


def num_blocks(n, block_size):
    return int(lib().snappy_hip_num_blocks(n, block_size))


def shard_block_range(num_blocks_total, shards, shard):
    """(first_block, block_count) of `shard` when a file of `num_blocks_total` blocks is split over `shards` devices: the
    contiguous ranges of ceil(B / G) blocks that snappy_compress_gpu / snappy_decompress_gpu use (csrc/snappy_hip.hip),
    i.e. the partitioning of the reference's input_blocks_per_dpu (snappy_compress.c:494-520)."""
    per = (num_blocks_total + shards - 1) // shards if num_blocks_total else 0
    first = min(num_blocks_total, shard * per)
    return first, min(num_blocks_total, first + per) - first


def k1_lds_waves_per_cu(block_size):
    return int(lib().snappy_hip_k1_lds_waves_per_cu(block_size))


def write_header(total_len, block_size):
    buf = (ctypes.c_uint8 * 10)()
    k = lib().snappy_hip_write_header(buf, total_len, block_size)
    return bytes(buf[:k])


def parse_header(data):
    a = np.frombuffer(data[:10], dtype=np.uint8).copy()
    total, bs = ctypes.c_uint32(), ctypes.c_uint32()
    h = lib().snappy_hip_parse_header(a.ctypes.data, a.size, ctypes.byref(total), ctypes.byref(bs))
    if h == 0:
        raise ValueError("malformed header")
    return total.value, bs.value, h


# ---------------------------------------------------------------------------
# resident API (device tensors)
# ---------------------------------------------------------------------------

def _stream_handle(torch):
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class CompressWorkspace:
    """Device buffers for compressing containers of up to `max_len` bytes at `block_size`."""

    def __init__(self, max_len, block_size, device="cuda", scratch=True):
        """scratch=False: no hash-table workspace of its own (a batch launch uses one workspace's scratch for all)."""
        import torch
        self._torch = torch
        self.block_size = block_size
        self.max_len = max_len
        self.stride = slot_stride(block_size)
        nb = num_blocks(max_len, block_size)
        self.slots = torch.empty(max(nb, 1) * self.stride, dtype=torch.uint8, device=device)
        self.block_bytes = torch.empty(max(nb, 1), dtype=torch.int32, device=device)
        self.offsets = torch.empty(nb + 1, dtype=torch.int64, device=device)
        self.stream_len = torch.zeros(1, dtype=torch.int64, device=device)
        self.scratch_bytes = int(lib().snappy_hip_compress_scratch_bytes()) if scratch else 0
        self.scratch = torch.empty(self.scratch_bytes + 256, dtype=torch.uint8, device=device) if scratch else None
        self.scratch_ptr = ((self.scratch.data_ptr() + 255) & ~255) if scratch else 0

    def lds_form_blocks(self):
        """Blocks of the last compress_blocks() launch that were taken by the LDS-table wavefronts (statistics)."""
        off = self.scratch_ptr - self.scratch.data_ptr()
        return int(self.scratch[off + 16:off + 20].view(self._torch.int32).item())

    def stream_capacity(self, n):
        return int(lib().snappy_hip_stream_bound(n, self.block_size))


def compress_blocks(d_in, n, ws):
    """K1 only: per-block compress into ws.slots / ws.block_bytes (async on the current stream)."""
    import torch
    _check(lib().snappy_hip_compress_blocks(d_in.data_ptr(), n, ws.block_size, ws.slots.data_ptr(), ws.stride,
                                            ws.block_bytes.data_ptr(), ws.scratch_ptr, ws.scratch_bytes,
                                            _stream_handle(torch)), "snappy_hip_compress_blocks")


class _CompressItem(ctypes.Structure):          # struct snappy_hip_compress_item
    _fields_ = [("d_input", ctypes.c_void_p), ("input_len", ctypes.c_uint64), ("d_slots", ctypes.c_void_p),
                ("d_block_bytes", ctypes.c_void_p)]


def compress_blocks_batch(jobs, scratch_ws=None):
    """K1 over several containers in one launch.  jobs: list of (d_in, n, ws); every ws has its own slots / block_bytes,
    the scratch of `scratch_ws` (default: the first job's) is the launch's hash-table workspace."""
    import torch
    if not jobs:
        return
    sw = scratch_ws or jobs[0][2]
    items = (_CompressItem * len(jobs))()
    for k, (d_in, n, ws) in enumerate(jobs):
        assert ws.block_size == sw.block_size and ws.stride == sw.stride
        items[k] = _CompressItem(d_in.data_ptr(), n, ws.slots.data_ptr(), ws.block_bytes.data_ptr())
    _check(lib().snappy_hip_compress_blocks_batch(ctypes.cast(items, ctypes.c_void_p), len(jobs), sw.block_size, sw.stride,
                                                  sw.scratch_ptr, sw.scratch_bytes, _stream_handle(torch)),
           "snappy_hip_compress_blocks_batch")


def compact(n, ws, d_stream):
    """scan + gather into d_stream (async); ws.stream_len[0] holds the stream length afterwards."""
    import torch
    _check(lib().snappy_hip_compact(ws.slots.data_ptr(), ws.stride, ws.block_bytes.data_ptr(), n, ws.block_size,
                                    d_stream.data_ptr(), ws.offsets.data_ptr(), ws.stream_len.data_ptr(),
                                    _stream_handle(torch)), "snappy_hip_compact")


def compress_resident(d_in, block_size=32768, n=None):
    """Compress a uint8 CUDA tensor; returns the framed stream as a CUDA uint8 tensor (exact length)."""
    import torch
    n = d_in.numel() if n is None else n
    ws = CompressWorkspace(n, block_size, d_in.device)
    d_stream = torch.empty(ws.stream_capacity(n) + 16, dtype=torch.uint8, device=d_in.device)
    compress_blocks(d_in, n, ws)
    compact(n, ws, d_stream)
    length = int(ws.stream_len.item())
    return d_stream[:length]


def make_stream_descs(entries, device="cuda"):
    """entries: list of dicts(stream=tensor, stream_len, block_offsets=tensor, result=tensor, total_len, block_size,
    header_len, num_blocks) -> device tensor of packed snappy_hip_stream_desc."""
    import torch
    arr = np.zeros(len(entries), dtype=STREAM_DESC_DTYPE)
    for i, e in enumerate(entries):
        arr[i] = (e["stream"].data_ptr(), e["stream_len"], e["block_offsets"].data_ptr(), e["result"].data_ptr(),
                  e["total_len"], e["block_size"], e["header_len"], e["num_blocks"])
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device)


def index_streams(d_descs, count):
    import torch
    _check(lib().snappy_hip_index_streams(d_descs.data_ptr(), count, _stream_handle(torch)), "snappy_hip_index_streams")


def verify_index(d_descs, count):
    """Parallel check of candidate indexes (num_blocks + 1 offsets per stream) against the streams' size chains."""
    import torch
    _check(lib().snappy_hip_verify_index(d_descs.data_ptr(), count, _stream_handle(torch)), "snappy_hip_verify_index")


def decompress_blocks(d_stream, stream_len, d_block_offsets, total_len, block_size, d_out, d_status):
    import torch
    _check(lib().snappy_hip_decompress_blocks(d_stream.data_ptr(), stream_len, d_block_offsets.data_ptr(), total_len,
                                              block_size, d_out.data_ptr(), d_status.data_ptr(), _stream_handle(torch)),
           "snappy_hip_decompress_blocks")


class _DecompressItem(ctypes.Structure):        # struct snappy_hip_decompress_item
    _fields_ = [("d_stream", ctypes.c_void_p), ("stream_len", ctypes.c_uint64), ("d_stream_len", ctypes.c_void_p),
                ("d_block_offsets", ctypes.c_void_p), ("total_len", ctypes.c_uint64), ("d_out", ctypes.c_void_p),
                ("d_status", ctypes.c_void_p)]


def decompress_blocks_batch(jobs, block_size):
    """K2 over several streams in one launch.  jobs: list of (d_stream, stream_len, d_block_offsets, total_len, d_out, d_status);
    stream_len is an int, or a device int64 tensor of one element (the length stays on the device)."""
    import torch
    if not jobs:
        return
    items = (_DecompressItem * len(jobs))()
    for k, (d_stream, slen, d_off, total, d_out, d_status) in enumerate(jobs):
        on_dev = hasattr(slen, "data_ptr")
        items[k] = _DecompressItem(d_stream.data_ptr(), 0 if on_dev else slen, slen.data_ptr() if on_dev else None, d_off.data_ptr(),
                                   total, d_out.data_ptr(), d_status.data_ptr())
    _check(lib().snappy_hip_decompress_blocks_batch(ctypes.cast(items, ctypes.c_void_p), len(jobs), block_size,
                                                    _stream_handle(torch)), "snappy_hip_decompress_blocks_batch")


def decompress_resident(d_stream, stream_len=None):
    """Decode a framed stream held in a CUDA uint8 tensor; returns (status, plaintext CUDA tensor)."""
    import torch
    stream_len = d_stream.numel() if stream_len is None else stream_len
    head = bytes(d_stream[:min(10, stream_len)].cpu().numpy())
    total, bs, hdr = parse_header(head)
    nb = num_blocks(total, bs) if bs else 0
    dev = d_stream.device
    d_out = torch.empty(total + 16, dtype=torch.uint8, device=dev)
    if nb == 0:
        return (0 if stream_len == hdr else 1), d_out[:total]
    d_boff = torch.empty(nb, dtype=torch.int64, device=dev)
    d_res = torch.full((2,), 7, dtype=torch.int32, device=dev)
    d_status = torch.full((nb,), 9, dtype=torch.int32, device=dev)
    descs = make_stream_descs([dict(stream=d_stream, stream_len=stream_len, block_offsets=d_boff, result=d_res,
                                    total_len=total, block_size=bs, header_len=hdr, num_blocks=nb)], dev)
    index_streams(descs, 1)
    res = d_res.cpu().numpy()
    if res[0] != 0 or res[1] != nb:
        return 1, d_out[:total]
    decompress_blocks(d_stream, stream_len, d_boff, total, bs, d_out, d_status)
    bad = int((d_status != 0).sum().item())
    return (1 if bad else 0), d_out[:total]


# ---------------------------------------------------------------------------
# drop-in pair (host buffers), driven the way dpu_snappy.c's main() drives the *_dpu functions
# ---------------------------------------------------------------------------

def compress_host(data, block_size=32768, out_capacity=None):
    """snappy_compress_gpu on a host buffer -> (status, stream bytes, runtime dict).  out_capacity: hand over a
    caller-owned output buffer of that many bytes (finite `max`) instead of letting the callee allocate."""
    a = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
    inp = HostBufferContext(b"<memory>", a.ctypes.data, a.ctypes.data, len(data), (1 << 64) - 1)
    if out_capacity is None:
        out = HostBufferContext(b"<memory>", None, None, 0, (1 << 64) - 1)
    else:
        buf = libc().malloc(max(1, out_capacity))
        out = HostBufferContext(b"<memory>", buf, buf, 0, out_capacity)
    rt = ProgramRuntime()
    st = lib().snappy_compress_gpu(ctypes.byref(inp), ctypes.byref(out), block_size, ctypes.byref(rt))
    stream = b""
    if st == SNAPPY_OK:
        stream = ctypes.string_at(out.buffer, out.length)
    if out.buffer:
        libc().free(out.buffer)
    return st, stream, rt.as_dict()


def decompress_host(stream, out_len_override=None):
    """setup_decompression (reference snappy_decompress.c:187-215) + snappy_decompress_gpu -> (status, bytes, runtime).
    out_len_override: bytes to allocate for the plaintext instead of the header's length (tests with hostile headers)."""
    a = np.frombuffer(stream, dtype=np.uint8).copy()
    # first varint: uncompressed length
    total, shift, used = 0, 0, 0
    ok = False
    for k in range(min(5, a.size)):
        c = int(a[k])
        total |= (c & 0x7f) << shift
        used = k + 1
        if not c & 0x80:
            ok = True
            break
        shift += 7
    if not ok:
        return SNAPPY_INVALID_INPUT, b"", {}
    size = (((total + 7) & ~7) | 2047) if out_len_override is None else max(8, out_len_override)
    buf = libc().malloc(size)
    inp = HostBufferContext(b"<memory>", a.ctypes.data, a.ctypes.data + used, a.size, (1 << 64) - 1)
    out = HostBufferContext(b"<memory>", buf, buf, total, (1 << 64) - 1)
    rt = ProgramRuntime()
    st = lib().snappy_decompress_gpu(ctypes.byref(inp), ctypes.byref(out), ctypes.byref(rt))
    plain = ctypes.string_at(out.buffer, total) if st == SNAPPY_OK else b""
    libc().free(buf)
    return st, plain, rt.as_dict()
