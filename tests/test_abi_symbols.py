"""CPU tests: the C-ABI library loads and exports every symbol include/snappy_hip.h declares;
host-side helpers (no GPU needed) behave like the reference's framing code."""
import ctypes
import os
import re

import oracle_lib as oracle
from conftest import ROOT, golden_bytes

import __graft_entry__ as entry


def _declared_functions():
    with open(os.path.join(ROOT, "include", "snappy_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(snappy_[a-z_]+)\s*\(", text)
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    path = entry.build_hip()
    L = ctypes.CDLL(path)
    names = _declared_functions()
    assert "snappy_compress_gpu" in names and "snappy_decompress_gpu" in names and len(names) >= 14
    for n in names:
        assert hasattr(L, n), n


def test_code_object_targets_gfx950():
    path = entry.build_hip()
    with open(path, "rb") as f:
        blob = f.read()
    assert b"gfx950" in blob
    import snappy_hip_binding as shb
    assert shb.lib().snappy_hip_arch() == b"gfx950"


def test_host_helpers_match_reference_framing():
    import snappy_hip_binding as shb
    # header bytes of the goldens
    for name, hexhdr in (("alice", "b802808002"), ("coding", "cf49808002"), ("terror2", "deb706808002"),
                         ("xml", "80a0c602808002")):
        s = golden_bytes(name + ".snappy")
        total, bs, hdr = shb.parse_header(s)
        assert s[:hdr] == bytes.fromhex(hexhdr)
        assert bs == 32768
        assert shb.write_header(total, bs) == s[:hdr]
        assert (total, bs, hdr) == oracle.read_header(s)
    assert shb.write_header(0, 32768) == bytes.fromhex("00808002")
    assert shb.slot_stride(32768) % 16 == 0 and shb.slot_stride(32768) >= 4 + 32 + 32768 + 32768 // 6
    assert shb.num_blocks(105438, 32768) == 4 and shb.num_blocks(0, 32768) == 0
    # every slot can hold the oracle's worst case (incompressible block)
    import datagen
    for bs in (64, 1000, 32768, 65535):
        c = oracle.compress(datagen.random_bytes(bs, seed=bs), bs)
        assert len(c) - 4 <= shb.slot_stride(bs) + 6


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the drop-in entry points must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        return
    import snappy_hip_binding as shb
    st, stream, _ = shb.compress_host(b"hello hello hello hello hello", 32768)
    assert st != 0 and stream == b""
    st, plain, _ = shb.decompress_host(golden_bytes("alice.snappy"))
    assert st != 0 and plain == b""
