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
    names = re.findall(r"\b(snappy_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    path = entry.build_hip()
    L = ctypes.CDLL(path)
    names = _declared_functions()
    assert "snappy_compress_gpu" in names and "snappy_decompress_gpu" in names and len(names) >= 14
    for n in names:
        assert hasattr(L, n), n


def test_library_exports_nothing_but_the_declared_c_symbols():
    """The drop-in library is linked into somebody else's C program: its unmangled dynamic symbols must be
    exactly the functions include/snappy_hip.h declares -- no helper of the implementation (`report`, `walk_chain`, ...)
    that a host program's own function of the same name could interpose.  (Built with -fvisibility=hidden; what remains
    beside them are the mangled kernel handles of namespace snappy_hip and weak libstdc++ instantiations.)"""
    import subprocess
    path = entry.build_hip()
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = [line.split() for line in out.splitlines() if line.strip()]
    c_names = sorted(parts[-1] for parts in exported if not parts[-1].startswith("_"))
    assert c_names == _declared_functions(), sorted(set(c_names) ^ set(_declared_functions()))
    text_syms = [parts[-1] for parts in exported if parts[-2] in ("T", "t")]
    assert sorted(text_syms) == _declared_functions(), text_syms
    for parts in exported:                       # everything else is C++-mangled (kernel handles, std:: templates) or the HIP unit id
        name = parts[-1]
        assert name in c_names or name.startswith("_ZN10snappy_hip") or name.startswith("_ZNSt") or name.startswith("_ZSt") \
            or name.startswith("__hip_"), name


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


def test_default_k1_launch_shape_by_block_size(monkeypatch):
    """Host logic of K1's default launch (no GPU needed): blocks of more than 8 KiB have full-size hash tables; their
    global-table wavefronts run behind the slot cache with ONE LDS-table wavefront per CU beside them; smaller blocks keep
    round 2's shape, as many LDS-table wavefronts as fit beside eight global-table ones."""
    import snappy_hip_binding as shb
    for k in ("SNAPPY_HIP_GT_CACHE", "SNAPPY_HIP_K1_STREAM", "SNAPPY_HIP_LDS_WAVES"):
        monkeypatch.delenv(k, raising=False)
    assert [shb.k1_lds_waves_per_cu(bs) for bs in (4096, 8192, 8193, 16384, 32768, 65535)] == [10, 6, 1, 1, 1, 1]
    monkeypatch.setenv("SNAPPY_HIP_GT_CACHE", "0")            # without the cache: three 36 KiB tables per CU
    assert shb.k1_lds_waves_per_cu(32768) == 3
    monkeypatch.setenv("SNAPPY_HIP_LDS_WAVES", "1024")
    assert shb.k1_lds_waves_per_cu(32768) == 4


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


def test_k2_back_references_use_global_not_flat_instructions(tmp_path):
    """K2 serves a back-reference with a load issued after the store of the same wavefront and no s_waitcnt in between
    (csrc/snappy_kernels.hpp, decompress_blocks_kernel).  That relies on vector memory operations of one wavefront
    completing in issue order, which holds for global_* instructions but NOT for flat_* ones
    (MI355X_MICROARCH.md: "flat_* excepted: out of order").  So the invariant is checked on the generated code: the shipped
    decoder must not contain a single flat_load / flat_store / flat_atomic."""
    import subprocess
    src = os.path.join(ROOT, "pim-compression_amd", "csrc", "snappy_hip.hip")
    asm = tmp_path / "device.s"
    subprocess.check_call([entry.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", str(asm)])
    text = asm.read_text()
    m = re.search(r"^(_ZN10snappy_hip24decompress_blocks_kernelE\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)
    assert m, "decompress_blocks_kernel (the per-window batch decoder) not found in the device code"
    body = m.group(2)
    assert len(re.findall(r"^\s*global_(?:load|store)", body, re.M)) >= 20
    assert re.findall(r"^\s*flat_\w+", body, re.M) == []
    # and the product library carries exactly these K1 instantiations and one K2 (no ablation instantiations): the LDS-table
    # kernel <look-ahead 64, form> and the global-table kernel <64, form, slot filter, cache slots>, each in the bulk (2) and
    # the stream (3) form, the global-table one with and without its 512-slot cache.  <64,3,1,512> + <64,3> is the default
    # launch for blocks of more than 8 KiB, <64,2,1,0> + <64,3> for smaller ones.
    kernels = set(re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M))
    k1 = sorted(k for k in kernels if "_blocks_" in k and "decompress" not in k)
    assert len([k for k in kernels if "decompress_blocks_kernel" in k]) == 1, sorted(kernels)
    expected = sorted(["_ZN10snappy_hip32compress_blocks_lds_table_kernelILj64ELi2EEEvNS_7K1BatchEjjPj",
                       "_ZN10snappy_hip32compress_blocks_lds_table_kernelILj64ELi3EEEvNS_7K1BatchEjjPj",
                       "_ZN10snappy_hip35compress_blocks_global_table_kernelILj64ELi2ELi1ELj0EEEvNS_7K1BatchEjjPjS2_",
                       "_ZN10snappy_hip35compress_blocks_global_table_kernelILj64ELi3ELi1ELj0EEEvNS_7K1BatchEjjPjS2_",
                       "_ZN10snappy_hip35compress_blocks_global_table_kernelILj64ELi2ELi1ELj512EEEvNS_7K1BatchEjjPjS2_",
                       "_ZN10snappy_hip35compress_blocks_global_table_kernelILj64ELi3ELi1ELj512EEEvNS_7K1BatchEjjPjS2_"])
    assert k1 == expected, k1
    # CachedGlobalTable::store_masked (csrc/snappy_kernels.hpp) issues a displaced slot's write-back and, in a later
    # instruction, the write-through of a lane that lost its cache word -- possibly to the SAME global u16, from another lane.
    # The later store must land last: the same issue-order guarantee, so the same check -- no flat_* in any K1 kernel.
    for name in expected:
        m = re.search(r"^(" + re.escape(name) + r"):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)
        assert m, name
        assert re.findall(r"^\s*flat_\w+", m.group(2), re.M) == [], name
        assert len(re.findall(r"^\s*global_(?:load|store)", m.group(2), re.M)) >= 10, name
