#!/usr/bin/env python3
"""Where do the global-table kernel's memory requests come from?  Builds a COUNTING copy of the library (the kernel sources
patched in a temporary directory: every speculative slot read that passes the LDS filter, every candidate fetch and the
64-byte lines it touches, every table store, every window gathered are counted per wavefront and summed with atomics --
the parse and the bytes are unchanged) and runs the global-table kernel alone on one container.  Compare with the
TCC_EA0_RDREQ / WRREQ totals of the product build (tools/pmc_k1.sh): what the counts do not explain is overhead of the
memory system (fills for partial-line writes, input/output streams).  Not a product build.
Usage: python tools/k1_lines_breakdown.py [MiB]"""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pim-compression_amd", "csrc")
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def rep(text, old, new):
    assert old in text, "anchor not found:\n" + old
    return text.replace(old, new, 1)


def build():
    tmp = tempfile.mkdtemp(prefix="snappy_count_")
    for f in os.listdir(CSRC):
        if os.path.isfile(os.path.join(CSRC, f)):
            shutil.copy(os.path.join(CSRC, f), tmp)
    k = open(os.path.join(tmp, "snappy_kernels.hpp")).read()
    k = rep(k, "namespace snappy_hip {\n", "namespace snappy_hip {\n__device__ unsigned long long g_lines[8];\n"
            "#define COUNT_LINES(i, mask) do { const unsigned long long cm_ = (mask); if (lane == 0) atomicAdd(&g_lines[i], (unsigned long long)__builtin_popcountll(cm_)); } while (0)\n")
    # gather: slot reads that pass the filter, candidate fetches and their lines
    k = rep(k, "        if (g) ent = table.load_lane(win.h0, mine_l);\n",
            "        if (g) ent = table.load_lane(win.h0, mine_l);\n"
            "        if constexpr (!std::is_same<Table, LdsTable>::value) { COUNT_LINES(0, __ballot(g && table.is_written(win.h0))); COUNT_LINES(5, 1ull); }\n")
    k = rep(k, "        const bool deep = kDeep && worth && (win.base + lane + 28u <= block_len);   // candidate < position, so it has 28 too\n",
            "        const bool deep = kDeep && worth && (win.base + lane + 28u <= block_len);   // candidate < position, so it has 28 too\n"
            "        COUNT_LINES(1, __ballot(worth));\n"
            "        COUNT_LINES(2, __ballot(worth && (((uintptr_t)(win.blk + (ent & 0xffffu)) & 63u) + (deep ? 28u : 12u) > 64u)));\n"
            "        COUNT_LINES(6, __ballot(deep));\n")
    # commits
    k = rep(k, "        if (__builtin_amdgcn_inverse_ballot_w64(m)) table.store_lane(win.h0, win.e0 | (win.base + lane));\n",
            "        if (__builtin_amdgcn_inverse_ballot_w64(m)) table.store_lane(win.h0, win.e0 | (win.base + lane));\n"
            "        COUNT_LINES(3, m);\n")
    open(os.path.join(tmp, "snappy_kernels.hpp"), "w").write(k)
    h = open(os.path.join(tmp, "snappy_hip.hip")).read()
    h = rep(h, "uint32_t snappy_hip_k1_lds_waves_per_cu(uint32_t block_size)",
            "__attribute__((visibility(\"default\"))) int snappy_hip_debug_lines(unsigned long long* out, int reset)\n{\n"
            "    if (reset) { unsigned long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(snappy_hip::g_lines), z, sizeof(z)); }\n"
            "    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(snappy_hip::g_lines), 8 * sizeof(unsigned long long));\n}\n\n"
            "uint32_t snappy_hip_k1_lds_waves_per_cu(uint32_t block_size)")
    h = h.replace('"../../include/snappy_hip.h"', '"%s"' % os.path.join(ROOT, "include", "snappy_hip.h"))
    open(os.path.join(tmp, "snappy_hip.hip"), "w").write(h)
    out = os.path.join(ROOT, "pim-compression_amd", "libsnappy_hip_count.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           os.path.join(tmp, "snappy_hip.hip"), "-o", out])
    shutil.rmtree(tmp)
    return out


def main():
    lib = build()
    import numpy as np, torch
    import silesia_mix
    import snappy_hip_binding as shb
    shb.LIB_PATH = lib
    L = shb.lib()
    L.snappy_hip_debug_lines.argtypes = [ctypes.c_void_p, ctypes.c_int]
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = mib << 20
    xs = np.frombuffer(open(os.path.join(ROOT, "tests/golden/xml.snappy"), "rb").read(), dtype=np.uint8).copy()
    st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
    unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
    d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
    ws = shb.CompressWorkspace(n, 32768)
    os.environ["SNAPPY_HIP_LDS_WAVES"] = "0"
    os.environ["SNAPPY_HIP_K1_STREAM"] = "0"
    shb.compress_blocks(d_in, n, ws); torch.cuda.synchronize()
    L.snappy_hip_debug_lines(None, 1)
    shb.compress_blocks(d_in, n, ws); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    L.snappy_hip_debug_lines(out, 0)
    c = list(out)
    w = n / 64.0
    print(f"global-table kernel alone, bulk form, {mib} MiB Silesia-mix = {int(w)} windows of 64 input bytes; per window:")
    print(f"   gathers issued                                   {c[5] / w:6.2f}")
    print(f"   slot reads that pass the LDS filter (1 line each) {c[0] / w:6.2f}")
    print(f"   candidate fetches (tag allows a hit)              {c[1] / w:6.2f}   of which 28-byte ones {c[6] / w:6.2f}")
    print(f"   ... that straddle a second 64-byte line           {c[2] / w:6.2f}")
    print(f"   table stores (1 line written each)                {c[3] / w:6.2f}")
    print(f"   counted read lines  = {(c[0] + c[1] + c[2]) / w:6.2f}  (+ 1.0 input stream)")
    print(f"   counted write lines = {c[3] / w:6.2f}  (+ ~0.5 compressed output)")


if __name__ == "__main__":
    main()
