#!/usr/bin/env python3
"""Build pim-compression_amd/libsnappy_hip_prof.so: a copy of the product sources with s_memtime probes around the phases of
K1's bulk parse (--k1) or K2's element loop (--k2), accumulated into a __device__ array that tools/prof_phases.py /
tools/prof_phases_k2.py read through snappy_hip_debug_prof().  The probes are inserted by exact-text replacement, so this
script has to follow the kernel source; every replacement asserts that it applied.  Not a product build.
Usage: python tools/make_probe_build.py --k1 | --k2"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pim-compression_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

PROLOG = ("namespace snappy_hip {\n__device__ unsigned long long g_prof[32];\n"
          "#define PROF_T() ((unsigned long long)__builtin_readcyclecounter())\n"
          "#define PROF_W() (__builtin_amdgcn_s_waitcnt(0), (unsigned long long)__builtin_readcyclecounter())\n")
EXPORT = ('#include "snappy_kernels.hpp"\nextern "C" int snappy_hip_debug_prof(unsigned long long* out, int reset) {\n'
          '    if (reset) { unsigned long long z[32] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(snappy_hip::g_prof), z, sizeof(z)); }\n'
          '    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(snappy_hip::g_prof), 32 * sizeof(unsigned long long));\n}\n')


class Patcher:
    def __init__(self, text, start_marker):
        a = text.index(start_marker)
        self.head, self.body = text[:a], text[a:]

    def rep(self, old, new):
        assert old in self.body, "probe anchor not found:\n" + old
        self.body = self.body.replace(old, new, 1)

    def text(self):
        return self.head + self.body


def probes_k1(src):
    p = Patcher(src, "__device__ __forceinline__ void compress_one_block_bulk(")
    p.rep("        State st;\n        uint32_t ip = 1;      // :305\n",
          "        State st;\n        uint32_t ip = 1;      // :305\n        const unsigned long long pb0 = PROF_T();\n"
          "        unsigned long long t_g1 = 0, t_g2 = 0, n_g = 0, t_seg = 0, n_seg = 0, t_ce = 0, t_single = 0, n_single = 0, t_win = 0,"
          " n_win = 0, t_drain = 0, t_res = 0, n_res = 0, t_lit1 = 0, n_lit1 = 0;\n")
    p.rep("            if (win.ensure(ip, lane)) st.invalidate();\n            uint32_t r = ip - win.base;\n"
          "            if (r >= uni(st.cov_end)) st.template gather<true, true>(table, win, dup_scratch, r, kChunk, lane, n);\n",
          "            { const unsigned long long w0 = PROF_T(); if (win.ensure(ip, lane)) { st.invalidate(); SNAPPY_PIN(win.x0);"
          " t_win += PROF_T() - w0; n_win++; } }\n            uint32_t r = ip - win.base;\n"
          "            if (r >= uni(st.cov_end)) { const unsigned long long d0 = PROF_T(); const unsigned long long g0 = PROF_W();"
          " t_drain += g0 - d0; st.template gather<true, true>(table, win, dup_scratch, r, kChunk, lane, n);"
          " const unsigned long long g1 = PROF_T(); SNAPPY_PIN(st.ent); const unsigned long long g2 = PROF_W();"
          " t_g1 += g1 - g0; t_g2 += g2 - g1; n_g++; }\n")
    p.rep("                const uint32_t r0 = r;\n                unsigned long long H = 0, COV = 0;\n",
          "                const uint32_t r0 = r;\n                unsigned long long H = 0, COV = 0;\n"
          "                const unsigned long long s0t = PROF_T();\n")
    p.rep("                    bool hit_r;\n                    uint32_t cand_r, ext_r;\n",
          "                    const unsigned long long rr0 = PROF_T(); n_res++;\n                    bool hit_r;\n"
          "                    uint32_t cand_r, ext_r;\n")
    p.rep("                    stopm &= ~(1ull << r);\n                    inter = hit_r ?",
          "                    t_res += PROF_T() - rr0;\n                    stopm &= ~(1ull << r);\n                    inter = hit_r ?")
    p.rep("                ip = win.base + r;\n                skip = 64u - B;\n",
          "                ip = win.base + r;\n                skip = 64u - B;\n"
          "                const unsigned long long s1t = PROF_T(); t_seg += s1t - s0t; n_seg++;\n")
    p.rep("                        op = emit_literal_windowed(dst, op, blk, next_emit, p0 - next_emit, win.base, win.x0, lane);\n"
          "                    }\n                    const unsigned long long LIT",
          "                        const unsigned long long l0 = PROF_T(); n_lit1++;\n"
          "                        op = emit_literal_windowed(dst, op, blk, next_emit, p0 - next_emit, win.base, win.x0, lane);\n"
          "                        t_lit1 += PROF_T() - l0;\n                    }\n                    const unsigned long long LIT")
    p.rep("                if (done) break;\n                if (why == 0 && r > kWave) {",
          "                t_ce += PROF_T() - s1t;\n                if (done) break;\n                if (why == 0 && r > kWave) {")
    p.rep("            uint32_t cand = 0, ext = 0, sat = 8;\n            bool hit;\n",
          "            const unsigned long long q0 = PROF_T(); n_single++;\n            uint32_t cand = 0, ext = 0, sat = 8;\n"
          "            bool hit;\n")
    p.rep("            if (!hit) {\n                ip += step;\n                ++skip;\n                continue;\n            }",
          "            if (!hit) {\n                ip += step;\n                ++skip;\n                t_single += PROF_T() - q0;\n"
          "                continue;\n            }")
    p.rep("            st.inserted |= 1ull << (ip - 1 - win.base);\n            skip = 31;\n        }\n    }\n",
          "            st.inserted |= 1ull << (ip - 1 - win.base);\n            skip = 31;\n            t_single += PROF_T() - q0;\n"
          "        }\n        if (lane == 0) {\n"
          "            atomicAdd(&g_prof[0], PROF_T() - pb0); atomicAdd(&g_prof[1], 1ull);\n"
          "            atomicAdd(&g_prof[2], t_g1); atomicAdd(&g_prof[3], t_g2); atomicAdd(&g_prof[4], n_g);\n"
          "            atomicAdd(&g_prof[5], t_seg); atomicAdd(&g_prof[6], n_seg); atomicAdd(&g_prof[7], t_ce);\n"
          "            atomicAdd(&g_prof[8], t_single); atomicAdd(&g_prof[9], n_single); atomicAdd(&g_prof[10], t_win);"
          " atomicAdd(&g_prof[11], n_win); atomicAdd(&g_prof[12], t_drain);\n"
          "            atomicAdd(&g_prof[13], t_res); atomicAdd(&g_prof[14], n_res); atomicAdd(&g_prof[17], t_lit1);"
          " atomicAdd(&g_prof[18], n_lit1);\n        }\n    }\n")
    return p.text()


def probes_k2(src):
    p = Patcher(src, "template <bool kLdsWindow>\n__global__ __launch_bounds__(64) void decompress_blocks_kernel(")
    p.rep("        uint32_t g = 0;             // window base, multiple of 64 (compressed offset)\n",
          "        const unsigned long long pb0 = PROF_T();\n        unsigned long long t_win = 0, n_win = 0, t_lit = 0, n_lit = 0,"
          " t_copy = 0, n_copy = 0, n_ovl = 0, t_longlit = 0, n_longlit = 0, t_fast = 0;\n"
          "        uint32_t g = 0;             // window base, multiple of 64 (compressed offset)\n")
    p.rep("            if (!have_window || cp >= g + 128) {\n",
          "            const unsigned long long w0t = PROF_T();\n            if (!have_window || cp >= g + 128) {\n")
    p.rep("            const uint32_t wend = (csz < g + 64) ? csz : g + 64;\n",
          "            const uint32_t wend = (csz < g + 64) ? csz : g + 64;\n            SNAPPY_PIN(meta); t_win += PROF_T() - w0t; n_win++;\n")
    p.rep("                    k2_fast_elements(meta, offv, (uint32_t)w0, lane, win, g, wend, out_len, cp, op);\n",
          "                    const unsigned long long f0t = PROF_T();\n"
          "                    k2_fast_elements(meta, offv, (uint32_t)w0, lane, win, g, wend, out_len, cp, op);\n"
          "                    t_fast += PROF_T() - f0t;\n")
    p.rep("                const uint32_t s = cp - g;                               // lane that holds this element's tag\n",
          "                const unsigned long long e0t = PROF_T();\n"
          "                const uint32_t s = cp - g;                               // lane that holds this element's tag\n")
    p.rep("                    __builtin_amdgcn_wave_barrier();\n                    cp += hdr + len;\n                    op += len;\n"
          "                    continue;\n",
          "                    __builtin_amdgcn_wave_barrier();\n                    { const unsigned long long dt = PROF_T() - e0t;"
          " if (s + hdr + len <= 128) { t_lit += dt; n_lit++; } else { t_longlit += dt; n_longlit++; } }\n"
          "                    cp += hdr + len;\n                    op += len;\n                    continue;\n")
    p.rep("                cp += hdr;\n                op += len;\n            }\n        }\n",
          "                t_copy += PROF_T() - e0t; n_copy++; if (off < len) n_ovl++;\n                cp += hdr;\n"
          "                op += len;\n            }\n        }\n        if (lane == 0) {\n"
          "            atomicAdd(&g_prof[0], PROF_T() - pb0); atomicAdd(&g_prof[1], 1ull);\n"
          "            atomicAdd(&g_prof[2], t_win); atomicAdd(&g_prof[3], n_win); atomicAdd(&g_prof[4], t_lit); atomicAdd(&g_prof[5], n_lit);\n"
          "            atomicAdd(&g_prof[6], t_copy); atomicAdd(&g_prof[7], n_copy); atomicAdd(&g_prof[8], n_ovl);"
          " atomicAdd(&g_prof[9], t_longlit); atomicAdd(&g_prof[10], n_longlit); atomicAdd(&g_prof[11], t_fast);\n        }\n")
    return p.text()


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "--k1"
    tmp = tempfile.mkdtemp(prefix="snappy_probe_")
    src = open(os.path.join(CSRC, "snappy_kernels.hpp")).read().replace("namespace snappy_hip {", PROLOG, 1)
    src = probes_k2(src) if which == "--k2" else probes_k1(src)
    open(os.path.join(tmp, "snappy_kernels.hpp"), "w").write(src)
    host = open(os.path.join(CSRC, "snappy_hip.hip")).read()
    host = host.replace('#include "snappy_kernels.hpp"', EXPORT, 1)
    host = host.replace('"../../include/snappy_hip.h"', '"%s"' % os.path.join(ROOT, "include", "snappy_hip.h"))
    open(os.path.join(tmp, "snappy_hip.hip"), "w").write(host)
    out = os.path.join(ROOT, "pim-compression_amd", "libsnappy_hip_prof.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC,   # the other headers
                           os.path.join(tmp, "snappy_hip.hip"), "-o", out])
    shutil.rmtree(tmp)
    print(out)


if __name__ == "__main__":
    main()
