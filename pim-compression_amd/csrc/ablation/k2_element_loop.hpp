// k2_element_loop.hpp -- round 1's K2: the element-at-a-time decoder (output window in global memory, or staged whole in LDS).
// Ablation code: compiled only with -DSNAPPY_ABLATION (tools/build_ablation.py, the CPU emulator); the product's decoder is
// the per-window batch form in snappy_kernels.hpp.  Included from there, inside no namespace.
#pragma once

namespace snappy_hip {

// Pre-decode the element that would start with the 8 bytes `w` at compressed offset `pos` of a block of
// `csz` compressed bytes (snappy_decompress.c:242-284 field rules), branch-free.
//   meta = type | hdr << 2 | out_len << 8 ; meta == 0 marks an element that cannot be valid here
//   (over-long literal, header or literal payload running past the block's compressed size).
//   off  = copy offset (0 for literals).
__device__ __forceinline__ void predecode(uint64_t w, uint32_t pos, uint32_t csz, uint32_t& meta, uint32_t& off)
{
    const uint32_t tag = (uint32_t)w & 0xff;
    const uint32_t type = tag & 3;
    const uint32_t v = tag >> 2;
    const uint32_t next4 = (uint32_t)(w >> 8);
    // literal (:244-256, :64-74)
    const uint32_t nb = (v >= 60) ? v - 59 : 0;                              // extra length bytes
    const uint32_t raw = next4 & (uint32_t)((1ull << (8 * nb)) - 1);
    const uint32_t lit_len = (v < 60) ? v + 1 : ((raw < 65536u) ? raw + 1 : 0);   // blocks are < 64 KiB
    // copies (:264-283, :83-133)
    const uint32_t c1_off = ((tag >> 5) << 8) | (next4 & 0xff);
    const uint32_t hdr = (type == 0) ? 1 + nb : ((type == 3) ? 5u : type + 1);
    const uint32_t olen = (type == 0) ? lit_len : ((type == 1) ? (v & 7) + 4 : v + 1);
    off = (type == 0) ? 0u : ((type == 1) ? c1_off : ((type == 2) ? (next4 & 0xffff) : next4));
    const uint32_t consumed = hdr + ((type == 0) ? olen : 0);
    const bool ok = (olen != 0) && (pos + consumed <= csz);
    // classes for the hand-scheduled element loop (k2_fast_elements): bit 5 = a copy it may take (offset != 0, no overlap,
    // <= 63 bytes), bit 6 = a literal whose payload lies inside this 64-byte granule; the offset-vs-output checks stay dynamic
    const uint32_t fast_copy = (type != 0 && off != 0 && off >= olen && olen <= 63u) ? 32u : 0u;
    const uint32_t fast_lit = (type == 0 && (pos & 63u) + consumed <= 64u) ? 64u : 0u;
    meta = ok ? (type | (hdr << 2) | fast_copy | fast_lit | (olen << 8)) : 0;
}

// kLdsWindow = true : the decoded block is staged in LDS (block_size bytes, ~4 blocks/CU) and written out at the end;
//   ~2.7x faster per wavefront (LDS back-references) but only 4 such wavefronts fit per CU.
// kLdsWindow = false: the decoded block is written straight to its place in global memory and back-references
//   are read from there (vector memory operations of one wavefront complete in issue order on gfx9-family
//   hardware, so a load issued after a store to the same bytes observes it); no LDS, 32 waves/CU.
// K2's element loop for the common elements, hand-scheduled for gfx950 (the compiler's version of this loop spends ~40
// instructions per copy and ~35 per literal; the static conditions are folded into two class bits by predecode()).
// Handles, for elements that start in the current 64-byte window: a literal whose payload lies inside the window (one
// exec-masked byte store from the window registers) and a non-overlapping copy of up to 63 bytes (one exec-masked byte
// load and, later, store; same-wave vector memory operations complete in order, so a later load sees an earlier store).
// Returns with s/op advanced as soon as it meets anything else -- an invalid or truncated element, a literal that runs
// into the next window, an overlapping or 64-byte copy -- and the C++ loop takes that element.
// The CPU emulator compiles an empty body: there the C++ loop does everything, which is also the specification.
// Copies overlap: a copy only ISSUES its load (into one of six data registers) and records where its bytes go; literals
// store at once; the deferred stores are made together -- one wait for the whole batch instead of one memory round trip
// per copy -- when the batch is full, when a copy wants bytes at or above the first deferred destination (`lo`: everything
// from there on may still be missing), at the end of the window, or before any element the loop does not take.  PMC on the
// one-copy-at-a-time loop: wavefronts spent 74 % of their cycles waiting on memory, 16 % executing; with the batches the
// loop is bound by the scalar unit (one SALU instruction per SIMD every four cycles), so the per-element fields come as
// separate registers read with v_readlane (a VALU slot, of which there are plenty) and the two classes as lane masks:
// 12 SALU instructions per copy (incl. its share of the batch store) and 8 per literal, down from 18 and 15.
//   lenv / advv / xlv / offv: per window lane, the element's output length, its compressed advance, the window lane of a
//   literal's first payload byte, a copy's offset.  cm / lm: lanes that start a copy / literal of the classes above.
//   s = cp - g (in/out), wlim = wend - g.
#define K2_TOP(I)                                                                                    \
    "k2_s" I "_%=:\n"                                                                                \
    "  s_cmp_ge_u32 %[s], %[wlim]\n"                                                                 \
    "  s_cbranch_scc1 k2_leave" I "_%=\n"                                                            \
    "  s_bitcmp1_b64 %[cm], %[s]\n"                                                                  \
    "  s_cbranch_scc1 k2_c" I "_%=\n"                                                                \
    "  s_bitcmp1_b64 %[lm], %[s]\n"                                                                  \
    "  s_cbranch_scc0 k2_leave" I "_%=\n"         /* neither class (or rejected by predecode) */     \
    /* ---- literal inside the granule: stored at once ---- */                                       \
    "  v_readlane_b32 %[len], %[lenv], %[s]\n"                                                       \
    "  v_readlane_b32 %[x], %[xlv], %[s]\n"       /* payload start, as a window lane */              \
    "  s_bfm_b64 exec, %[len], %[x]\n"            /* len <= 63 here */                               \
    "  s_sub_u32 %[m], %[op], %[x]\n"                                                                \
    "  v_add_u32 %[va], %[m], %[lane]\n"                                                             \
    "  s_add_u32 %[op], %[op], %[len]\n"                                                             \
    "  s_cmp_gt_u32 %[op], %[outlen]\n"                                                              \
    "  s_cbranch_scc1 k2_undo" I "_%=\n"          /* would overrun the block's output */             \
    "  global_store_byte %[va], %[w0], %[win]\n"                                                     \
    "  v_readlane_b32 %[adv], %[advv], %[s]\n"                                                       \
    "  s_add_u32 %[s], %[s], %[adv]\n"                                                               \
    "  s_branch k2_s" I "_%=\n"                                                                      \
    /* ---- copy without overlap, <= 63 bytes ---- */                                                \
    "k2_c" I "_%=:\n"                                                                                \
    "  v_readlane_b32 %[len], %[lenv], %[s]\n"                                                       \
    "  v_readlane_b32 %[off], %[offv], %[s]\n"                                                       \
    "  s_cmp_gt_u32 %[off], %[op]\n"                                                                 \
    "  s_cbranch_scc1 k2_leave" I "_%=\n"         /* reaches before the block start */               \
    "  s_sub_u32 %[x], %[op], %[off]\n"
#define K2_NEEDS_DEFERRED(I)                      /* source end above the first deferred destination */\
    "  s_add_u32 %[m], %[x], %[len]\n"                                                               \
    "  s_cmp_gt_u32 %[m], %[lo]\n"                                                                   \
    "  s_cbranch_scc1 k2_again" I "_%=\n"
#define K2_ISSUE(I, VD, VA, PL)                                                                      \
    "  s_bfm_b64 exec, %[len], 0\n"                                                                  \
    "  v_add_u32 %[va], %[x], %[lane]\n"                                                             \
    "  v_add_u32 " VA ", %[op], %[lane]\n"        /* where the bytes go, kept until the batch store */\
    "  s_add_u32 %[op], %[op], %[len]\n"                                                             \
    "  s_cmp_gt_u32 %[op], %[outlen]\n"                                                              \
    "  s_cbranch_scc1 k2_undo" I "_%=\n"                                                             \
    "  global_load_ubyte " VD ", %[va], %[win]\n"                                                    \
    "  s_mov_b32 " PL ", %[len]\n"                                                                   \
    "  v_readlane_b32 %[adv], %[advv], %[s]\n"                                                       \
    "  s_add_u32 %[s], %[s], %[adv]\n"
#define K2_STATE(I, NEXT, VD, VA, PL)             /* I copies deferred, I >= 1 */                    \
    K2_TOP(I)                                                                                        \
    K2_NEEDS_DEFERRED(I)                                                                             \
    K2_ISSUE(I, VD, VA, PL)                                                                          \
    "  s_branch k2_s" NEXT "_%=\n"                                                                   \
    "k2_undo" I "_%=:\n"                                                                             \
    "  s_sub_u32 %[op], %[op], %[len]\n"                                                             \
    "k2_leave" I "_%=:\n"                                                                            \
    "  s_mov_b32 %[ret], 1\n"                                                                        \
    "  s_waitcnt vmcnt(0)\n"                                                                         \
    "  s_branch k2_f" I "_%=\n"                                                                      \
    "k2_again" I "_%=:\n"                                                                            \
    "  s_mov_b32 %[ret], 0\n"                                                                        \
    "  s_waitcnt vmcnt(0)\n"                                                                         \
    "  s_branch k2_f" I "_%=\n"
#define K2_STORE_DEFERRED(I, VD, VA, PL)                                                             \
    "k2_f" I "_%=:\n"                                                                                \
    "  s_bfm_b64 exec, " PL ", 0\n"                                                                  \
    "  global_store_byte " VA ", " VD ", %[win]\n"

__device__ __forceinline__ void k2_fast_elements(uint32_t lenv, uint32_t advv, uint32_t xlv, uint32_t offv, uint64_t cm, uint64_t lm,
                                                 uint32_t w0_lo, uint32_t lane, uint8_t* win, uint32_t wlim, uint32_t out_len,
                                                 uint32_t& s, uint32_t& op)
{
#ifndef SNAPPY_EMU
    uint32_t m, len, x, off, adv, lo, ret;
    uint32_t pl0, pl1, pl2, pl3, pl4, pl5;
    uint32_t va, vd0, vd1, vd2, vd3, vd4, vd5, va0, va1, va2, va3, va4, va5;
    asm volatile(
        // ---- nothing deferred ----
        K2_TOP("0")
        "  s_mov_b32 %[lo], %[op]\n"
        K2_ISSUE("0", "%[vd0]", "%[va0]", "%[pl0]")
        "  s_branch k2_s1_%=\n"
        "k2_undo0_%=:\n"
        "  s_sub_u32 %[op], %[op], %[len]\n"
        "k2_leave0_%=:\n"
        "  s_branch k2_done_%=\n"
        K2_STATE("1", "2", "%[vd1]", "%[va1]", "%[pl1]")
        K2_STATE("2", "3", "%[vd2]", "%[va2]", "%[pl2]")
        K2_STATE("3", "4", "%[vd3]", "%[va3]", "%[pl3]")
        K2_STATE("4", "5", "%[vd4]", "%[va4]", "%[pl4]")
        K2_STATE("5", "6", "%[vd5]", "%[va5]", "%[pl5]")
        // ---- all six registers in use: store them, then carry on with nothing deferred ----
        "k2_s6_%=:\n"
        "  s_mov_b32 %[ret], 0\n"
        "  s_waitcnt vmcnt(0)\n"
        K2_STORE_DEFERRED("6", "%[vd5]", "%[va5]", "%[pl5]")
        K2_STORE_DEFERRED("5", "%[vd4]", "%[va4]", "%[pl4]")
        K2_STORE_DEFERRED("4", "%[vd3]", "%[va3]", "%[pl3]")
        K2_STORE_DEFERRED("3", "%[vd2]", "%[va2]", "%[pl2]")
        K2_STORE_DEFERRED("2", "%[vd1]", "%[va1]", "%[pl1]")
        K2_STORE_DEFERRED("1", "%[vd0]", "%[va0]", "%[pl0]")
        "  s_cmp_eq_u32 %[ret], 0\n"
        "  s_cbranch_scc1 k2_s0_%=\n"
        "k2_done_%=:\n"
        "  s_mov_b64 exec, -1\n"                      // nothing inside the loop depends on exec beyond what it sets itself
        : [s] "+s"(s), [op] "+s"(op), [m] "=&s"(m), [len] "=&s"(len), [x] "=&s"(x), [off] "=&s"(off), [adv] "=&s"(adv),
          [lo] "=&s"(lo), [ret] "=&s"(ret), [pl0] "=&s"(pl0), [pl1] "=&s"(pl1), [pl2] "=&s"(pl2), [pl3] "=&s"(pl3), [pl4] "=&s"(pl4),
          [pl5] "=&s"(pl5), [va] "=&v"(va), [vd0] "=&v"(vd0), [vd1] "=&v"(vd1), [vd2] "=&v"(vd2), [vd3] "=&v"(vd3), [vd4] "=&v"(vd4),
          [vd5] "=&v"(vd5), [va0] "=&v"(va0), [va1] "=&v"(va1), [va2] "=&v"(va2), [va3] "=&v"(va3), [va4] "=&v"(va4), [va5] "=&v"(va5)
        : [lenv] "v"(lenv), [advv] "v"(advv), [xlv] "v"(xlv), [offv] "v"(offv), [w0] "v"(w0_lo), [lane] "v"(lane), [win] "s"(win),
          [cm] "s"(cm), [lm] "s"(lm), [wlim] "s"(wlim), [outlen] "s"(out_len)
        : "scc", "memory");
#else
    (void)lenv; (void)advv; (void)xlv; (void)offv; (void)cm; (void)lm; (void)w0_lo; (void)lane; (void)win; (void)wlim;
    (void)out_len; (void)s; (void)op;
#endif
}
#undef K2_TOP
#undef K2_NEEDS_DEFERRED
#undef K2_ISSUE
#undef K2_STATE
#undef K2_STORE_DEFERRED

// amdgpu_num_sgpr: measured on gfx950, the 81st SGPR costs the eighth wavefront per SIMD (8.3 -> 9.0 ms per container).
template <bool kLdsWindow>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) void decompress_blocks_element_kernel(const K2Batch w, uint32_t block_size,
                                                                                                    uint32_t* next_block)
{
    HIP_DYNAMIC_SHARED(uint8_t, lds_win)   // kLdsWindow: block_size rounded up to 16; dynamic LDS starts 16-byte aligned
    const uint32_t lane = threadIdx.x;
    const uint32_t num_blocks = w.first_block[w.count];

    // Persistent: wavefronts draw blocks from *next_block (zeroed per launch), so the LDS-window form and the
    // global-window form can run concurrently on one stream of blocks and balance themselves.
    for (;;) {
        uint32_t drawn = 0;
        if (lane == 0) drawn = atomicAdd(next_block, 1u);
        const uint32_t gb = uni(drawn);
        if (gb >= num_blocks) break;
        uint32_t c = 0;
        while (c + 1 < w.count && gb >= w.first_block[c + 1]) ++c;
        const uint32_t b = gb - w.first_block[c];
        const uint8_t* __restrict__ stream = w.stream[c];
        const uint64_t* len_dev = w.stream_len_dev[c];
        const uint64_t stream_len = len_dev ? uld64(reinterpret_cast<const uint8_t*>(len_dev)) : w.stream_len[c];
        const uint64_t* __restrict__ block_offsets = w.block_offsets[c];
        const uint64_t total_len = w.total_len[c];
        uint32_t* __restrict__ status = w.status[c];
        const uint64_t ostart = (uint64_t)b * block_size;
        const uint64_t oleft = total_len - ostart;
        const uint32_t out_len = (oleft < block_size) ? (uint32_t)oleft : block_size;
        uint8_t* dst = w.out[c] + ostart;
        uint8_t* win = kLdsWindow ? lds_win : dst;

        uint32_t st = kBlockOk;
        const uint64_t at = block_offsets[b];
        uint32_t csz = 0;
        if (at + 4 > stream_len) {
            st = kBlockInvalid;
        } else {
            csz = uld32(stream + at);                                    // snappy_decompress.c:229-230
            if (at + 4 + (uint64_t)csz > stream_len) st = kBlockInvalid;
        }
        const uint8_t* __restrict__ src = stream + at + 4;
        const uint64_t avail = (st == kBlockOk) ? stream_len - (at + 4) : 0;

        uint32_t g = 0;             // window base, multiple of 64 (compressed offset)
        uint32_t cp = 0, op = 0;    // compressed / output cursors
        uint64_t w0 = 0;
        WindowLoad next = {0, 64};  // prefetch of the following 64 bytes (W1), shift applied at use
        bool have_window = false;
        while (st == kBlockOk && cp < csz) {                             // one iteration per 64-byte window
            if (!have_window || cp >= g + 128) {
                g = cp & ~63u;
                const WindowLoad cur = window_issue(src, (uint64_t)g + lane, avail);
                w0 = window_value(cur);
                have_window = true;
            } else {                                                     // cp in [g+64, g+128): slide by 64
                g += 64;
                w0 = window_value(next);
            }
            uint32_t meta = 0, offv = 0;
            predecode(w0, g + lane, csz, meta, offv);
            // issue the prefetch only after w0 has been consumed, so the wait for w0 cannot cover it
            __builtin_amdgcn_sched_barrier(0);
            next = window_issue(src, (uint64_t)g + 64 + lane, avail);    // stays in flight during this window
            const uint32_t wend = (csz < g + 64) ? csz : g + 64;
#ifndef SNAPPY_EMU
            // operands of the hand-scheduled loop: the pre-decoded fields one register each, the two classes as lane masks
            const uint32_t e_hdr = (meta >> 2) & 7u, e_len = meta >> 8;
            const uint32_t advv = e_hdr + ((meta & 3u) ? 0u : e_len);
            const uint32_t xlv = lane + e_hdr;
            const uint64_t copy_lanes = kLdsWindow ? 0 : __builtin_amdgcn_ballot_w64((meta & 32u) != 0);
            const uint64_t literal_lanes = kLdsWindow ? 0 : __builtin_amdgcn_ballot_w64((meta & 64u) != 0);
#endif

            while (cp < wend) {                                          // :232, elements that start in this window
#ifndef SNAPPY_EMU
                if (!kLdsWindow) {                                       // the common elements, hand-scheduled
                    uint32_t rel = cp - g;
                    k2_fast_elements(e_len, advv, xlv, offv, copy_lanes, literal_lanes, (uint32_t)w0, lane, win, wend - g, out_len,
                                     rel, op);
                    cp = g + rel;
                    if (cp >= wend) break;
                }
#endif
                const uint32_t s = cp - g;                               // lane that holds this element's tag
                const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)meta, (int)s);
                const uint32_t type = m & 3, hdr = (m >> 2) & 7, len = m >> 8;
                if (m == 0 || op + len > out_len) {                      // m == 0: rejected by predecode
                    st = kBlockInvalid;
                    break;
                }
                if (type == 0) {                                         // literal, :244-256
                    const uint32_t rel = s + hdr;                        // payload start relative to g
                    if (rel + len <= 128) {
                        // payload bytes are byte 0 of window lanes rel .. rel+len-1
                        if (lane >= rel && lane < rel + len) win[op + lane - rel] = (uint8_t)w0;
                        if (rel + len > 64) {                            // spills into W1 (wave-uniform)
                            WindowLoad nx = next;
                            SNAPPY_PIN(nx.shift);                        // first use of the prefetch: wait here, not earlier
                            const uint64_t w1 = window_value(nx);
                            if (lane + 64 >= rel && lane + 64 < rel + len) win[op + lane + 64 - rel] = (uint8_t)w1;
                        }
                    } else {
                        const uint8_t* __restrict__ p = src + cp + hdr;  // long literal: straight from memory
                        uint32_t i = 4 * lane;
                        for (; i + 4 <= len; i += 4 * kWave) st32(win + op + i, ld32(p + i));
                        for (; i < len; ++i) win[op + i] = p[i];
                    }
                    __builtin_amdgcn_wave_barrier();
                    cp += hdr + len;
                    op += len;
                    continue;
                }
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)offv, (int)s);
                // strict: the source must lie inside this block's own output (cf. :167-173)
                if (off == 0 || off > op) {
                    st = kBlockInvalid;
                    break;
                }
                // :174-181 forward byte copy == periodic replication of the last `off` bytes.
                // Source bytes are [op-off, op-off+min(len,off)); they must not be pending in the batch.
                {   // :174-181 forward byte copy == periodic replication of the last `off` bytes
                    uint32_t src_idx = lane;
                    if (off < len) {                                     // overlap: lane % off (lane < 64, off < 64)
                        const uint32_t q = (lane * kRecip16[off]) >> 16;
                        src_idx = lane - q * off;
                    }
                    // every source byte lies before `op` and every destination at or after it, so the lanes of
                    // one element never depend on each other: one predicated load+store, one barrier afterwards
                    if (lane < len) win[op + lane] = win[op - off + src_idx];
                    __builtin_amdgcn_wave_barrier();
                }
                cp += hdr;
                op += len;
            }
        }
        if (st == kBlockOk && (op != out_len || cp != csz)) st = kBlockInvalid;

        // write-out: LDS -> global, 16 B per lane when the destination allows it
        __syncthreads();
        if (kLdsWindow && st == kBlockOk) {
            if ((((uintptr_t)dst) & 15) == 0) {
                const uint32_t body = out_len & ~15u;
                const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(lds_win);
                uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
                for (uint32_t i = lane; i < body / 16; i += kWave) d4[i] = w4[i];
                for (uint32_t i = body + lane; i < out_len; i += kWave) dst[i] = lds_win[i];
            } else {
                for (uint32_t i = lane; i < out_len; i += kWave) dst[i] = lds_win[i];
            }
        }
        if (lane == 0) status[b] = st;
        __syncthreads();
    }
}

}  // namespace snappy_hip
