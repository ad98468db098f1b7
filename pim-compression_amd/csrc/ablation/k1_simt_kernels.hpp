// k1_simt_kernels.hpp -- K1 experiments that put several blocks on one wavefront: one block per lane, and four blocks per wavefront (16-lane groups).
// Ablation code: compiled only with -DSNAPPY_ABLATION (tools/build_ablation.py -> libsnappy_hip_ablation.so); the product
// library contains ONE K1 pair (bulk parse: global-table + LDS-table kernels), the two-wavefront LDS form, and one K2.
// Every form here is bit-exact with the product (tests/test_gpu_ablation.py, tests/test_emulated_kernels.py).
#pragma once

namespace snappy_hip {

// ---------------------------------------------------------------------------
// K1, lane-per-block form: every LANE owns one Snappy block (64 blocks per wavefront) and runs the
// sequential parse as ordinary SIMT code -- all VALU, no wave-uniform scalar chain, so one
// wave-instruction advances up to 64 parses.  Each lane's u16 hash table (<= 32 KiB) lives in a global
// scratch; the dependent table -> candidate loads are hidden by the other resident waves.
// Same bytes as compress_one_block (snappy_compress.c:284-413).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_emit_literal(uint8_t* __restrict__ dst, uint32_t op,
                                                      const uint8_t* __restrict__ src, uint32_t len)
{
    const uint32_t n = len - 1;                                  // snappy_compress.c:202-225
    if (n < 60) {
        dst[op++] = (uint8_t)(n << 2);
    } else if (n < 256) {
        dst[op++] = (uint8_t)(60 << 2);
        dst[op++] = (uint8_t)n;
    } else if (n < 65536) {
        dst[op++] = (uint8_t)(61 << 2);
        dst[op++] = (uint8_t)n;
        dst[op++] = (uint8_t)(n >> 8);
    } else {
        dst[op++] = (uint8_t)(62 << 2);
        dst[op++] = (uint8_t)n;
        dst[op++] = (uint8_t)(n >> 8);
        dst[op++] = (uint8_t)(n >> 16);
    }
    uint32_t i = 0;
    for (; i + 4 <= len; i += 4) st32(dst + op + i, ld32(src + i));
    for (; i < len; ++i) dst[op + i] = src[i];
    return op + len;
}

__device__ __forceinline__ uint32_t lane_emit_copy(uint8_t* __restrict__ dst, uint32_t op, uint32_t off, uint32_t len)
{
    for (;;) {                                                   // snappy_compress.c:234-272
        uint32_t piece = len;
        if (len >= 68) piece = 64;
        else if (len > 64) piece = 60;
        if (piece < 12 && off < 2048) {
            dst[op++] = (uint8_t)(1 + ((piece - 4) << 2) + ((off >> 8) << 5));
            dst[op++] = (uint8_t)off;
        } else {
            dst[op++] = (uint8_t)(2 + ((piece - 1) << 2));
            dst[op++] = (uint8_t)off;
            dst[op++] = (uint8_t)(off >> 8);
        }
        len -= piece;
        if (len == 0) return op;
    }
}

__global__ __launch_bounds__(64) void compress_blocks_lane_kernel(const uint8_t* __restrict__ in, uint64_t in_len,
                                                                  uint32_t block_size, uint8_t* __restrict__ slots,
                                                                  uint32_t slot_stride, uint32_t* __restrict__ block_bytes,
                                                                  uint32_t num_blocks, uint16_t* __restrict__ tables,
                                                                  uint32_t lanes_per_block)
{
    // lanes_per_block > 1 replicates each block's (identical) work over a lane group: fewer blocks per wave,
    // same addresses within a group (coalesced), more waves for the same number of blocks.
    const uint32_t b = (blockIdx.x * 64 + threadIdx.x) / lanes_per_block;
    if (b >= num_blocks) return;
    const uint64_t start = (uint64_t)b * block_size;
    const uint64_t left = in_len - start;
    const uint32_t n = (left < block_size) ? (uint32_t)left : block_size;
    const uint8_t* __restrict__ blk = in + start;
    uint8_t* __restrict__ dst = slots + (uint64_t)b * slot_stride;
    uint16_t* __restrict__ table = tables + (size_t)b * kMaxTableEntries;

    const uint32_t ts = table_entries_for(n);                    // snappy_compress.c:139-146
    const uint32_t shift = (uint32_t)__builtin_clz(ts) + 1;      // :288
    {
        uint4* t = reinterpret_cast<uint4*>(table);
        for (uint32_t i = 0; i < ts / 8; ++i) t[i] = make_uint4(0, 0, 0, 0);
    }
    uint32_t op = 4, next_emit = 0;
    if (n >= kInputMargin) {
        const uint32_t limit = n - kInputMargin;
        uint32_t ip = 1;
        uint32_t cur = ld32(blk + ip);
        for (;;) {
            uint32_t skip = 32, cand;
            bool out_of_input = false;
            for (;;) {                                           // :336-348
                const uint32_t h = (cur * kHashMul) >> shift;
                const uint32_t next_ip = ip + (skip++ >> 5);
                if (next_ip > limit) {
                    out_of_input = true;
                    break;
                }
                const uint32_t nxt = ld32(blk + next_ip);
                cand = table[h];
                table[h] = (uint16_t)ip;
                if (cur == ld32(blk + cand)) break;
                ip = next_ip;
                cur = nxt;
            }
            if (out_of_input) break;
            op = lane_emit_literal(dst, op, blk + next_emit, ip - next_emit);   // :355
            bool done = false;
            uint32_t tail = 0;
            for (;;) {                                           // :370-398
                const uint32_t base = ip;
                uint32_t a = cand + 4;
                ip += 4;
                while (ip + 4 <= n && ld32(blk + ip) == ld32(blk + a)) {        // :176-193
                    ip += 4;
                    a += 4;
                }
                while (ip < n && blk[ip] == blk[a]) {
                    ++ip;
                    ++a;
                }
                op = lane_emit_copy(dst, op, base - cand, ip - base);
                next_emit = ip;
                if (ip >= limit) {
                    done = true;
                    break;
                }
                const uint64_t w = ld64(blk + ip - 1);
                const uint32_t here = (uint32_t)(w >> 8);
                tail = (uint32_t)(w >> 16);
                table[((uint32_t)w * kHashMul) >> shift] = (uint16_t)(ip - 1);
                const uint32_t hc = (here * kHashMul) >> shift;
                cand = table[hc];
                table[hc] = (uint16_t)ip;
                if (here != ld32(blk + cand)) break;
            }
            if (done) break;
            ++ip;
            cur = tail;
        }
    }
    if (next_emit < n) op = lane_emit_literal(dst, op, blk + next_emit, n - next_emit);   // :405-410
    st32(dst, op - 4);                                           // :412
    block_bytes[b] = op;
}

// ---------------------------------------------------------------------------
// K1, group form (ablation, SNAPPY_HIP_COMPRESS_VARIANT=5): FOUR blocks per wavefront.  Each 16-lane group owns
// one block; the parse is a flat state machine executed as predicated VALU code -- every loop iteration performs
// ONE probe (scan probe or post-copy probe, snappy_compress.c:336-348 / :391-398) for each of the wave's four
// groups, so one wave-instruction advances four parses and four independent table -> candidate chains are in
// flight per wave, none of it on the scalar unit.  Group-uniform state is replicated in the group's lanes; the
// lanes cooperate on table clears, literal payloads and multi-piece copies.  Hash tables: one u16[16384] per
// group in a global scratch.  Groups pull blocks from a shared atomic counter.
// The only wave collective is the loop condition; wave_barrier()s at the top level of the loop body separate
// the table store -> load -> store phases (free on hardware, where a wave runs in lockstep).
// Measured: 26.8 GB/s with 32768 groups in flight (5.4 us per iteration: three dependent HBM-random accesses),
// i.e. not faster than the wave-per-block form; kept as the starting point for the tag-filtered table idea.
// ---------------------------------------------------------------------------
constexpr uint32_t kGroupLanes = 16;

__device__ __forceinline__ uint32_t group_emit_literal(uint8_t* dst, uint32_t op, const uint8_t* src, uint32_t len,
                                                       uint32_t gl)
{
    const uint32_t n1 = len - 1;                                 // snappy_compress.c:202-225
    const uint32_t hdr = (n1 < 60) ? 1u : ((n1 < 256u) ? 2u : ((n1 < 65536u) ? 3u : 4u));
    if (gl < hdr) {
        const uint32_t tag = (n1 < 60) ? (n1 << 2) : ((58 + hdr) << 2);
        dst[op + gl] = (gl == 0) ? (uint8_t)tag : (uint8_t)(n1 >> (8 * (gl - 1)));
    }
    for (uint32_t i = gl; i < len; i += kGroupLanes) dst[op + hdr + i] = src[i];
    return op + hdr + len;
}

__device__ __forceinline__ uint32_t group_emit_copy(uint8_t* dst, uint32_t op, uint32_t off, uint32_t len, uint32_t gl)
{
    if (len > 64) {                                              // snappy_compress.c:254-272
        const uint32_t n64 = (len >= 68) ? ((len - 68) / 64 + 1) : 0;
        len -= 64 * n64;
        const uint32_t has60 = (len > 64) ? 1u : 0u;
        if (has60) len -= 60;
        const uint32_t nfull = n64 + has60;
        for (uint32_t k = gl; k < nfull; k += kGroupLanes) {
            const uint32_t plen = (k < n64) ? 64u : 60u;
            uint8_t* p = dst + op + 3 * k;
            p[0] = (uint8_t)(2 + ((plen - 1) << 2));
            p[1] = (uint8_t)off;
            p[2] = (uint8_t)(off >> 8);
        }
        op += 3 * nfull;
    }
    if (len < 12 && off < 2048) {                                // snappy_compress.c:234-245
        if (gl < 2) dst[op + gl] = (gl == 0) ? (uint8_t)(1 + ((len - 4) << 2) + ((off >> 8) << 5)) : (uint8_t)off;
        return op + 2;
    }
    if (gl < 3) dst[op + gl] = (gl == 0) ? (uint8_t)(2 + ((len - 1) << 2)) : ((gl == 1) ? (uint8_t)off : (uint8_t)(off >> 8));
    return op + 3;
}

// broadcast a value from the group's leader lane to the whole 16-lane group (LDS crossbar, no memory)
__device__ __forceinline__ uint32_t group_bcast(uint32_t v, uint32_t leader) { return (uint32_t)__shfl((int)v, (int)leader); }
__device__ __forceinline__ uint64_t group_bcast64(uint64_t v, uint32_t leader)
{
    return (uint64_t)group_bcast((uint32_t)v, leader) | ((uint64_t)group_bcast((uint32_t)(v >> 32), leader) << 32);
}

__global__ __launch_bounds__(64) void compress_blocks_group_kernel(const uint8_t* __restrict__ in, uint64_t in_len,
                                                                   uint32_t block_size, uint8_t* __restrict__ slots,
                                                                   uint32_t slot_stride, uint32_t* __restrict__ block_bytes,
                                                                   uint32_t num_blocks, uint32_t* tables,
                                                                   uint32_t* next_block)
{
    enum : uint32_t { kInit = 0, kScan = 1, kCopy = 2, kDone = 3 };
    const uint32_t lane = threadIdx.x;
    const uint32_t gl = lane & (kGroupLanes - 1);
    const uint32_t leader = lane & ~(kGroupLanes - 1);
    const bool lead = gl == 0;
    const uint32_t groups_per_wave = kWave / kGroupLanes;
    const uint32_t slot = blockIdx.x * groups_per_wave + (lane / kGroupLanes);
    const uint32_t total_slots = gridDim.x * groups_per_wave;
    uint32_t* table = tables + (size_t)slot * kMaxTableEntries;   // tagged entries: tag << 16 | position

    uint32_t mode = kInit;
    uint32_t n = 0, limit = 0, shift = 0, ip = 0, skip = 32, next_emit = 0, op = 4, cur_block = 0;
    const uint8_t* blk = in;
    uint8_t* dst = slots;
    // cursor cache: the 16 bytes at block offset cbase, so most probes need no cursor load
    uint32_t cbase = 0;
    uint64_t clo = 0, chi = 0;
    (void)total_slots;

    // Memory discipline: every (group-uniform) global load is issued by the group's leader lane only and
    // broadcast with group_bcast -- 4 active lanes per wave-instruction instead of 64 redundant ones.
    while (__ballot(mode != kDone)) {
        // ---------------- block start (get_hash_table, snappy_compress.c:139-146, :288-301) ----------------
        // Groups pull blocks from a shared counter (*next_block zeroed per launch), so a group that drew a
        // cheap block simply takes another one.
        bool fresh = false;
        const bool want = (mode == kInit);
        if (__ballot(want)) {
            uint32_t drawn = 0;
            if (want && lead) drawn = atomicAdd(next_block, 1u);
            drawn = group_bcast(drawn, leader);
            if (want) {
                if (drawn >= num_blocks) {
                    mode = kDone;
                } else {
                    cur_block = drawn;
                    const uint64_t start = (uint64_t)cur_block * block_size;
                    const uint64_t left = in_len - start;
                    n = (left < block_size) ? (uint32_t)left : block_size;
                    blk = in + start;
                    dst = slots + (uint64_t)cur_block * slot_stride;
                    op = 4;
                    next_emit = 0;
                    if (n < kInputMargin) {                      // whole block is one literal (:405-412)
                        op = group_emit_literal(dst, op, blk, n, gl);
                        if (lead) {
                            st32(dst, op - 4);
                            block_bytes[cur_block] = op;
                        }
                    } else {
                        const uint32_t ts = table_entries_for(n);
                        shift = (uint32_t)__builtin_clz(ts) + 1;
                        limit = n - kInputMargin;
                        ip = 1;
                        skip = 32;
                        cbase = 0;
                        mode = kScan;
                        fresh = true;
                    }
                }
            }
        }
        const bool probing = (mode == kScan) || (mode == kCopy);

        // ---------------- phase A0: (re)load the cursor cache when bytes ip-1 .. ip+6 are not inside it ----------------
        const bool reload = probing && (fresh || ip - 1 < cbase || ip + 7 > cbase + 16);
        if (__ballot(reload)) {
            if (reload) {
                if (!fresh) cbase = (ip - 1 + 16 <= n) ? ip - 1 : n - 16;
                if (lead) {
                    clo = ld64(blk + cbase);
                    chi = (cbase + 16 <= n) ? ld64(blk + cbase + 8) : 0;   // only a 15-byte block lacks the 16th byte
                }
            }
            clo = group_bcast64(clo, leader);
            chi = group_bcast64(chi, leader);
        }
        if (fresh) {
            // an empty slot means "candidate = position 0" (:346 on a zeroed table): store position 0's entry
            const uint32_t e_zero = (((uint32_t)clo * kHashMul) << (32 - shift)) & 0xffff0000u;
            const uint32_t ts = table_entries_for(n);
            uint4* t = reinterpret_cast<uint4*>(table);
            for (uint32_t i = gl; i < ts / 4; i += kGroupLanes) t[i] = make_uint4(e_zero, e_zero, e_zero, e_zero);
        }

        // ---------------- phase A: cursor bytes; post-copy insert of ip-1 (:391-392) ----------------
        uint32_t cur = 0, h = 0, mine = 0, next_ip = 0;
        bool exhausted = false;
        if (probing) {
            const uint32_t sh = 8 * (ip - 1 - cbase);            // 0..64 bits
            const uint64_t w = (sh == 0) ? clo : ((sh < 64) ? ((clo >> sh) | (chi << (64 - sh))) : chi);
            cur = (uint32_t)(w >> 8);
            const uint32_t prod = cur * kHashMul;
            h = prod >> shift;
            mine = ((prod << (32 - shift)) & 0xffff0000u) | ip;
            if (mode == kCopy) {
                const uint32_t pprod = (uint32_t)w * kHashMul;
                if (lead) table[pprod >> shift] = ((pprod << (32 - shift)) & 0xffff0000u) | (ip - 1);
            } else {
                next_ip = ip + (skip >> 5);                      // :339-343
                ++skip;
                exhausted = next_ip > limit;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase B: candidate lookup (:346, :395) ----------------
        const bool lookup = probing && !exhausted;
        uint32_t old = 0;
        if (lookup && lead) old = table[h];
        old = group_bcast(old, leader);
        // ---------------- phase C: table update, hit test ----------------
        if (lookup && lead) table[h] = mine;                     // :347, :397
        const uint32_t cand = old & 0xffffu;
        const bool tagmatch = lookup && (((old ^ mine) >> 16) == 0);
        bool hit = false;
        uint64_t c01 = 0;
        uint32_t c2 = 0;
        if (__ballot(tagmatch)) {                                // same tag: fetch candidate bytes (cand + 16 <= n)
            if (tagmatch && lead) {
                c01 = ld64(blk + cand);
                c2 = ld32(blk + cand + 8);
            }
            c01 = group_bcast64(c01, leader);
            c2 = group_bcast(c2, leader);
            hit = tagmatch && (cur == (uint32_t)c01);
        }
        if (lookup && !hit) {
            // miss: keep scanning (:348) or fall back from the copy chain to scanning (:398-401)
            if (mode == kCopy) {
                mode = kScan;
                skip = 32;
                ip += 1;
            } else {
                ip = next_ip;
            }
        }
        // ---------------- phase D: hit path -- literal, match length, copy (:355-389) ----------------
        if (__ballot(hit)) {
            uint64_t ahead = 0;
            if (hit && lead) ahead = ld64(blk + ip + 4);
            ahead = group_bcast64(ahead, leader);
            if (hit) {
                if (mode == kScan) op = group_emit_literal(dst, op, blk + next_emit, ip - next_emit, gl);   // :355
                // find_match_length (:176-193): 8 bytes at once, then 8-byte / 1-byte steps
                const uint64_t theirs = (c01 >> 32) | ((uint64_t)c2 << 32);
                const uint64_t diff = ahead ^ theirs;
                uint32_t matched;
                if (diff) {
                    matched = 4 + ((uint32_t)__builtin_ctzll(diff) >> 3);
                } else {
                    matched = 12;
                    while (ip + matched + 8 <= n) {
                        const uint64_t d = ld64(blk + ip + matched) ^ ld64(blk + cand + matched);
                        if (d) {
                            matched += (uint32_t)__builtin_ctzll(d) >> 3;
                            break;
                        }
                        matched += 8;
                    }
                    if (ip + matched + 8 > n)
                        while (ip + matched < n && blk[ip + matched] == blk[cand + matched]) ++matched;
                }
                op = group_emit_copy(dst, op, ip - cand, matched, gl);   // :380
                ip += matched;
                next_emit = ip;
                if (ip >= limit) exhausted = true;               // :388-389
                else mode = kCopy;
            }
        }
        if (probing && exhausted) {                              // emit_remainder (:405-412)
            if (next_emit < n) op = group_emit_literal(dst, op, blk + next_emit, n - next_emit, gl);
            if (lead) {
                st32(dst, op - 4);
                block_bytes[cur_block] = op;
            }
            mode = kInit;
        }
    }
}

// ---------------------------------------------------------------------------
// scan + gather: slots -> contiguous framed stream
// ---------------------------------------------------------------------------


}  // namespace snappy_hip
