// k1_windowed_form.hpp -- K1, windowed form (one wave-uniform probe at a time, optional look-ahead entry cache): round 1's first look-ahead kernel.
// Ablation code: compiled only with -DSNAPPY_ABLATION (tools/build_ablation.py -> libsnappy_hip_ablation.so); the product
// library contains ONE K1 pair (bulk parse: global-table + LDS-table kernels), the two-wavefront LDS form, and one K2.
// Every form here is bit-exact with the product (tests/test_gpu_ablation.py, tests/test_emulated_kernels.py).
#pragma once

namespace snappy_hip {

// 12 candidate bytes through the scalar cache: c0 = le32(cand) for the hit test, c1/c2 = the next 8 bytes for
// the match extension.  Aligned dwords + 64-bit shifts; needs cand + 16 <= block length (true for every
// candidate: cand < ip <= n - 15).  (Shifting all three eagerly measured 10 % faster than deferring c1/c2.)
struct CandidateBytes {
    uint32_t c0, c1, c2;
    __device__ __forceinline__ void fetch(const uint8_t* __restrict__ base16, uint64_t abs_pos)
    {
        const uint32_t* w = static_cast<const uint32_t*>(__builtin_assume_aligned(base16 + (abs_pos & ~3ull), 4));
        const uint32_t sh = 8 * (uint32_t)(abs_pos & 3);
        const uint64_t v01 = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        const uint64_t v12 = (uint64_t)w[1] | ((uint64_t)w[2] << 32);
        const uint64_t v23 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        c0 = (uint32_t)(v01 >> sh);
        c1 = (uint32_t)(v12 >> sh);
        c2 = (uint32_t)(v23 >> sh);
    }
    __device__ __forceinline__ uint64_t next8() const { return (uint64_t)c1 | ((uint64_t)c2 << 32); }
};


// Speculative table-entry cache of the windowed form ("look-ahead").  The hash of every position in the cursor window is
// already in a register, so the table slots the parse MAY probe next are known before it gets there.  When a probe lands on
// a window lane whose slot has not been read yet, lanes r .. r+kAhead-1 read their slots in ONE gather (`ent`), and the
// lanes whose tag matches read their 12 candidate bytes in a second (`k0..k2`, `kmask`).  Later probes inside the covered
// range take entry and candidate bytes from registers: a chain of short matches or a scan run costs two memory round trips
// per kAhead positions instead of two per probe.  The cached entries are kept equal to the table: every table write
// (probe inserts, :347/:397, and the post-match insert, :391-392) also overwrites `ent` in the lanes that hash to the
// written slot, and drops their cached candidate bytes (`kmask`), which belong to the previous occupant.
// A probe therefore sees exactly the entry the reference's sequential table would hold.
template <class Table, uint32_t kAhead>
struct EntryCache {
    uint32_t ent = 0;                 // per lane: table[h0] as of now, valid for lanes in [.., cov_end)
    uint32_t k0 = 0, k1 = 0, k2 = 0;  // per lane: 12 bytes at (ent & 0xffff), valid where kmask has the lane's bit
    unsigned long long kmask = 0;     // wave-uniform
    uint32_t cov_end = 0;             // wave-uniform: window lanes below this have a valid `ent` (lanes behind ip are dead)

    __device__ __forceinline__ void invalidate()
    {
        cov_end = 0;
        kmask = 0;
    }
    // read slots for lanes [r, r+span) of the window (clipped at 64), and candidate bytes where the tag allows a hit
    __device__ __forceinline__ void gather(const Table& table, const CursorWindow& win, uint32_t r, uint32_t span,
                                           uint32_t lane)
    {
        const bool g = lane >= r && lane < r + span;
        const uint32_t mine_l = win.e0 | (win.base + lane);
        if (g) ent = table.load_lane(win.h0, mine_l);
        const bool worth = g && !Table::certain_miss(ent, mine_l);
        if (worth) {                                  // every stored position p has p + 16 <= block length
            const uint8_t* __restrict__ c = win.blk + (ent & 0xffffu);
            k0 = ld32(c);
            k1 = ld32(c + 4);
            k2 = ld32(c + 8);
        }
        kmask = __ballot(worth);                      // lanes outside [r, r+span) are dead or not yet covered
        cov_end = (r + span < kWave) ? r + span : kWave;
    }
    // the table slot `h` now holds `entry`
    __device__ __forceinline__ void wrote(const CursorWindow& win, uint32_t h, uint32_t entry)
    {
        const bool same = win.h0 == h;
        ent = same ? entry : ent;
        kmask &= ~__ballot(same);
    }
};

// One probe (:344-348 / :393-398) at `ip`, which must be inside the window: returns the previous table entry, inserts
// `ip`, and reports whether the 4 bytes at the candidate equal `cur`; on a hit `cb` holds the candidate's 12 bytes.
template <class Table, uint32_t kAhead>
__device__ __forceinline__ bool probe_cached(const Table& table, EntryCache<Table, kAhead>& ec, const CursorWindow& win,
                                             const uint8_t* __restrict__ base16, uint64_t start, uint32_t ip, uint32_t cur,
                                             uint32_t span, uint32_t lane, uint32_t& cand, CandidateBytes& cb)
{
    const uint32_t r = ip - win.base;
    const uint32_t h = win.hash_at(ip);
    const uint32_t mine = win.entry_at(ip);
    if (r >= ec.cov_end) ec.gather(table, win, r, span, lane);
    const uint32_t old = (uint32_t)__builtin_amdgcn_readlane((int)ec.ent, (int)r);
    const bool cached_bytes = (ec.kmask >> r) & 1ull;     // before wrote() clears lane r's own bit
    table.put(h, mine, lane);
    ec.wrote(win, h, mine);
    cand = old & 0xffffu;
    if (Table::certain_miss(old, mine)) return false;
    if (cached_bytes) {
        cb.c0 = (uint32_t)__builtin_amdgcn_readlane((int)ec.k0, (int)r);
        if (cb.c0 != cur) return false;
        cb.c1 = (uint32_t)__builtin_amdgcn_readlane((int)ec.k1, (int)r);
        cb.c2 = (uint32_t)__builtin_amdgcn_readlane((int)ec.k2, (int)r);
        return true;
    }
    cb.fetch(base16, start + cand);                        // the slot was rewritten after the gather
    return cb.c0 == cur;
}

// kAhead > 0 enables the speculative entry cache above; 0 is the plain serial probe.
template <class Table, uint32_t kAhead = 0>
__device__ __forceinline__ void compress_one_block_windowed(const uint8_t* __restrict__ base16, uint64_t start,
                                                            uint64_t in_len, uint32_t n, uint8_t* __restrict__ dst,
                                                            const Table table_in, uint32_t lane,
                                                            uint32_t* __restrict__ block_bytes_out)
{
    const uint8_t* __restrict__ blk = base16 + start;
    // get_hash_table, snappy_compress.c:139-146 (+ shift, :288)
    const uint32_t ts = table_entries_for(n);
    const uint32_t shift = (uint32_t)__builtin_clz(ts) + 1;
    // "empty" = candidate position 0 (:346 on a zeroed table), carrying position 0's tag
    const uint32_t e_zero = (n >= kInputMargin) ? (((uld32(blk) * kHashMul) << (32 - shift)) & 0xffff0000u) : 0u;
    const Table table = table_in.with_empty(e_zero);
    if (n >= kInputMargin) table.init(ts, e_zero, lane);
    __builtin_amdgcn_wave_barrier();

    uint32_t op = 4;          // :291
    uint32_t next_emit = 0;   // :298

    if (n >= kInputMargin) {  // :301
        const uint32_t limit = n - kInputMargin;
        const uint64_t left = in_len - start;
        CursorWindow win;
        win.blk = blk;
        win.avail = (left < 0x7fffffffull) ? (uint32_t)left : 0x7fffffffu;
        win.shift = shift;
        win.reset(0, lane);
        EntryCache<Table, kAhead> ec;
        uint32_t ip = 1;      // :305
        for (;;) {
            // ---- step 1: scan for a 4-byte match (:333-348) ----
            uint32_t skip = 32;
            uint32_t cand = 0;
            CandidateBytes cb;
            bool hit = false;
            for (;;) {
                if (win.ensure(ip, lane) && kAhead) ec.invalidate();
                const uint32_t cur = win.bytes_at(ip);
                const uint32_t stride = skip++ >> 5;
                const uint32_t next_ip = ip + stride;
                if (next_ip > limit) break;     // :342-343, before touching the table
                if (kAhead) {
                    // look ahead only while the scan moves one position at a time (:339); wider strides probe one slot
                    hit = probe_cached<Table, kAhead>(table, ec, win, base16, start, ip, cur, stride == 1 ? kAhead : 1u, lane,
                                                      cand, cb);
                    if (hit) break;
                } else {
                    const uint32_t h = win.hash_at(ip);
                    const uint32_t mine = win.entry_at(ip);
                    const uint32_t old = table.exchange(h, mine, lane);
                    cand = old & 0xffffu;
                    if (!Table::certain_miss(old, mine)) {           // same tag: only now are the bytes worth fetching
                        cb.fetch(base16, start + cand);
                        if (cur == cb.c0) {
                            hit = true;
                            break;
                        }
                    }
                }
                ip = next_ip;
            }
            if (!hit) break;

            // ---- step 2: literal run [next_emit, ip) (:355); ip is inside the window ----
            op = emit_literal_windowed(dst, op, blk, next_emit, ip - next_emit, win.base, win.x0, lane);

            // ---- step 3: copy chain (:370-398) ----
            bool done = false;
            for (;;) {
                const uint32_t base = ip;
                // find_match_length (:176-193): first 8 bytes on the scalar side, the rest by 64 lanes
                const uint64_t mine = (uint64_t)win.bytes_near(ip + 4) | ((uint64_t)win.bytes_near(ip + 8) << 32);
                const uint64_t diff = mine ^ cb.next8();
                uint32_t matched;
                if (diff) {
                    matched = 4 + ((uint32_t)__builtin_ctzll(diff) >> 3);
                } else {
                    matched = 12 + match_extend(blk, cand + 12, ip + 12, n, lane);
                }
                ip += matched;
                op = emit_copy_packed(dst, op, base - cand, matched, lane);
                next_emit = ip;
                if (ip >= limit) {              // :388-389
                    done = true;
                    break;
                }
                if (win.ensure(ip - 1, lane) && kAhead) ec.invalidate();
                {
                    const uint32_t hp = win.hash_at(ip - 1);
                    const uint32_t ep = win.entry_at(ip - 1);
                    table.put(hp, ep, lane);                                  // :391-392
                    if (kAhead) ec.wrote(win, hp, ep);
                }
                if (win.ensure(ip, lane) && kAhead) ec.invalidate();
                const uint32_t here = win.bytes_at(ip);
                if (kAhead) {
                    if (!probe_cached<Table, kAhead>(table, ec, win, base16, start, ip, here, kAhead, lane, cand, cb)) break;
                } else {
                    const uint32_t mine_e = win.entry_at(ip);
                    const uint32_t old = table.exchange(win.hash_at(ip), mine_e, lane);   // :394-397
                    cand = old & 0xffffu;
                    if (Table::certain_miss(old, mine_e)) break;   // different tag: certain miss (:398)
                    cb.fetch(base16, start + cand);
                    if (here != cb.c0) break;                  // :396,:398
                }
            }
            if (done) break;
            ++ip;                                                // :400-401
        }
    }

    // emit_remainder (:405-410) and the size prefix (:412)
    if (next_emit < n) op = emit_literal(dst, op, blk + next_emit, n - next_emit, lane);
    if (lane == 0) {
        st32(dst, op - 4);
        *block_bytes_out = op;
    }
    __builtin_amdgcn_wave_barrier();
}


}  // namespace snappy_hip
