// k1_masked_form.hpp -- K1, masked form (probes of a window resolved with lane masks, serial exchange for shared slots): the step between the windowed and the bulk form.
// Ablation code: compiled only with -DSNAPPY_ABLATION (tools/build_ablation.py -> libsnappy_hip_ablation.so); the product
// library contains ONE K1 pair (bulk parse: global-table + LDS-table kernels), the two-wavefront LDS form, and one K2.
// Every form here is bit-exact with the product (tests/test_gpu_ablation.py, tests/test_emulated_kernels.py).
#pragma once

namespace snappy_hip {

// first 8 bytes of find_match_length (:176-193) on the scalar side, for a probe resolved by the serial exchange
__device__ __forceinline__ uint32_t ext_from_candidate(const CursorWindow& win, uint32_t ip, const CandidateBytes& cb)
{
    const uint64_t mine = (uint64_t)win.bytes_near(ip + 4) | ((uint64_t)win.bytes_near(ip + 8) << 32);
    const uint64_t diff = mine ^ cb.next8();
    return diff ? ((uint32_t)__builtin_ctzll(diff) >> 3) : 8u;
}

template <class Table, uint32_t kChunk>
__device__ __forceinline__ void compress_one_block_masked(const uint8_t* __restrict__ base16, uint64_t start, uint64_t in_len,
                                                          uint32_t n, uint8_t* __restrict__ dst, const Table table_in,
                                                          uint32_t lane, uint32_t* __restrict__ block_bytes_out,
                                                          lds_bytes_t dup_scratch)
{
    using State = MaskedWindowState<Table, kChunk>;
    const uint8_t* __restrict__ blk = base16 + start;
    const uint32_t ts = table_entries_for(n);                    // get_hash_table, :139-146 (+ shift, :288)
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
        State st;
        uint32_t ip = 1;      // :305
        for (;;) {
            // ---- step 1: scan for a 4-byte match (:333-348) ----
            uint32_t skip = 32;
            uint32_t cand = 0, ext = 0;
            bool hit = false;
            for (;;) {
                const uint32_t stride = skip >> 5;
                if (ip + stride > limit) break;                  // :342-343, before touching the table
                if (win.ensure(ip, lane)) st.invalidate();
                uint32_t r = ip - win.base;
                if (r >= uni(st.cov_end)) st.gather(table, win, dup_scratch, r, stride == 1 ? kChunk : 1u, lane);
                bool serial = true;
                if (stride == 1) {
                    // lanes [r, hi): resolved, probed one position apart, and allowed by :342 (position + 1 <= limit)
                    uint32_t hi = uni(st.cov_end);
                    const uint32_t budget = r + (64u - skip);    // the stride becomes 2 once skip reaches 64 (:339)
                    hi = budget < hi ? budget : hi;
                    const uint32_t lim = limit - win.base;
                    hi = lim < hi ? lim : hi;
                    const uint32_t run = hi - r;                 // >= 1
                    const uint32_t f0 = ctz64_or((st.hit | st.dup) >> r, 64u);
                    const uint32_t f = f0 < run ? f0 : run;      // plain misses in front of the first hit / DUP lane
                    if (f) {
                        State::commit(table, win, lane_range(r, f), lane);
                        ip += f;
                        skip += f;
                    }
                    if (f == run) continue;                      // coverage, stride-1 budget or limit ran out: re-evaluate
                    r += f;
                    serial = (st.dup >> r) & 1ull;
                    if (!serial) {                               // a resolved hit
                        State::commit(table, win, 1ull << r, lane);
                        cand = (uint32_t)__builtin_amdgcn_readlane((int)st.ent, (int)r) & 0xffffu;
                        ext = (uint32_t)__builtin_amdgcn_readlane((int)st.extv, (int)r);
                        hit = true;
                        break;
                    }
                } else {
                    serial = (st.dup >> r) & 1ull;
                    if (!serial) {
                        State::commit(table, win, 1ull << r, lane);
                        if ((st.hit >> r) & 1ull) {
                            cand = (uint32_t)__builtin_amdgcn_readlane((int)st.ent, (int)r) & 0xffffu;
                            ext = (uint32_t)__builtin_amdgcn_readlane((int)st.extv, (int)r);
                            hit = true;
                            break;
                        }
                        ip += stride;
                        ++skip;
                        continue;
                    }
                }
                // serial probe: this lane shares its table slot with another lane of the window, ask the table itself
                {
                    const uint32_t cur = win.bytes_at(ip);
                    const uint32_t mine = win.entry_at(ip);
                    const uint32_t old = table.exchange(win.hash_at(ip), mine, lane);
                    cand = old & 0xffffu;
                    if (!Table::certain_miss(old, mine)) {
                        CandidateBytes cb;
                        cb.fetch(base16, start + cand);
                        if (cur == cb.c0) {
                            ext = ext_from_candidate(win, ip, cb);
                            hit = true;
                            break;
                        }
                    }
                    ip += skip >> 5;
                    ++skip;
                }
            }
            if (!hit) break;

            // ---- step 2: literal run [next_emit, ip) (:355); ip is inside the window ----
            op = emit_literal_windowed(dst, op, blk, next_emit, ip - next_emit, win.base, win.x0, lane);

            // ---- step 3: copy chain (:370-398) ----
            bool done = false;
            for (;;) {
                const uint32_t mbase = ip;
                uint32_t matched = 4 + ext;                      // find_match_length (:176-193)
                if (ext == 8) matched = 12 + match_extend(blk, cand + 12, ip + 12, n, lane);
                ip += matched;
                op = emit_copy_packed(dst, op, mbase - cand, matched, lane);
                next_emit = ip;
                if (ip >= limit) {                               // :388-389
                    done = true;
                    break;
                }
                if (win.ensure(ip - 1, lane)) st.invalidate();
                State::commit(table, win, 1ull << (ip - 1 - win.base), lane);   // :391-392
                if (win.ensure(ip, lane)) st.invalidate();
                const uint32_t r = ip - win.base;
                if (r >= uni(st.cov_end)) st.gather(table, win, dup_scratch, r, kChunk, lane);
                if (!((st.dup >> r) & 1ull)) {                   // :393-398 from the cache
                    State::commit(table, win, 1ull << r, lane);
                    if (!((st.hit >> r) & 1ull)) break;
                    cand = (uint32_t)__builtin_amdgcn_readlane((int)st.ent, (int)r) & 0xffffu;
                    ext = (uint32_t)__builtin_amdgcn_readlane((int)st.extv, (int)r);
                } else {
                    const uint32_t here = win.bytes_at(ip);
                    const uint32_t mine_e = win.entry_at(ip);
                    const uint32_t old = table.exchange(win.hash_at(ip), mine_e, lane);
                    cand = old & 0xffffu;
                    if (Table::certain_miss(old, mine_e)) break;
                    CandidateBytes cb;
                    cb.fetch(base16, start + cand);
                    if (here != cb.c0) break;
                    ext = ext_from_candidate(win, ip, cb);
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
