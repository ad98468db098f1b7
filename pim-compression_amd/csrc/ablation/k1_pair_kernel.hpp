// snappy_k1_pair.hpp -- K1, workgroup-per-block form: TWO wavefronts share one Snappy block and ONE u16 hash table in
// LDS (the reference's own table, snappy_compress.c:139-146; the table lives in the local memory of the unit that parses
// the block, as dpu-compress/dpu_compress.c:472-476 keeps it in WRAM).
//
// Why two wavefronts: the greedy parse (snappy_compress.c:284-413) is a dependent chain, and a wavefront alone on its SIMD
// issues such a chain at ~9 cycles per instruction.  LDS holds four to five 32 KiB tables per CU, so an LDS-table kernel
// has four or five chains per CU whatever it does; the only way to more throughput per table is to take work OFF the
// chain.  The block is cut into its 64-byte windows; window k belongs to wavefront k mod 2.  Per window the owner does
//   PREP      parse-independent: window bytes, hashes, ONE speculative gather of all 64 table slots and candidate bytes,
//             the would-hit mask and match lengths, the duplicate-slot analysis               (compress_one_block_bulk's
//             gather, unchanged)
//   CRITICAL  holds the block's TOKEN (cursor, skip counter, output offset): walks the window's matches with lane masks,
//             settles the lanes whose speculation is in doubt, commits the window's table inserts, passes the token on
//   POST      emits the window's elements (literal bytes, headers, copies) at the offsets fixed in CRITICAL
// and only CRITICAL is serial between the two wavefronts: while one holds the token the other emits its last window and
// prepares its next one.
//
// Exactness.  The gather of window k runs before window k-1 has committed its inserts (that is the overlap), so a slot
// it read may be stale -- but only through an insert of a position of window k-1 (every insert of a position of window w
// is made inside CRITICAL(w), and CRITICAL(k-2), this wavefront's own, has finished before the gather starts).  A lane
// of window k can be affected only if its hash equals that of a lane of window k-1: those lanes (`sus`, found like the
// in-window duplicate slots, a superset) never trust the gather; when the walk reaches one, its candidate is taken from
// registers: the latest INSERTED lane with the same hash in this window, else in the previous window (whose final insert
// mask travels with the token), else the gathered entry, which is then provably current.  Decisions, table contents at
// every read and output bytes are those of snappy_compress.c:284-413; tests compare with the oracle bit for bit.
#pragma once

namespace snappy_hip {

#ifdef SNAPPY_PAIR_PROBE      // diagnostic build only (tools/prof_pair.py): cycles per phase, summed over all wavefronts
__device__ unsigned long long g_pair_prof[16];
#define PAIR_T() ((unsigned long long)__builtin_readcyclecounter())
#define PAIR_ADD(i, v) (pp[i] += (v))
#else
#define PAIR_ADD(i, v) ((void)0)
#endif

constexpr uint32_t kPairWaves = 2;
constexpr uint32_t kPairScratchPerWave = 2048;      // 1 KiB duplicate-slot race tables + 1 KiB previous-window marks
constexpr uint32_t kPairTokenBytes = 64;

struct PairToken {                // two 16-byte rows, each written / read with one LDS instruction
    uint32_t seq;                 // row 0: window the token is for; this row is written last and polled by that window's owner
    uint32_t ip, skip, next_emit;
    uint32_t op;                  // row 1
    uint32_t ins_lo, ins_hi;      // lanes of window seq-1 whose positions were inserted
    uint32_t flags;               // 1 = block finished, 2 = position ip-1 still has to be inserted (snappy_compress.c:391-392)
};

// Ordering between the two wavefronts of a workgroup goes through LDS only, and LDS executes one wavefront's operations in
// issue order: all that is needed is that the COMPILER keeps the order written here.  (A workgroup-scope fence would also
// wait for the wavefront's outstanding global stores -- the elements it has just emitted -- on every hand-over.)
__device__ __forceinline__ void pair_lds_order()
{
#ifndef SNAPPY_EMU
    asm volatile("" ::: "memory");
#endif
}
// one token row = one 16-byte LDS access (ds_read_b128 / ds_write_b128)
#ifdef SNAPPY_EMU
typedef volatile uint32_t* pair_rows_t;
__device__ __forceinline__ void pair_row_load(pair_rows_t t, uint32_t row, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    a = t[4 * row];
    b = t[4 * row + 1];
    c = t[4 * row + 2];
    d = t[4 * row + 3];
}
__device__ __forceinline__ void pair_row_store(pair_rows_t t, uint32_t row, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    t[4 * row + 1] = b;
    t[4 * row + 2] = c;
    t[4 * row + 3] = d;
    t[4 * row] = a;
}
#else
typedef uint32_t pair_row_vec __attribute__((ext_vector_type(4)));
typedef volatile __attribute__((address_space(3))) pair_row_vec* pair_rows_t;
__device__ __forceinline__ void pair_row_load(pair_rows_t t, uint32_t row, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    const pair_row_vec v = t[row];
    a = v.x;
    b = v.y;
    c = v.z;
    d = v.w;
}
__device__ __forceinline__ void pair_row_store(pair_rows_t t, uint32_t row, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    pair_row_vec v;
    v.x = a;
    v.y = b;
    v.z = c;
    v.w = d;
    t[row] = v;
}
#endif
constexpr uint32_t kPairDone = 1u, kPairPendingInsert = 2u;

// dynamic LDS of one workgroup: [u16 table of table_entries_for(block_size)] [scratch of wave 0] [scratch of wave 1] [token]
__host__ __device__ inline uint32_t pair_table_entries(uint32_t block_size) { return lds_table_entries(block_size); }
__host__ __device__ inline uint32_t pair_lds_bytes(uint32_t block_size)
{
    return 2u * pair_table_entries(block_size) + kPairWaves * kPairScratchPerWave + kPairTokenBytes;
}

// lanes of the current window (hashes h) that share a table slot with a lane of the previous window (hashes hp): a
// superset, by the same two-overlapping-halves test as dup_slot_lanes.  `gen` (1..255) stamps this call's marks, so the
// region needs no clearing between windows (a mark of 255 windows ago can only add a lane to the superset).
__device__ __forceinline__ unsigned long long prev_window_sharers(lds_bytes_t marks, uint32_t hp, uint32_t h, uint32_t gen)
{
    marks[hp & (kDupSlots / 2 - 1)] = (uint8_t)gen;
    marks[kDupSlots / 2 + ((hp >> 5) & (kDupSlots / 2 - 1))] = (uint8_t)gen;
    __builtin_amdgcn_wave_barrier();
    const uint32_t a = marks[h & (kDupSlots / 2 - 1)];
    const uint32_t b = marks[kDupSlots / 2 + ((h >> 5) & (kDupSlots / 2 - 1))];
    __builtin_amdgcn_wave_barrier();
    return __ballot(a == gen && b == gen);
}

// le32 at (window start + lane + d), d in {4, 8}, for a window whose dwords are in `cur` and whose successor's are in `nxt`
__device__ __forceinline__ uint32_t bytes_ahead2(uint32_t cur, uint32_t nxt, uint32_t lane, uint32_t d)
{
    const uint32_t src = (lane + d) & 63u;
    const uint32_t a = (uint32_t)__shfl((int)cur, (int)src);
    const uint32_t b = (uint32_t)__shfl((int)nxt, (int)src);
    return (lane + d < kWave) ? a : b;
}

__device__ __forceinline__ uint32_t literal_header_bytes(uint32_t len)      // snappy_compress.c:202-225
{
    const uint32_t n1 = len - 1;
    return n1 < 60u ? 1u : (n1 < 256u ? 2u : (n1 < 65536u ? 3u : 4u));
}

// What POST needs to emit one segment of a window (the masks of compress_one_block_bulk's emission step).
struct PairPendingEmit {
    unsigned long long H, COV;
    uint32_t why, r, r_end, next_emit, op;
    bool valid;
};

// compress_one_block_bulk's emission of one segment (snappy_compress.c:355, :202-245), from the stashed masks
template <class State>
__device__ __forceinline__ void pair_emit_segment(const PairPendingEmit& e, const CursorWindow& win, const State& st,
                                                  const uint8_t* __restrict__ blk, uint8_t* __restrict__ dst, uint32_t lane)
{
    const unsigned long long H = e.H, COV = e.COV;
    uint32_t op = e.op;
    const uint32_t first_hit = (uint32_t)__builtin_ctzll(H);
    const uint32_t last_end = (e.why == 0) ? e.r_end : 64u - (uint32_t)__builtin_clzll(COV);
    uint32_t s0 = first_hit;
    const uint32_t p0 = win.base + first_hit;
    if (e.next_emit >= win.base && p0 - e.next_emit <= 60u) {
        s0 = e.next_emit - win.base;               // the first run is inside the window too
    } else if (p0 > e.next_emit) {                 // it started in an earlier window (or is 61+ bytes)
        op = emit_literal_windowed(dst, op, blk, e.next_emit, p0 - e.next_emit, win.base, win.x0, lane);
    }
    const unsigned long long LIT = ((~0ull << s0) & lanes_below(last_end)) & ~COV;
    const unsigned long long LS = LIT & ~(LIT << 1);            // first lane of each literal run
    const uint32_t off = win.base + lane - (st.ent & 0xffffu);  // meaningful in H lanes
    const uint32_t len = 4u + st.extv;
    const bool is_hit = __builtin_amdgcn_inverse_ballot_w64(H);
    const bool three = off >= 2048u || len >= 12u;                 // :234-245
    const unsigned long long H3 = __ballot(is_hit && three);
    uint32_t P = mbcnt64(H, 0);
    P = mbcnt64(LIT, op + 2u * P);
    P = mbcnt64(H3, P);
    P = mbcnt64(LS, P);
    const bool is_ls = __builtin_amdgcn_inverse_ballot_w64(LS);
    if (__builtin_amdgcn_inverse_ballot_w64(LIT)) dst[P + (is_ls ? 1u : 0u)] = (uint8_t)win.x0;
    if (is_ls) {
        const uint32_t runlen = (uint32_t)__builtin_ctzll(~LIT >> lane);   // a copy follows every run
        dst[P] = (uint8_t)((runlen - 1) << 2);                          // :202-207, runs here are <= 60
    }
    if (is_hit) {
        uint32_t b0;
        if (!three) b0 = 1u + ((len - 4u) << 2) + ((off >> 8) << 5);        // :234-239
        else b0 = 2u + ((len - 1u) << 2);                                   // :240-245
        dst[P] = (uint8_t)b0;
        dst[P + 1] = (uint8_t)off;
        if (three) dst[P + 2] = (uint8_t)(off >> 8);
    }
}

// One block, two wavefronts.  `table`, `tok` are the workgroup's; `dup_scratch`, `prev_marks` this wavefront's own.
// Both wavefronts call this with the same arguments (except wave / scratch) and leave it together.
__device__ __forceinline__ void compress_one_block_pair(const uint8_t* __restrict__ base16, uint64_t start, uint64_t in_len,
                                                        uint32_t n, uint8_t* __restrict__ dst, uint16_t* table, uint32_t wave,
                                                        uint32_t lane, uint32_t* __restrict__ block_bytes_out,
                                                        lds_bytes_t dup_scratch, lds_bytes_t prev_marks,
                                                        volatile PairToken* tok)
{
    pair_rows_t tok_rows = (pair_rows_t)tok;
    using Table = LdsTable;
    using State = MaskedWindowState<Table, 64>;
    const uint8_t* __restrict__ blk = base16 + start;
    const uint32_t ts = table_entries_for(n);                    // get_hash_table, :139-146 (+ shift, :288)
    const uint32_t shift = (uint32_t)__builtin_clz(ts) + 1;
    const Table tbl{table};
    const uint32_t tid = wave * kWave + lane;

    // ---- block start: zeroed table (:145), cleared marks, the token at the reference's initial state (:291-305) ----
    if (n >= kInputMargin) {
        uint4* q = reinterpret_cast<uint4*>(table);
        for (uint32_t i = tid; i < ts / 8; i += kPairWaves * kWave) q[i] = make_uint4(0, 0, 0, 0);
        for (uint32_t i = lane; i < kDupSlots; i += kWave) prev_marks[i] = 0;
    }
    if (tid == 0) {
        pair_row_store(tok_rows, 1, 4, 0, 0, 0);                 // op = 4 (:291), nothing inserted, no flags
        pair_row_store(tok_rows, 0, 0, 1, 32, 0);                // window 0, ip = 1 (:305), skip = 32 (:333), next_emit = 0 (:298)
    }
    __syncthreads();

    if (n < kInputMargin) {                                      // :301: the whole block is one literal (:405-412)
        if (wave == 0) {
            const uint32_t op = emit_literal(dst, 4, blk, n, lane);
            if (lane == 0) {
                st32(dst, op - 4);
                *block_bytes_out = op;
            }
        }
        return;
    }

    const uint32_t limit = n - kInputMargin;
    const uint64_t left = in_len - start;
    CursorWindow win;
    win.blk = blk;
    win.avail = (left < 0x7fffffffull) ? (uint32_t)left : 0x7fffffffu;
    win.shift = shift;
    State st;

#ifdef SNAPPY_PAIR_PROBE
    unsigned long long pp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PAIR_SUB(i, t0) (pp[i] += PAIR_T() - (t0))
#define PAIR_NOW() PAIR_T()
    unsigned long long pt = PAIR_T();
#define PAIR_LAP(i) do { const unsigned long long now_ = PAIR_T(); pp[i] += now_ - pt; pt = now_; } while (0)
#else
#define PAIR_LAP(i) ((void)0)
#define PAIR_SUB(i, t0) ((void)0)
#define PAIR_NOW() 0ull
#endif
    uint32_t k = wave;                                           // this wavefront's window
    // dwords of windows k-1, k, k+1 (the neighbours serve the bytes-ahead and the previous-window tests)
    uint32_t xprev = k ? win.load_at(64u * (k - 1u), lane) : 0u;
    uint32_t xcur = win.load_at(64u * k, lane);
    uint32_t xnext = win.load_at(64u * k + 64u, lane);

    for (;;) {
        // ================= PREP(k): everything that does not depend on the parse =================
        unsigned long long sus = 0;                              // lanes that share a slot with a lane of window k-1
        uint32_t hp = 0, xap = 0, xbp = 0;
        const bool have_window = 64u * k < n;
        st.invalidate();
        win.base = 64u * k;
        if (have_window) {
            win.x0 = xcur;
            win.x1 = xnext;
            win.rehash();
            st.template gather<true, true>(tbl, win, dup_scratch, 0, 64, lane, n);
            if (k) {
                hp = (xprev * kHashMul) >> shift;
                xap = bytes_ahead2(xprev, xcur, lane, 4);
                xbp = bytes_ahead2(xprev, xcur, lane, 8);
                sus = prev_window_sharers(prev_marks, hp, win.h0, (k % 255u) + 1u);
            }
        }

        PAIR_LAP(0);                                             // prep
        // ================= wait for the token of window k =================
        // The token is two 16-byte rows; every lane reads the same address (an LDS broadcast), so a poll is one
        // ds_read_b128 and the row that carries `seq` arrives together with ip / skip / next_emit.
        uint32_t ip, skip, next_emit, op, flags, ins_lo, ins_hi;
        for (;;) {
            uint32_t seq;
            pair_row_load(tok_rows, 0, seq, ip, skip, next_emit);
            if (uni(seq) == k) break;
        }
        pair_lds_order();                                        // the second row is read after the first was seen complete
        pair_row_load(tok_rows, 1, op, ins_lo, ins_hi, flags);
        ip = uni(ip);
        skip = uni(skip);
        next_emit = uni(next_emit);
        op = uni(op);
        flags = uni(flags);
        const unsigned long long insp = (unsigned long long)uni(ins_lo) | ((unsigned long long)uni(ins_hi) << 32);
        PAIR_LAP(1);                                             // wait
        if (flags & kPairDone) break;

        // ================= CRITICAL(k) =================
        const uint32_t wbase = 64u * k;
        bool finished = false;
        PairPendingEmit pe;
        pe.valid = false;
        // position ip-1 behind a copy that ended beyond its window (:391-392): inserted by the window that holds it
        if ((flags & kPairPendingInsert) && ip - 1u >= wbase && ip - 1u < wbase + 64u) {
            State::commit(tbl, win, 1ull << (ip - 1u - wbase), lane);
            st.inserted |= 1ull << (ip - 1u - wbase);
            flags &= ~kPairPendingInsert;
        }
        for (;;) {
            const uint32_t stride = skip >> 5;
            const uint32_t step = stride ? stride : 1u;
            if (ip + step > limit) {                             // :342-343 / :388-389
                finished = true;
                break;
            }
            if (ip >= wbase + 64u) break;                        // the cursor left this window
            uint32_t r = ip - wbase;
            unsigned long long stopm = st.dup | sus | st.longm;

            bool need_single = true;
            if (stride <= 1 || !((stopm >> r) & 1ull)) {
                // ---------------- segment (compress_one_block_bulk) ----------------
                uint32_t hi = kWave;
                const uint32_t lim = limit - wbase;              // lanes below may be probed (position + 1 <= limit)
                hi = lim < hi ? lim : hi;
                if (hi < kWave) stopm |= ~0ull << hi;
                unsigned long long inter = st.hit | stopm;
                unsigned long long pre = 0;                      // lanes probed by the strided prefix
                uint32_t B = 64u - skip;                         // stride-1 probes left before :339 widens the stride
                if (stride > 1) {
                    const uint32_t x = lane - r;
                    bool mine = lane == r;
                    if (stride < kWave) {
                        const uint32_t q = (x * kRecip16[stride]) >> 16;     // x / stride for x < 64
                        mine = lane >= r && x == q * stride;
                    }
                    const uint32_t nrem = 32u - (skip & 31u);
                    uint32_t hs = r + nrem * stride;
                    hs = hs < hi ? hs : hi;
                    const uint32_t lims = lim - (stride - 1u);               // position + stride <= limit
                    hs = hs < lims ? hs : lims;                              // > r by the check at the top of the loop
                    const unsigned long long smask = __ballot(mine) & lanes_below(hs);
                    const unsigned long long m = inter & smask;
                    const uint32_t p = ctz64_or(m, kWave);
                    pre = p < kWave ? (p ? smask & lanes_below(p) : 0ull) : smask;
                    const uint32_t cnt = (uint32_t)__builtin_popcountll(pre);
                    skip += cnt;
                    if (p == kWave || ((stopm >> p) & 1ull)) {               // no hit at this stride level here
                        State::commit(tbl, win, pre, lane);                  // (no shared-slot lane among them: those are stops)
                        st.inserted |= pre;
                        ip += cnt * stride;
                        continue;
                    }
                    r = p;
                    B = 1;                                                   // the walk takes the hit at p right away
                }
                const uint32_t r0 = r;
                unsigned long long H = 0, COV = 0;
                uint32_t why;
                need_single = false;
                [[maybe_unused]] const unsigned long long tw0 = PAIR_NOW();
                PAIR_ADD(7, 1);
                for (;;) {
                    why = segment_walk(inter, stopm, 4u + st.extv, hi, r, B, H, COV);
                    if (why != 2 || r >= hi) break;
                    PAIR_ADD(8, 1);
                    [[maybe_unused]] const unsigned long long tr0 = PAIR_NOW();
                    // The walk stands on a lane whose speculation is in doubt, or on a hit of 28+ bytes: settle it here
                    bool hit_r;
                    uint32_t cand_r, ext_r;
                    uint32_t sat_r = 8;                          // where ext_r saturates: 8, or 24 for a deep lane's own result
                    if (((st.dup | sus) >> r) & 1ull) {
                        const uint32_t hr = (uint32_t)__builtin_amdgcn_readlane((int)win.h0, (int)r);
                        const unsigned long long inner = COV & ~H;
                        const unsigned long long so_far = st.inserted | pre | (((~0ull << r0) & ((1ull << r) - 1ull)) & ~inner) |
                                                          (COV & ~(inner >> 1));
                        const unsigned long long J = __ballot(win.h0 == hr) & so_far & ((1ull << r) - 1ull);
                        const unsigned long long Jp = J ? 0ull : (__ballot(hp == hr) & insp);
                        if (J) {                                 // the slot holds the latest inserted lane with this hash
                            const uint32_t j = 63u - (uint32_t)__builtin_clzll(J);
                            cand_r = wbase + j;
                            hit_r = (uint32_t)__builtin_amdgcn_readlane((int)win.x0, (int)r) ==
                                    (uint32_t)__builtin_amdgcn_readlane((int)win.x0, (int)j);
                            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)r) ^
                                                (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)j);
                            const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)r) ^
                                                (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)j);
                            ext_r = d0 ? ((uint32_t)__builtin_ctz(d0) >> 3) : (d1 ? 4u + ((uint32_t)__builtin_ctz(d1) >> 3) : 8u);
                        } else if (Jp) {                         // ... or, failing that, of the previous window
                            const uint32_t j = 63u - (uint32_t)__builtin_clzll(Jp);
                            cand_r = wbase - 64u + j;
                            hit_r = (uint32_t)__builtin_amdgcn_readlane((int)win.x0, (int)r) ==
                                    (uint32_t)__builtin_amdgcn_readlane((int)xprev, (int)j);
                            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)r) ^
                                                (uint32_t)__builtin_amdgcn_readlane((int)xap, (int)j);
                            const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)r) ^
                                                (uint32_t)__builtin_amdgcn_readlane((int)xbp, (int)j);
                            ext_r = d0 ? ((uint32_t)__builtin_ctz(d0) >> 3) : (d1 ? 4u + ((uint32_t)__builtin_ctz(d1) >> 3) : 8u);
                        } else {                                 // nothing inserted on this slot since the gather: it stands
                            hit_r = (st.hit >> r) & 1ull;
                            cand_r = (uint32_t)__builtin_amdgcn_readlane((int)st.ent, (int)r) & 0xffffu;
                            ext_r = (uint32_t)__builtin_amdgcn_readlane((int)st.extv, (int)r);
                            sat_r = ((st.deepm >> r) & 1ull) ? 24u : 8u;
                        }
                    } else {                                     // a resolved hit whose compared bytes all match
                        hit_r = true;
                        cand_r = (uint32_t)__builtin_amdgcn_readlane((int)st.ent, (int)r) & 0xffffu;
                        sat_r = ((st.deepm >> r) & 1ull) ? 24u : 8u;
                        ext_r = sat_r;
                    }
                    uint32_t len_r = 4u + ext_r;
                    if (hit_r && ext_r == sat_r) {
                        PAIR_ADD(10, 1);
                        len_r = 4u + sat_r + match_extend(blk, cand_r + 4u + sat_r, wbase + r + 4u + sat_r, n, lane);
                    }
                    PAIR_SUB(9, tr0);
                    if (hit_r && len_r > 63u) {                  // more than one copy element (:254-272): single step below
                        need_single = true;
                        break;
                    }
                    if (lane == r) {
                        st.extv = len_r - 4u;
                        st.ent = cand_r;
                    }
                    stopm &= ~(1ull << r);
                    inter = hit_r ? (inter | (1ull << r)) : (inter & ~(1ull << r));
                }
                PAIR_SUB(6, tw0);                                // walk incl. resolves
                [[maybe_unused]] const unsigned long long tc0 = PAIR_NOW();
                if (why == 1) {                                  // B (or all remaining) lanes of misses
                    const uint32_t room = hi - r;
                    const uint32_t adv = B < room ? B : room;
                    r += adv;
                    B -= adv;
                }
                ip = wbase + r;
                skip = 64u - B;
                const bool done = (why == 0) && ip >= limit;     // :388-389 behind the last copy
                const uint32_t r_end = r < kWave ? r : kWave;

                // ---- table: every probed lane (:346-347, :397) and every "ip - 1" lane (:391-392) inserts its position ----
                const unsigned long long interior = COV & ~H;
                const unsigned long long walked = r_end > r0 ? ((~0ull << r0) & lanes_below(r_end)) : 0ull;   // may be empty
                unsigned long long C = pre | (walked & ~interior);                                  // probed lanes
                unsigned long long endl = COV & ~(interior >> 1);                              // last lane of each copy
                if (why == 0 && (r > kWave || done)) endl &= ~(1ull << (r_end - 1));           // the last copy's is not (yet) due
                C |= endl;
                State::commit(tbl, win, C & ~st.dup, lane);
                st.inserted |= C;
                for (unsigned long long d = C & st.dup; d; d &= d - 1)                         // shared slots: in position order
                    State::commit(tbl, win, d & (~d + 1), lane);

                if (H) {
                    // ---- emission is POST's; here only the output offset moves on (same arithmetic, no stores) ----
                    if (pe.valid) pair_emit_segment(pe, win, st, blk, dst, lane);   // a second segment in this window: rare
                    pe.H = H;
                    pe.COV = COV;
                    pe.why = why;
                    pe.r = r;
                    pe.r_end = r_end;
                    pe.next_emit = next_emit;
                    pe.op = op;
                    pe.valid = true;
                    const uint32_t first_hit = (uint32_t)__builtin_ctzll(H);
                    const uint32_t last_end = (why == 0) ? r_end : 64u - (uint32_t)__builtin_clzll(COV);
                    uint32_t s0 = first_hit;
                    const uint32_t p0 = wbase + first_hit;
                    if (next_emit >= wbase && p0 - next_emit <= 60u) {
                        s0 = next_emit - wbase;
                    } else if (p0 > next_emit) {
                        const uint32_t l0 = p0 - next_emit;
                        op += l0 + literal_header_bytes(l0);
                    }
                    const unsigned long long LIT = ((~0ull << s0) & lanes_below(last_end)) & ~COV;
                    const unsigned long long LS = LIT & ~(LIT << 1);
                    const uint32_t off = wbase + lane - (st.ent & 0xffffu);
                    const uint32_t len = 4u + st.extv;
                    const bool is_hit = __builtin_amdgcn_inverse_ballot_w64(H);
                    const unsigned long long H3 = __ballot(is_hit && (off >= 2048u || len >= 12u));
                    op += (uint32_t)__builtin_popcountll(LIT) + 2u * (uint32_t)__builtin_popcountll(H) +
                          (uint32_t)__builtin_popcountll(H3) + (uint32_t)__builtin_popcountll(LS);
                    next_emit = wbase + ((why == 0) ? r : last_end);
                }
                PAIR_SUB(11, tc0);                               // commit + offset accounting
                if (done) {
                    finished = true;
                    break;
                }
                if (why == 0 && r > kWave) flags |= kPairPendingInsert;   // ip - 1 lies in a later window
                if (!need_single) continue;
                r = ip - wbase;                                  // a copy of 64+ bytes starts here
            }

            // ---------------- single step: shared-slot lane, long match, or stride > 1 ----------------
            PAIR_ADD(12, 1);
            uint32_t cand = 0, ext = 0, sat = 8;
            bool hit;
            unsigned long long J = 0, Jp = 0;
            if (((st.dup | sus) >> r) & 1ull) {
                const uint32_t hr = win.hash_at(ip);
                J = __ballot(win.h0 == hr) & st.inserted & ((1ull << r) - 1ull);
                if (!J) Jp = __ballot(hp == hr) & insp;
            }
            State::commit(tbl, win, 1ull << r, lane);
            st.inserted |= 1ull << r;
            if (J) {
                const uint32_t j = 63u - (uint32_t)__builtin_clzll(J);
                cand = wbase + j;
                hit = win.bytes_at(ip) == (uint32_t)__builtin_amdgcn_readlane((int)win.x0, (int)j);
                if (hit) {
                    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)r) ^
                                        (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)j);
                    const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)r) ^
                                        (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)j);
                    ext = d0 ? ((uint32_t)__builtin_ctz(d0) >> 3) : (d1 ? 4u + ((uint32_t)__builtin_ctz(d1) >> 3) : 8u);
                }
            } else if (Jp) {
                const uint32_t j = 63u - (uint32_t)__builtin_clzll(Jp);
                cand = wbase - 64u + j;
                hit = win.bytes_at(ip) == (uint32_t)__builtin_amdgcn_readlane((int)xprev, (int)j);
                if (hit) {
                    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)r) ^
                                        (uint32_t)__builtin_amdgcn_readlane((int)xap, (int)j);
                    const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)r) ^
                                        (uint32_t)__builtin_amdgcn_readlane((int)xbp, (int)j);
                    ext = d0 ? ((uint32_t)__builtin_ctz(d0) >> 3) : (d1 ? 4u + ((uint32_t)__builtin_ctz(d1) >> 3) : 8u);
                }
            } else {
                hit = (st.hit >> r) & 1ull;
                if (hit) {
                    cand = (uint32_t)__builtin_amdgcn_readlane((int)st.ent, (int)r) & 0xffffu;
                    ext = (uint32_t)__builtin_amdgcn_readlane((int)st.extv, (int)r);
                    sat = ((st.deepm >> r) & 1ull) ? 24u : 8u;
                }
            }
            if (!hit) {
                ip += step;
                ++skip;
                continue;
            }
            if (ip > next_emit) op = emit_literal_windowed(dst, op, blk, next_emit, ip - next_emit, wbase, win.x0, lane);   // :355
            const uint32_t mbase = ip;
            uint32_t matched = 4 + ext;                          // find_match_length (:176-193)
            if (ext == sat) matched = 4 + sat + match_extend(blk, cand + 4 + sat, ip + 4 + sat, n, lane);
            ip += matched;
            op = emit_copy_packed(dst, op, mbase - cand, matched, lane);
            next_emit = ip;
            if (ip >= limit) {                                   // :388-389
                finished = true;
                break;
            }
            if (ip <= wbase + 64u) {                             // :391-392, position ip - 1 is in this window
                State::commit(tbl, win, 1ull << (ip - 1u - wbase), lane);
                st.inserted |= 1ull << (ip - 1u - wbase);
            } else {
                flags |= kPairPendingInsert;                     // ... or in a later one, whose owner inserts it
            }
            skip = 31;
        }

        PAIR_LAP(2);                                             // critical
        PAIR_ADD(5, 1);
        // ================= pass the token on =================
        if (finished) flags |= kPairDone;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            // row 1 first, then row 0 with `seq`: LDS operations of one wavefront execute in order, and this window's
            // table inserts were issued before both (pair_lds_order keeps the compiler from moving them)
            pair_lds_order();
            pair_row_store(tok_rows, 1, op, (uint32_t)st.inserted, (uint32_t)(st.inserted >> 32), flags);
            pair_lds_order();
            pair_row_store(tok_rows, 0, k + 1u, ip, skip, next_emit);
        }
        __builtin_amdgcn_wave_barrier();

        PAIR_LAP(3);                                             // token write
        // ================= POST(k) =================
        // the dwords of the next window of this wavefront first, so that they travel while the elements are stored
        const uint32_t xn0 = finished ? 0u : win.load_at(64u * (k + 2u), lane);
        const uint32_t xn1 = finished ? 0u : win.load_at(64u * (k + 3u), lane);
        if (pe.valid) pair_emit_segment(pe, win, st, blk, dst, lane);
        if (finished) {
            // emit_remainder (:405-410) and the size prefix (:412)
            if (next_emit < n) op = emit_literal(dst, op, blk + next_emit, n - next_emit, lane);
            if (lane == 0) {
                st32(dst, op - 4);
                *block_bytes_out = op;
            }
            break;
        }
        PAIR_LAP(4);                                             // post
        xprev = xnext;                                           // window k+1 is the predecessor of window k+2
        xcur = xn0;
        xnext = xn1;
        k += kPairWaves;
    }
#ifdef SNAPPY_PAIR_PROBE
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_pair_prof[i], pp[i]);
#endif
}

// Persistent workgroups of two wavefronts; blocks are drawn from *next_block (zeroed per launch), so this kernel can run
// beside compress_blocks_global_table_kernel on the same containers.  Dynamic LDS: pair_lds_bytes(block_size).
__global__ __launch_bounds__(kPairWaves * 64) void compress_blocks_pair_kernel(const K1Batch w, uint32_t block_size,
                                                                               uint32_t slot_stride, uint32_t* next_block)
{
    HIP_DYNAMIC_SHARED(uint8_t, pair_lds)
    const uint32_t num_blocks = w.first_block[w.count];
    const uint32_t table_bytes = 2u * pair_table_entries(block_size);
    uint16_t* table = reinterpret_cast<uint16_t*>(pair_lds);
    const uint32_t wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63u;   // uni(): wave-uniform for the compiler too
    lds_bytes_t dup_scratch = (lds_bytes_t)(pair_lds + table_bytes + wave * kPairScratchPerWave);
    lds_bytes_t prev_marks = dup_scratch + kDupSlots;
    volatile PairToken* tok = reinterpret_cast<volatile PairToken*>(pair_lds + table_bytes + kPairWaves * kPairScratchPerWave);
    volatile uint32_t* drawn = reinterpret_cast<volatile uint32_t*>(pair_lds + table_bytes + kPairWaves * kPairScratchPerWave +
                                                                    sizeof(PairToken));
#ifndef SNAPPY_EMU
    __builtin_amdgcn_s_setprio(3);      // few wavefronts, no table traffic: let the arbiter prefer them (as the LDS-table kernel)
#endif
    for (;;) {
        if (threadIdx.x == 0) *drawn = atomicAdd(next_block, 1u);
        __syncthreads();
        const uint32_t b = uni(*drawn);
        if (b >= num_blocks) break;
        const uint32_t c = batch_container_of(w, b);
        const uint32_t lb = b - w.first_block[c];
        const uint8_t* __restrict__ in = w.in[c];
        const uint64_t in_len = w.in_len[c];
        uint8_t* __restrict__ slot = w.slots[c] + (uint64_t)lb * slot_stride;
        uint32_t* __restrict__ bytes_out = w.block_bytes[c] + lb;
        const uint64_t start = (uint64_t)lb * block_size;
        const uint64_t left = in_len - start;
        const uint32_t n = (left < block_size) ? (uint32_t)left : block_size;
        compress_one_block_pair(in, start, in_len, n, slot, table, wave, lane, bytes_out, dup_scratch, prev_marks, tok);
        if (threadIdx.x == 0) atomicAdd(next_block + 4, 1u);     // statistics: blocks taken by LDS-table workgroups
        __syncthreads();                                         // both wavefronts are done with the table, the token, *drawn
    }
}

}  // namespace snappy_hip
