// snappy_k1_stream.hpp -- K1, stream form: the bulk parse of snappy_kernels.hpp re-timed as a software pipeline over the
// block's 64-byte windows.  Same decisions, same table contents at every read, same bytes as snappy_compress.c:284-413;
// what changes is WHEN things are issued and how much of it is scalar.  One wavefront per block as before.
//
// A wavefront alone on its SIMD (the LDS-table wavefronts: LDS capacity allows four per CU) issues one instruction per ~8
// cycles, and the bulk form spends a window's ~5,700 cycles partly waiting on its own round trips, partly on ~700
// instructions (profiles/r02_phase_profile_k1_forms.txt).  Only three steps of a window depend on the parse:
//     walk(W)  ->  commit(W)  ->  table read(W+1)
// The stream form keeps exactly that chain serial, issues the next window's table read and candidate loads right behind the
// commit, and does everything else underneath them:
//
//     finalize(W): candidate bytes have arrived -> HIT mask, match lengths, jump vector          (vector)
//     walk(W):     the hits and stop lanes of the window, one scalar step each (stream_walk)      (scalar chain)
//     commit(W):   every probed lane and every "ip - 1" lane inserts its position                 (masked stores)
//     window W+1:  registers rotate (its bytes were loaded a window ago), hashes, table read, candidate loads ISSUED
//     emit(W):     literal bytes, literal headers, copy elements by the lanes themselves          (under the loads)
//     analyse(W+1): which lanes share a table slot with an EARLIER lane of the window, and with which one;
//                  what a probe there would find if that lane had been inserted                   (under the loads)
//
// The walk visits only the lanes where something happens: every lane carries the distance to the next HIT-or-stop lane above
// it (misses in between are stride-1 scan probes, :336-348, and need no step of their own), a hit lane carries its copy
// length, and a stop lane carries 64, which ends the walk by the same carry-out that ends it at the window's end.  A stop lane
// is one whose candidate depends on this window's own inserts (its nearest earlier lane with the same hash lies at or above
// the cursor: the slot holds that lane if it was inserted, else what the gather read), or a hit whose 28 compared bytes all
// match.  The first lane of a group of equal hashes is NOT a stop (nothing can have changed its slot), nor is a lane whose
// partner lies below the cursor -- two thirds of the lanes the bulk form settles one by one.  Settling picks between two
// results that analyse() and finalize() computed for all lanes at once; only a longer chain of equal hashes, or a match that
// runs on, costs more.  The reference's skip counter (:339) is not tracked per step: the walk assumes stride 1 and the masks
// it leaves behind are checked afterwards (a run of more misses than the counter allows sends the window to the bulk form
// before anything was committed).
//
// Copies of 64+ bytes (several elements, :254-272) are taken in place: the window's segment in front of them is committed
// and emitted, the copy emitted by the generic emitter, and the pipeline restarts at the window the copy lands in.
// What this form does not take -- stride > 1 (incompressible data), the last 192 bytes of a block -- goes through
// bulk_run(), which hands back at the next window boundary.
#pragma once

namespace snappy_hip {

#ifdef SNAPPY_EMU
// emulator statistics: [0] windows taken, [1] windows sent back (stride widened), [2] copies of 64+ bytes, [3] stop lanes settled,
// [4] stream_run calls, [5] bulk_run calls from the stream form, [6] settled with the ballot (chains, hidden lanes, fallback),
// [7] windows whose duplicate analysis fell back to the race tables, [8] matches extended past the compared bytes,
// [9] settled from the partner lane, [10] settled from the gathered entry
inline unsigned long long g_stream_stats[16] = {0};
#define STREAM_STAT(i) do { if (lane == 0) ++g_stream_stats[i]; } while (0)
#else
#define STREAM_STAT(i) ((void)0)
#endif

// -DSNAPPY_PROF (tools/prof_stream.py): lap timers around the phases of the stream form, summed into g_prof by the lanes.
// Every cycle of a block lands in exactly one bucket; the counts are the number of laps.  Not defined in a product build.
#ifdef SNAPPY_PROF
__device__ unsigned long long g_prof[32];
// lane i of `acc` / `cnt` holds bucket i, so the timers cost four vector registers and no scalar ones
struct StreamProf {
    unsigned long long acc = 0, cnt = 0, last = 0;
};
#define PROF_START() (prof.last = __builtin_readcyclecounter())
#define PROF_LAP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == (i)) { prof.acc += now_ - prof.last; prof.cnt++; } prof.last = now_; } while (0)
#define PROF_WAIT() __builtin_amdgcn_s_waitcnt(0)
#define PROF_FLUSH() do { if (lane < 16) { atomicAdd(&g_prof[lane], prof.acc); atomicAdd(&g_prof[16 + lane], prof.cnt); } } while (0)
#elif defined(SNAPPY_MARK)
// -DSNAPPY_MARK (tools/isa_phase_counts.py): the lap points become comments in the assembly, so that the instructions of the
// stream form can be counted per phase in the .s file.  Not a product build.
struct StreamProf {};
#define PROF_START() asm volatile("; PHASE start")
#define PROF_LAP(i) asm volatile("; PHASE " #i)
#define PROF_WAIT() ((void)0)
#define PROF_FLUSH() ((void)0)
#else
struct StreamProf {};
#define PROF_START() ((void)0)
#define PROF_LAP(i) ((void)0)
#define PROF_WAIT() ((void)0)
#define PROF_FLUSH() ((void)0)
#endif

// a table type that declares kGathersOnce (the ceiling experiment's) wants every window gathered exactly once
template <class T, class = void> struct table_gathers_once : std::false_type {};
template <class T> struct table_gathers_once<T, std::void_t<decltype(T::kGathersOnce)>> : std::true_type {};

constexpr uint32_t kStreamRoom = 192;   // the stream form takes a window when base + kStreamRoom <= limit: every lane can be
                                        // probed (:342), has 32 bytes to load, and no copy of < 64 bytes reaches the limit
// slots per table of analyse() (two tables of dwords): 4 KiB of LDS beside the 32 KiB hash table, 2 KiB beside the 2 KiB
// slot filter of a global-table wavefront
#ifndef SNAPPY_STREAM_SLOTS_LDS
#define SNAPPY_STREAM_SLOTS_LDS 512
#endif
constexpr uint32_t kStreamSlotsLds = SNAPPY_STREAM_SLOTS_LDS, kStreamSlotsGlobal = 256;
__host__ __device__ constexpr uint32_t stream_scratch_bytes(uint32_t slots) { return 8u * slots; }

// unaligned 16-byte load (gfx950 runs with unaligned VMEM access enabled: one global_load_dwordx4)
__device__ __forceinline__ uint4 ld128(const uint8_t* p)
{
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

struct StreamWindow {
    uint32_t base;          // window base (multiple of 64)
    uint4 a, b;             // per lane: le32(blk + base + lane + 0 / 4 / 8 / 12) and (+ 16 / 20 / 24 / 28)
    uint32_t h0, e0;        // per lane: hash (:161-166) and content tag << 16 of a.x
};

// `last16` = the last offset a 16-byte load may start at; lanes past it read a clamped address (only in windows no form of
// the parse probes)
__device__ __forceinline__ void stream_load_window(StreamWindow& w, const uint8_t* __restrict__ blk, uint32_t base, uint32_t last16,
                                                   uint32_t lane)
{
    w.base = base;
    const uint32_t q = base + lane;
    w.a = ld128(blk + (q < last16 ? q : last16));
    w.b = ld128(blk + (q + 16u < last16 ? q + 16u : last16));
}
__device__ __forceinline__ void stream_hash_window(StreamWindow& w, uint32_t shift)
{
    const uint32_t prod = w.a.x * kHashMul;
    w.h0 = prod >> shift;
    w.e0 = (prod << (32 - shift)) & 0xffff0000u;
}

// the speculative gather of one window: table slot of every lane, 28 candidate bytes where the tag allows a hit
struct StreamGather {
    uint32_t ent;
    uint4 ka, kb;
    unsigned long long worthm;      // lanes whose candidate bytes were loaded (the tag allows a hit)
};
template <class Table>
__device__ __forceinline__ void stream_issue_gather(StreamGather& g, const Table& table, const StreamWindow& w,
                                                    const uint8_t* __restrict__ blk, uint32_t lane)
{
    const uint32_t mine_l = w.e0 | (w.base + lane);
    g.ent = table.load_lane(w.h0, mine_l);
    const bool worth = !Table::certain_miss(g.ent, mine_l);
    g.worthm = __ballot(worth);
    g.ka = g.kb = make_uint4(0, 0, 0, 0);
    if (worth) {                                               // candidate <= position, and position + 32 <= block length
        const uint8_t* __restrict__ c = blk + (g.ent & 0xffffu);
        g.ka = ld128(c);
        g.kb = ld128(c + 16);
    }
}

// ---------------------------------------------------------------------------
// analyse(W): for every lane the nearest EARLIER lane of the window with the same hash.  Two tables of kSlots dwords,
// indexed by overlapping parts of the hash (bits 0.. and 5..): every lane does an atomic max of (lane + 1) on its slot in
// each and gets back the highest lane that came before it.  LDS executes the lanes of one instruction in lane order; that is
// checked, not assumed (a returned lane at or above the own one sends the window to the race tables of the bulk form).  The
// lane returned is the nearest earlier lane in the same SLOT; if its hash is the same it is the nearest with the same HASH.
// If it is not (an alias), the other table decides; a lane with an alias in front of it in both tables may have a partner
// hidden behind them and is settled the long way (`cx`).  The tables are left zeroed.
// ---------------------------------------------------------------------------
#ifdef SNAPPY_EMU
__device__ __forceinline__ uint32_t lds_max_rtn(lds_words_t p, uint32_t v)
{
    const uint32_t old = *p;
    if (v > old) *p = v;
    return old;
}
#else
__device__ __forceinline__ uint32_t lds_max_rtn(lds_words_t p, uint32_t v)
{
    return __hip_atomic_fetch_max((__attribute__((address_space(3))) uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
#endif

struct StreamDup {
    uint32_t j1 = 0;                    // per lane of nf: nearest earlier lane with the same hash
    uint32_t extj = 0;                  // per lane with a partner: bytes of the 8 behind the key that match the partner's (0..8)
    unsigned long long nf = 0;          // lanes that have a partner
    unsigned long long hitj = 0;        // of those: the 4-byte keys are equal (a probe hits if the partner is the slot's content)
    unsigned long long deep = 0;        // of those: the partner has (or may have, cx) a partner itself: a chain of three or more
    unsigned long long cx = 0;          // lanes to be settled the long way (hidden partner, or every sharer after a fallback)
};

template <uint32_t kSlots>
__device__ __forceinline__ void stream_analyse(StreamDup& d, const StreamWindow& w, lds_bytes_t scratch, uint32_t lane)
{
    lds_words_t tab = (lds_words_t)scratch;
    const uint32_t sa = w.h0 & (kSlots - 1u);
    const uint32_t sb = kSlots + ((w.h0 >> 5) & (kSlots - 1u));
#ifdef SNAPPY_EMU
    // the emulator's fibers run in no particular order between two collectives: take turns, as the hardware's lanes do
    // (EMU_LDS_UNORDERED=1 leaves the order to the fibers and so exercises the fallback below)
    static const bool unordered = getenv("EMU_LDS_UNORDERED") != nullptr;
    uint32_t oa = 0, ob = 0;
    if (unordered) {
        oa = lds_max_rtn(tab + sa, lane + 1u);
        ob = lds_max_rtn(tab + sb, lane + 1u);
    } else {
        for (uint32_t turn = 0; turn < kWave; ++turn) {
            if (lane == turn) {
                oa = lds_max_rtn(tab + sa, lane + 1u);
                ob = lds_max_rtn(tab + sb, lane + 1u);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#else
    const uint32_t oa = lds_max_rtn(tab + sa, lane + 1u);
    const uint32_t ob = lds_max_rtn(tab + sb, lane + 1u);
#endif
    __builtin_amdgcn_wave_barrier();
    tab[sa] = 0;
    tab[sb] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t ja = oa - 1u, jb = ob - 1u;
    d = StreamDup();
    if (__ballot((oa && ja >= lane) || (ob && jb >= lane))) {    // not in lane order: the bulk form's race tables, every sharer a stop
        STREAM_STAT(7);
        d.cx = dup_slot_lanes(scratch, w.h0, lane);
        for (uint32_t i = lane; i < 2u * kSlots; i += kWave) tab[i] = 0;   // the race tables live in the same bytes
        __builtin_amdgcn_wave_barrier();
        return;
    }
    const uint32_t ha = (uint32_t)__shfl((int)w.h0, (int)(ja & 63u));
    const uint32_t hb = (uint32_t)__shfl((int)w.h0, (int)(jb & 63u));
    const unsigned long long OA = __ballot(oa != 0), OB = __ballot(ob != 0);
    const unsigned long long MA = OA & __ballot(ha == w.h0), MB = OB & __ballot(hb == w.h0);
    const bool ma = __builtin_amdgcn_inverse_ballot_w64(MA);
    d.j1 = ma ? ja : jb;                                         // (meaningful in the lanes of nf)
    d.nf = MA | MB;
    d.cx = OA & OB & ~d.nf;
    if (d.nf) {
        const uint32_t jl = d.j1 & 63u;
        const uint32_t xj = (uint32_t)__shfl((int)w.a.x, (int)jl);
        const uint32_t e1 = (uint32_t)__shfl((int)w.a.y, (int)jl) ^ w.a.y;
        const uint32_t e2 = (uint32_t)__shfl((int)w.a.z, (int)jl) ^ w.a.z;
        d.hitj = d.nf & __ballot(xj == w.a.x);
        d.extj = min_u32(min_u32(first_bit(e1), first_bit(e2) | 32u), 64u) >> 3;
        const unsigned long long chained = d.nf | d.cx;
        d.deep = d.nf & __ballot(((uint32_t)(chained >> jl) & 1u) != 0);
    }
}

// The walk of one window.  t = cursor lane - 64 (mod 2^32), so "t += advance" carries out exactly when the cursor leaves the
// window; s_bitset1_b64 / v_readlane_b32 / s_bfm_b64 only look at the low six bits of their index, which t and the lane
// share.  advv = per lane: copy length for a hit, distance to the next HIT-or-stop lane for a miss, 64 for a stop lane;
// clv = copy length for a hit, else 0.  V collects the lanes visited, COV the lanes covered by the copies taken.
// On gfx950 this is 7 instructions per visited lane (four lanes per trip of the loop); the C++ body is the same algorithm for the CPU emulator.
__device__ __forceinline__ void stream_walk(uint32_t advv, uint32_t clv, uint32_t& t, unsigned long long& V, unsigned long long& COV)
{
#ifdef SNAPPY_EMU
    for (;;) {
        const uint32_t l = t & 63u;
        V |= 1ull << l;
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)advv, (int)l);
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)clv, (int)l);
        if (c) COV |= ((1ull << c) - 1ull) << l;
        const uint32_t nt = t + a;
        const bool carry = nt < t;
        t = nt;
        if (carry) break;
    }
#else
    uint32_t a, c;
    unsigned long long m;
    asm volatile(                                // four lanes per trip: a branch that falls through is cheaper than one taken
        "1:\n"
        "  s_bitset1_b64 %[V], %[t]\n"
        "  v_readlane_b32 %[a], %[advv], %[t]\n"
        "  v_readlane_b32 %[c], %[clv], %[t]\n"
        "  s_bfm_b64 %[m], %[c], %[t]\n"        // ((1 << c) - 1) << t
        "  s_or_b64 %[COV], %[COV], %[m]\n"
        "  s_add_u32 %[t], %[t], %[a]\n"
        "  s_cbranch_scc1 2f\n"
        "  s_bitset1_b64 %[V], %[t]\n"
        "  v_readlane_b32 %[a], %[advv], %[t]\n"
        "  v_readlane_b32 %[c], %[clv], %[t]\n"
        "  s_bfm_b64 %[m], %[c], %[t]\n"        // ((1 << c) - 1) << t
        "  s_or_b64 %[COV], %[COV], %[m]\n"
        "  s_add_u32 %[t], %[t], %[a]\n"
        "  s_cbranch_scc1 2f\n"
        "  s_bitset1_b64 %[V], %[t]\n"
        "  v_readlane_b32 %[a], %[advv], %[t]\n"
        "  v_readlane_b32 %[c], %[clv], %[t]\n"
        "  s_bfm_b64 %[m], %[c], %[t]\n"        // ((1 << c) - 1) << t
        "  s_or_b64 %[COV], %[COV], %[m]\n"
        "  s_add_u32 %[t], %[t], %[a]\n"
        "  s_cbranch_scc1 2f\n"
        "  s_bitset1_b64 %[V], %[t]\n"
        "  v_readlane_b32 %[a], %[advv], %[t]\n"
        "  v_readlane_b32 %[c], %[clv], %[t]\n"
        "  s_bfm_b64 %[m], %[c], %[t]\n"        // ((1 << c) - 1) << t
        "  s_or_b64 %[COV], %[COV], %[m]\n"
        "  s_add_u32 %[t], %[t], %[a]\n"
        "  s_cbranch_scc0 1b\n"
        "2:\n"
        : [t] "+s"(t), [V] "+s"(V), [COV] "+s"(COV), [a] "=&s"(a), [c] "=&s"(c), [m] "=&s"(m)
        : [advv] "v"(advv), [clv] "v"(clv)
        : "scc");
#endif
}

#ifndef SNAPPY_EMU
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
    return (unsigned long long)uni((uint32_t)v) | ((unsigned long long)uni((uint32_t)(v >> 32)) << 32);
}
#else
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) { return v; }
#endif

// what a copy that lands on lane `land` adds to its advance: the distance from there to the next lane where something
// happens, when the landing lane itself is a plain miss inside the window (see finalize)
__device__ __forceinline__ uint32_t stream_past(uint32_t dist, unsigned long long inter, uint32_t land)
{
    if (land >= kWave || ((inter >> land) & 1ull)) return 0;
    return (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)land);
}

// m has a run of 34 or more consecutive set bits
__device__ __forceinline__ bool has_run_of_34(unsigned long long m)
{
    unsigned long long r = m & (m >> 1);      // runs of >= 2
    r &= r >> 2;                              // >= 4
    r &= r >> 4;                              // >= 8
    r &= r >> 8;                              // >= 16
    r &= r >> 16;                             // >= 32
    return (r & (m >> 32) & (m >> 33)) != 0;
}

// insert position `pos` whose 4 bytes are x (scalar form of StreamWindow's h0 / e0)
template <class Table>
__device__ __forceinline__ void stream_put(const Table& table, uint32_t x, uint32_t pos, uint32_t shift, uint32_t lane)
{
    const uint32_t prod = x * kHashMul;
    if constexpr (Table::kCollectiveStore)
        table.store_masked(1ull, prod >> shift, pos, lane);
    else
        table.put(prod >> shift, ((prod << (32 - shift)) & 0xffff0000u) | pos, lane);
}
// insert the positions of the lanes in `m` of window `w` (each its own slot)
template <class Table>
__device__ __forceinline__ void stream_store(const Table& table, const StreamWindow& w, unsigned long long m, uint32_t lane)
{
    if constexpr (Table::kCollectiveStore) {
        table.store_masked(m, w.h0, w.base + lane, lane);
    } else {
        if (__builtin_amdgcn_inverse_ballot_w64(m)) table.store_lane(w.h0, w.e0 | (w.base + lane));
        __builtin_amdgcn_wave_barrier();
    }
}

// What the stream form hands to "the other half" of a window's work -- the duplicate analysis in front of the parse and the
// emission behind it.  One wavefront per block: both are done in place.  (The Mate parameter is the seam round 3's duo form used -- a second wavefront doing both: profiles/HISTORY.md -- and round 4's free-table experiment uses: csrc/ablation/k1_oracle_table.hpp.)
__device__ __forceinline__ void stream_emit(uint8_t* __restrict__ dst, const uint8_t* __restrict__ blk, uint32_t& op, uint32_t& next_emit,
                                            uint32_t base, uint32_t x0, uint32_t ent, uint32_t extv, unsigned long long H,
                                            unsigned long long COV, bool by_copy, uint32_t r_out, bool long_copy, uint32_t ip,
                                            uint32_t long_cand, uint32_t long_len, uint32_t lane)
{
    if (H) emit_segment(dst, blk, op, next_emit, base, x0, ent, extv, H, COV, by_copy, r_out, lane);
    if (long_copy) {
        if (ip > next_emit) op = emit_literal_windowed(dst, op, blk, next_emit, ip - next_emit, base, x0, lane);
        op = emit_copy(dst, op, ip - long_cand, long_len, lane);
        next_emit = ip + long_len;
    }
}

struct SoloMate {
    static constexpr bool kAnalysesInPlace = true;               // analyse()'s tables live in the caller's scratch
    __device__ __forceinline__ void begin(uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ void drain(uint32_t&, uint32_t&, uint32_t) {}
    template <uint32_t kSlots>
    __device__ __forceinline__ void analysis(StreamDup& d, const StreamWindow& w, lds_bytes_t scratch, uint32_t lane)
    {
        stream_analyse<kSlots>(d, w, scratch, lane);
    }
    __device__ __forceinline__ void emit(uint8_t* __restrict__ dst, const uint8_t* __restrict__ blk, uint32_t& op, uint32_t& next_emit,
                                         uint32_t base, uint32_t x0, uint32_t ent, uint32_t extv, unsigned long long H,
                                         unsigned long long COV, bool by_copy, uint32_t r_out, bool long_copy, uint32_t ip,
                                         uint32_t long_cand, uint32_t long_len, uint32_t lane)
    {
        stream_emit(dst, blk, op, next_emit, base, x0, ent, extv, H, COV, by_copy, r_out, long_copy, ip, long_cand, long_len, lane);
    }
};

// Stream form from `ps` on.  Precondition: stride 1 (ps.skip < 64) and the window of ps.ip is eligible
// (base + kStreamRoom <= limit).  Takes windows until one is not eligible, a run of misses widens the stride, or the
// scan is over behind a long copy; `ps` is then exactly the reference's state in front of the probe at ps.ip.  May
// return without progress (first window sent back): the caller follows with bulk_run().
template <class Table, uint32_t kSlots, class Mate>
__device__ __forceinline__ void stream_run(const uint8_t* __restrict__ blk, uint32_t avail, uint32_t n, uint32_t shift,
                                           uint8_t* __restrict__ dst, const Table table, uint32_t lane, lds_bytes_t dup_scratch,
                                           ParseState& ps, StreamProf& prof, Mate& mate)
{
    const uint32_t limit = n - kInputMargin;
    const uint32_t last16 = avail - 16u;
    uint32_t ip = ps.ip, skip = ps.skip, op = ps.op, next_emit = ps.next_emit;
    mate.begin(op, next_emit, lane);

    for (;;) {   // ---- (re)start of the pipeline at the window of ip ----
        StreamWindow cur, nxt;
        stream_load_window(cur, blk, ip & ~63u, last16, lane);
        stream_hash_window(cur, shift);
        StreamGather g;
        stream_issue_gather(g, table, cur, blk, lane);
        __builtin_amdgcn_sched_barrier(0);
        stream_load_window(nxt, blk, cur.base + 64u, last16, lane);
        StreamDup dup;
        mate.template analysis<kSlots>(dup, cur, dup_scratch, lane);
        PROF_LAP(0);                                             // 0: priming (first window's loads issued, analysed)
        bool restart = false;                                    // a long copy was taken: the pipeline restarts where it landed

        for (;;) {
            const uint32_t r0 = ip - cur.base;
            const uint32_t b_in = 64u - skip;                    // stride-1 probes left before :339 widens the stride (1..33)
            PROF_WAIT();
            PROF_LAP(1);                                         // 1: waiting for the candidate loads (probe builds only: all of them)

            // ---------------- finalize(W): what a probe at each lane would find ----------------
            unsigned long long hitm = __ballot(g.ka.x == cur.a.x) & g.worthm;
            // bytes of the 24 behind the key that match the candidate's: first differing bit of the 192, found without a branch
            // (first_bit: 0..31, or all ones for a zero word, which the ORs keep as "none" and the minimum ignores)
            const uint32_t t1 = min_u32(first_bit(g.ka.y ^ cur.a.y), first_bit(g.ka.z ^ cur.a.z) | 32u);
            const uint32_t t2 = min_u32(first_bit(g.ka.w ^ cur.a.w), first_bit(g.kb.x ^ cur.b.x) | 32u) | 64u;
            const uint32_t t3 = min_u32(first_bit(g.kb.y ^ cur.b.y), first_bit(g.kb.z ^ cur.b.z) | 32u) | 128u;
            uint32_t extv = min_u32(min_u32(t1, t2), min_u32(t3, 192u)) >> 3;
            uint32_t ent = g.ent & 0xffffu;
            // stop lanes: the partner lies at or above the cursor (below it, nothing was inserted there since the gather: the
            // gathered entry stands), hidden partners, and hits whose compared bytes all match
            const unsigned long long pend = dup.nf & __ballot(dup.j1 >= r0);
            const unsigned long long satm = hitm & __ballot(extv == 24u);
            unsigned long long stopm = pend | dup.cx | satm;
            const unsigned long long above_r0 = ~0ull << r0;
            // jump vector: the next lane above this one where something happens (HIT or stop), else the window's end
            const unsigned long long inter = hitm | stopm;
            const unsigned long long up = (lane < 63u) ? (inter >> (lane + 1u)) : 0ull;
            const uint32_t dist = up ? (uint32_t)__builtin_ctzll(up) + 1u : 64u - lane;
            const bool stopl = __builtin_amdgcn_inverse_ballot_w64(stopm);
            const bool hitl = __builtin_amdgcn_inverse_ballot_w64(hitm);
            // what the lane does if the gathered entry stands (advt / clt) and if its partner is the slot's content (advj / clj);
            // a stop lane carries 64 / 0 until it is settled with one of the two
            // A copy that lands on a lane where nothing happens goes straight on to the next lane where something does: the
            // landing lane is a plain miss (a scan probe, :336-348) and needs no step of its own.  `past` = what to add to a
            // copy of the given length from this lane: the landing lane's own distance, 0 when the landing lane is a hit or
            // a stop (or beyond the window)
            const uint32_t clt = hitl ? 4u + extv : 0u;
            const bool hitjl = __builtin_amdgcn_inverse_ballot_w64(dup.hitj);
            const uint32_t clj = hitjl ? 4u + dup.extj : 0u;
            const uint32_t land_t = lane + clt, land_j = lane + clj;
            const uint32_t dist_t = (uint32_t)__shfl((int)dist, (int)(land_t & 63u));
            const uint32_t dist_j = (uint32_t)__shfl((int)dist, (int)(land_j & 63u));
            const bool plain_t = land_t < 64u && !((inter >> (land_t & 63u)) & 1ull);
            const bool plain_j = land_j < 64u && !((inter >> (land_j & 63u)) & 1ull);
            const uint32_t advt = hitl ? clt + (plain_t ? dist_t : 0u) : dist;
            const uint32_t advj = hitjl ? clj + (plain_j ? dist_j : 0u) : dist;
            const unsigned long long satjm = dup.hitj & __ballot(dup.extj == 8u);
            uint32_t clv = stopl ? 0u : clt;
            uint32_t advv = stopl ? 64u : advt;
            PROF_LAP(2);                                         // 2: finalize

            // ---------------- walk(W) ----------------
            unsigned long long V = 0, COV = 0;
            uint32_t t = r0 - 64u;
            uint32_t r_out = 0;                                  // where the cursor ends, relative to the window base
            bool long_copy = false;                              // the lane at r_out starts a copy of 64+ bytes
            uint32_t long_cand = 0, long_len = 0;
            for (;;) {
                // (the structurizer turns these loop-carried scalars into vector PHIs around the settle's branches; read them
                // back before the walk's inline assembly, which takes them in SGPRs)
                t = uni(t);
                stream_walk(advv, clv, t, V, COV);
                const uint32_t s = 63u - (uint32_t)__builtin_clzll(V);   // the lane visited last
                if (!((stopm >> s) & 1ull)) {                    // left the window: behind a copy, or by a miss at lane 63
                    r_out = t + 64u;
                    break;
                }
                PROF_LAP(3);                                     // 3: walk proper
                // ---- the walk stands on a stop lane: settle that one probe ----
                STREAM_STAT(3);
                const unsigned long long bit_s = 1ull << s;
                const unsigned long long below_s = bit_s - 1ull;
                const unsigned long long interior_s = COV & ~(V & hitm & below_s);
                uint32_t how = 0;                                // 0: the gathered entry stands, 1: the partner lane, 2: ask the ballot
                uint32_t j = 0;
                if (pend & bit_s) {
                    // the slot holds the partner if it was inserted since the gather: probed, or the last lane of a copy --
                    // i.e. NOT a covered lane whose successor is covered too
                    j = (uint32_t)__builtin_amdgcn_readlane((int)dup.j1, (int)s);
                    const unsigned long long not_inserted = interior_s & (interior_s >> 1);
                    if (!((not_inserted >> j) & 1ull)) how = 1;
                    else if (dup.deep & bit_s) how = 2;          // a further lane of the chain may have been
                }
                if (dup.cx & bit_s) how = 2;
                // what the lane does: (advance, copy length) for the walk, (candidate, length - 4) for the emission, and
                // whether the compared bytes all matched (the match may run on)
                uint32_t adv_s, cl_s, cand_s = 0, sat_s = 24;
                bool hit_s, more_s;
                if (how == 2) {
                    // the long way: the latest inserted lane with this hash, by ballot (as the bulk form does for every sharer)
                    STREAM_STAT(6);
                    const unsigned long long ins = ((above_r0 & below_s & ~interior_s) | (COV & ~(interior_s >> 1))) & below_s;
                    const uint32_t hs = (uint32_t)__builtin_amdgcn_readlane((int)cur.h0, (int)s);
                    const unsigned long long J = __ballot(cur.h0 == hs) & ins;
                    how = 0;
                    if (J) {
                        j = 63u - (uint32_t)__builtin_clzll(J);
                        hit_s = (uint32_t)__builtin_amdgcn_readlane((int)cur.a.x, (int)s) == (uint32_t)__builtin_amdgcn_readlane((int)cur.a.x, (int)j);
                        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)cur.a.y, (int)s) ^ (uint32_t)__builtin_amdgcn_readlane((int)cur.a.y, (int)j);
                        const uint32_t e2 = (uint32_t)__builtin_amdgcn_readlane((int)cur.a.z, (int)s) ^ (uint32_t)__builtin_amdgcn_readlane((int)cur.a.z, (int)j);
                        const uint32_t ext_s = e1 ? ((uint32_t)__builtin_ctz(e1) >> 3) : (e2 ? 4u + ((uint32_t)__builtin_ctz(e2) >> 3) : 8u);
                        cand_s = cur.base + j;
                        sat_s = 8;
                        more_s = hit_s && ext_s == 8u;
                        cl_s = hit_s ? 4u + ext_s : 0u;
                        adv_s = hit_s ? cl_s : (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)s);
                        if (hit_s && !more_s) adv_s += stream_past(dist, hitm | stopm, s + cl_s);
                        how = 3;
                    }
                }
                if (how == 1) {                                  // both results were computed for all lanes at once
                    STREAM_STAT(9);
                    hit_s = (dup.hitj & bit_s) != 0;
                    more_s = (satjm & bit_s) != 0;
                    adv_s = (uint32_t)__builtin_amdgcn_readlane((int)advj, (int)s);
                    cl_s = (uint32_t)__builtin_amdgcn_readlane((int)clj, (int)s);
                    cand_s = cur.base + j;
                    sat_s = 8;
                } else if (how == 0) {
                    STREAM_STAT(10);
                    hit_s = (hitm & bit_s) != 0;
                    more_s = (satm & bit_s) != 0;
                    adv_s = (uint32_t)__builtin_amdgcn_readlane((int)advt, (int)s);
                    cl_s = (uint32_t)__builtin_amdgcn_readlane((int)clt, (int)s);
                }
                if (more_s) {
                    STREAM_STAT(8);
                    if (how == 0) cand_s = (uint32_t)__builtin_amdgcn_readlane((int)ent, (int)s);
                    cl_s = 4u + sat_s + match_extend(blk, cand_s + 4u + sat_s, cur.base + s + 4u + sat_s, n, lane);
                    adv_s = cl_s <= 63u ? cl_s + stream_past(dist, hitm | stopm, s + cl_s) : cl_s;
                    if (cl_s > 63u) {                            // more than one copy element (:254-272): taken behind this segment
                        STREAM_STAT(2);
                        V &= ~bit_s;
                        r_out = s;
                        long_copy = true;
                        long_cand = cand_s;
                        long_len = cl_s;
                        break;
                    }
                }
                stopm &= ~bit_s;
                hitm = hit_s ? (hitm | bit_s) : (hitm & ~bit_s);
                if (lane == s) {
                    advv = adv_s;
                    clv = cl_s;
                    if (how != 0) ent = cand_s;                  // (the gathered entry is there already)
                    if (how != 0 || more_s) extv = cl_s - 4u;
                }
                t = s - 64u;
                PROF_LAP(4);                                     // 4: settles
            }
            PROF_LAP(3);

            // ---------------- what the walk did, as masks ----------------
            const uint32_t r_end = r_out < kWave ? r_out : kWave;
            const unsigned long long range = r_end > r0 ? (above_r0 & lanes_below(r_end)) : 0ull;
            const unsigned long long H = V & hitm & range;
            const unsigned long long interior = COV & ~H;
            const unsigned long long probed = range & ~interior;
            const unsigned long long miss = probed & ~H;
            const uint32_t top = probed ? 63u - (uint32_t)__builtin_clzll(probed) : 0u;    // the lane probed last
            const bool by_copy = probed && ((H >> top) & 1ull);                             // the cursor stands right behind a copy
            // The reference's skip counter (:339): the first run of misses may use up b_in probes, every later one 33.  With no
            // more than b_in misses in the whole window nothing can have gone wrong; else look at the runs.
            uint32_t skip_out = skip;
            bool widened = false;
            if (probed) {
                const uint32_t tail = by_copy ? 0u : (uint32_t)__builtin_clzll(~(miss << (63u - top)));   // misses ending at `top`
                const bool copy_before = (H & ((1ull << top) - 1ull)) != 0;
                const uint32_t allowed = copy_before ? 33u : b_in;
                if ((uint32_t)__builtin_popcountll(miss) > b_in) {
                    const uint32_t lead = (uint32_t)__builtin_ctzll(~(miss >> r0));            // misses from r0 on
                    widened = lead > b_in || has_run_of_34(miss) || tail > allowed;
                }
                skip_out = by_copy ? 31u : 64u - (allowed - tail);
            }
            if (widened) {                                       // nothing committed, nothing emitted: ps still describes r0
                STREAM_STAT(1);
                break;
            }
            STREAM_STAT(0);

            // ---------------- commit(W): every probed lane (:346-347, :397) and every "ip - 1" lane (:391-392) ----------------
            unsigned long long endl = COV & ~(interior >> 1) & range;
            if (by_copy && r_out > kWave) endl &= ~(1ull << 63); // the last copy ends in the next window: its lane is due there
            unsigned long long C = probed | endl;
            if (long_copy) C |= 1ull << r_out;                   // the probe that found the long copy (:346-347)
            // Lanes without a partner have slots of their own (two of them with equal hashes: the later one would have the
            // earlier one as its partner); then the lanes whose partner has none (again distinct among themselves, and later
            // than their partners); the few that are left -- longer chains, hidden partners -- one by one in lane order.
            {
                const unsigned long long later = dup.nf | dup.cx;
                stream_store(table, cur, C & ~later, lane);
                const unsigned long long second = C & dup.nf & ~dup.deep & ~dup.cx;
                if (second) stream_store(table, cur, second, lane);
                for (unsigned long long dd = C & (dup.deep | dup.cx); dd; dd &= dd - 1) stream_store(table, cur, dd & (~dd + 1), lane);
            }
            ip = cur.base + r_out;
            skip = skip_out;
            PROF_LAP(5);                                         // 5: masks, skip check, commit

            // ---------------- window W+1: table read and candidate loads go out before anything else ----------------
            const bool go = !long_copy && skip < 64u && cur.base + 64u + kStreamRoom <= limit;
            const StreamWindow old = cur;
            const uint32_t old_ent = ent, old_extv = extv;
            if (go) {
                cur = nxt;
                stream_hash_window(cur, shift);
                if (r_out > kWave) {                             // :391-392 for the copy that ended in this window
                    stream_store(table, cur, 1ull << (r_out - kWave - 1u), lane);
                }
                __builtin_amdgcn_sched_barrier(0);
                stream_load_window(nxt, blk, cur.base + 64u, last16, lane);
                stream_issue_gather(g, table, cur, blk, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            PROF_LAP(6);                                         // 6: next window's table read and loads issued

            // ---------------- emit(W) ----------------
            // (the segment's elements, and behind them the copy of 64+ bytes at ip if there is one: the literal in front of it,
            // :355, and its elements, :254-272)
            if (H || long_copy)
                mate.emit(dst, blk, op, next_emit, old.base, old.a.x, old_ent, old_extv, H, COV, by_copy, r_out, long_copy, ip, long_cand,
                          long_len, lane);
            PROF_LAP(7);                                         // 7: emission

            if (long_copy) {
                ip += long_len;                                  // the cursor behind the copy
                skip = 31;
                if (ip < limit) {                                // :388-392
                    stream_put(table, uld32(blk + ip - 1u), ip - 1u, shift, lane);
                    restart = (ip & ~63u) + kStreamRoom <= limit;
                }
                PROF_LAP(9);                                     // 9: long copies
                break;
            }
            if (!go) {
                if (r_out > kWave)                               // the "ip - 1" lane of a copy that ended beyond this window
                    stream_put(table, uld32(blk + ip - 1u), ip - 1u, shift, lane);
                break;
            }
            // ---------------- analyse(W+1) ----------------
            mate.template analysis<kSlots>(dup, cur, dup_scratch, lane);
            PROF_LAP(8);                                         // 8: duplicate-slot analysis
        }
        if (!restart) break;
    }
    mate.drain(op, next_emit, lane);
    ps.ip = ip;
    ps.skip = skip;
    ps.op = op;
    ps.next_emit = next_emit;
}

// One block: the stream form for every window it takes, the bulk form for the rest.
template <class Table, uint32_t kSlots, class Mate>
__device__ __forceinline__ void compress_one_block_stream(const uint8_t* __restrict__ base16, uint64_t start, uint64_t in_len,
                                                          uint32_t n, uint8_t* __restrict__ dst, const Table table_in, uint32_t lane,
                                                          uint32_t* __restrict__ block_bytes_out, lds_bytes_t dup_scratch, Mate& mate)
{
    const uint8_t* __restrict__ blk = base16 + start;
    const uint32_t ts = table_entries_for(n);                    // get_hash_table, :139-146 (+ shift, :288)
    const uint32_t shift = (uint32_t)__builtin_clz(ts) + 1;
    const uint32_t e_zero = (n >= kInputMargin) ? (((uld32(blk) * kHashMul) << (32 - shift)) & 0xffff0000u) : 0u;
    const Table table = table_in.with_empty(e_zero);
    if (n >= kInputMargin) table.init(ts, e_zero, lane);
    if (Mate::kAnalysesInPlace)
        for (uint32_t i = lane; i < 2u * kSlots; i += kWave) ((lds_words_t)dup_scratch)[i] = 0;   // analyse() keeps its tables zeroed
    __builtin_amdgcn_wave_barrier();
    ParseState ps;
    StreamProf prof;
    PROF_START();
    if (n >= kInputMargin) {  // :301
        const uint32_t limit = n - kInputMargin;
        const uint64_t left = in_len - start;
        const uint32_t avail = (left < 0x7fffffffull) ? (uint32_t)left : 0x7fffffffu;
        for (;;) {
            if (ps.skip < 64u && (ps.ip & ~63u) + kStreamRoom <= limit) {
                STREAM_STAT(4);
                PROF_LAP(12);                                    // 12: table clear, loop glue
                stream_run<Table, kSlots, Mate>(blk, avail, n, shift, dst, table, lane, dup_scratch, ps, prof, mate);
                if (ps.ip >= limit) break;                       // the scan ended behind a long copy (:388-389)
            }
            STREAM_STAT(5);
            // at least one step of the bulk form, then on to the next window boundary at stride 1
            const bool over = bulk_run<Table, 64, table_gathers_once<Table>::value>(blk, avail, n, shift, dst, table, lane, dup_scratch, ps,
                                                                                    (ps.ip | 63u) + 1u);
            if (Mate::kAnalysesInPlace)
                for (uint32_t i = lane; i < 2u * kSlots; i += kWave) ((lds_words_t)dup_scratch)[i] = 0;   // its race tables used the bytes
            __builtin_amdgcn_wave_barrier();
            PROF_LAP(11);                                        // 11: the bulk form
            if (over) break;
        }
    }
    finish_block(blk, n, dst, ps, lane, block_bytes_out);
    PROF_LAP(12);
    PROF_FLUSH();
}

}  // namespace snappy_hip
