// k1_duo_form.hpp -- K1, duo form (round 3; ablation code, compiled only with -DSNAPPY_ABLATION): the stream form of snappy_k1_stream.hpp on TWO wavefronts per block.
//
// A wavefront alone on its SIMD issues one dependent instruction per ~8 cycles (profiles/r01_microbench/issue_probe.log),
// and the stream form's window is ~600 of them: the LDS-table wavefronts -- four per CU by LDS capacity -- are bound by
// their own instruction stream, not by memory (profiles/r03_phase_profile_stream_form.txt: 230 cycles of a window's 5,900
// wait for loads).  A third of that stream does not depend on the parse, or only consumes it:
//     analyse(W): which lanes of a window share a table slot with an earlier lane     (needs the window's bytes only)
//     emit(W):    literal bytes, headers, copy elements of a parsed window            (needs the parse's masks only)
// The duo form gives those two to a second wavefront of the same workgroup, the MATE.  The PARSER keeps gather, finalize,
// walk, commit (stream_run with DuoParser as its Mate); the mate analyses the windows ahead of it and emits the windows
// behind it.  They talk through a mailbox in LDS:
//     ana[2]: results of analyse() for the window at ana_base[i] - 1, written by the mate, consumed (and freed) by the parser;
//     seg[2]: one parsed window each -- masks, cursor, per-lane candidate and length, a trailing copy of 64+ bytes --
//             written by the parser, consumed by the mate in order;
//     op / next_emit: the emission state; the mate's while segments are pending, readable by the parser once drained
//             (the bulk form's steps on the parser, and the block's remainder, emit by themselves).
// LDS operations of one wavefront execute in order, so a record is complete when its sequence word is seen; the parse, the
// table and every output byte are those of the one-wavefront stream form (same code, other Mate).
// Measured ceiling of the split (parser without emission and analysis, wrong bytes: profiles/r03_stream_split_bound_experiment.txt):
// LDS-table kernel alone 30.2 -> 44.6 GB/s.  Measured RESULT (profiles/r03_duo_form.txt): 30.0 GB/s alone, 62.6-64.8 in the mix
// against 68.3 -- the parser's side of the handshake (publish a segment: ~580 cycles, wait for / read an analysis: ~710) costs
// what the moved work saved (~720 + ~820 cycles per window): on a wavefront that issues one instruction per 8-13 cycles an
// 80-instruction protocol is as long as the 120 instructions it replaces.  Bit-exact (emulator under shuffled schedules, GPU);
// not shipped.
#pragma once

namespace snappy_hip {

constexpr uint32_t kDuoSegSlots = 2;

// mailbox layout, in dwords
constexpr uint32_t kDuoSegHead = 0;       // parser: segments published
constexpr uint32_t kDuoSegTail = 1;       // mate: segments emitted
constexpr uint32_t kDuoWantBase = 2;      // parser: base of the window it will need analysed next (the mate also prepares the one after)
constexpr uint32_t kDuoQuit = 3;          // parser: the block's parse is over
constexpr uint32_t kDuoOp = 4;            // emission state (see above)
constexpr uint32_t kDuoNextEmit = 5;
constexpr uint32_t kDuoAnaBase = 6;       // [2] mate: window base + 1 held by ana slot i, 0 = free
constexpr uint32_t kDuoAna = 8;           // [2] x { lane[64]: j1 | extj << 8 ; nf, hitj, deep, cx as 8 dwords }
constexpr uint32_t kDuoAnaDwords = 64 + 8;
constexpr uint32_t kDuoSeg = kDuoAna + 2 * kDuoAnaDwords;   // [kDuoSegSlots] x { lane[64]: ent | extv << 16 ; 10 dwords }
constexpr uint32_t kDuoSegDwords = 64 + 12;
constexpr uint32_t kDuoBoxDwords = kDuoSeg + kDuoSegSlots * kDuoSegDwords;
__host__ __device__ constexpr uint32_t duo_box_bytes() { return 4u * kDuoBoxDwords; }
constexpr uint32_t kDuoNoWindow = 0xffffffffu;

__device__ __forceinline__ void duo_put64(lds_words_t p, unsigned long long v)
{
    p[0] = (uint32_t)v;
    p[1] = (uint32_t)(v >> 32);
}
// (uniform address: every lane reads the same two words; read back as wave-uniform values)
__device__ __forceinline__ unsigned long long duo_get64(lds_words_t p)
{
    const uint32_t lo = p[0], hi = p[1];
    return (unsigned long long)uni(lo) | ((unsigned long long)uni(hi) << 32);
}
// the 64-bit mask held as two dwords in lanes k, k + 1 of v
__device__ __forceinline__ unsigned long long duo_lanes64(uint32_t v, uint32_t k)
{
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)v, (int)k) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)v, (int)(k + 1u)) << 32);
}
__device__ __forceinline__ void duo_nap()
{
#ifndef SNAPPY_EMU
    __builtin_amdgcn_s_sleep(1);
#endif
}

// ---------------------------------------------------------------------------
// the parser's side
// ---------------------------------------------------------------------------
struct DuoParser {
    static constexpr bool kAnalysesInPlace = false;              // the mate analyses, in tables of its own
    lds_words_t box;
    uint32_t limit;               // n - 15: windows with base + kStreamRoom <= limit are worth preparing
    uint32_t head = 0;            // segments published (the parser is the only writer of box[kDuoSegHead])
    uint32_t tail_seen = 0;       // the mate's count when it was last read: re-read only when the ring looks full

    __device__ __forceinline__ void begin(uint32_t op, uint32_t next_emit, uint32_t lane)
    {
        if (lane == 0) {                                         // (nothing is pending: the parser drained before it left)
            box[kDuoOp] = op;
            box[kDuoNextEmit] = next_emit;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // wait until the mate has emitted everything, then take the emission state back
    __device__ __forceinline__ void drain(uint32_t& op, uint32_t& next_emit, uint32_t lane)
    {
        (void)lane;
        while ((tail_seen = uni(box[kDuoSegTail])) != head) duo_nap();
        const uint32_t a = box[kDuoOp], b = box[kDuoNextEmit];   // (both reads in flight together)
        op = uni(a);
        next_emit = uni(b);
    }
    template <uint32_t kSlots>
    __device__ __forceinline__ void analysis(StreamDup& d, const StreamWindow& w, lds_bytes_t, uint32_t lane)
    {
        const uint32_t slot = (w.base >> 6) & 1u;
        if (lane == 0) box[kDuoWantBase] = w.base;
        __builtin_amdgcn_wave_barrier();
        for (;;) {
            const uint32_t have = uni(box[kDuoAnaBase + slot]);
            if (have == w.base + 1u) break;
            if (have != 0 && lane == 0) box[kDuoAnaBase + slot] = 0;     // another window's (prepared before a jump): not needed
            duo_nap();
        }
        // one round trip for the whole record: the per-lane word and, in lanes 0..7, the four masks
        lds_words_t a = box + kDuoAna + slot * kDuoAnaDwords;
        const uint32_t packed = a[lane];
        const uint32_t mrow = a[64 + (lane & 7u)];
        d.j1 = packed & 0xffu;
        d.extj = packed >> 8;
        d.nf = duo_lanes64(mrow, 0);
        d.hitj = duo_lanes64(mrow, 2);
        d.deep = duo_lanes64(mrow, 4);
        d.cx = duo_lanes64(mrow, 6);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            box[kDuoAnaBase + slot] = 0;                         // consumed
            box[kDuoWantBase] = w.base + 64u;                    // the window after this one is the likely next
        }
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void emit(uint8_t* __restrict__, const uint8_t* __restrict__, uint32_t&, uint32_t&, uint32_t base, uint32_t,
                                         uint32_t ent, uint32_t extv, unsigned long long H, unsigned long long COV, bool by_copy,
                                         uint32_t r_out, bool long_copy, uint32_t ip, uint32_t long_cand, uint32_t long_len, uint32_t lane)
    {
        while (head - tail_seen >= kDuoSegSlots) {               // both slots still to be emitted, as far as we know: look again
            tail_seen = uni(box[kDuoSegTail]);
            if (head - tail_seen >= kDuoSegSlots) duo_nap();
        }
        lds_words_t r = box + kDuoSeg + (head % kDuoSegSlots) * kDuoSegDwords;
        r[lane] = (ent & 0xffffu) | (extv << 16);
        // the scalars of the record, one lane each (one store for all of them)
        uint32_t sv = (uint32_t)H;
        sv = lane == 1 ? (uint32_t)(H >> 32) : sv;
        sv = lane == 2 ? (uint32_t)COV : sv;
        sv = lane == 3 ? (uint32_t)(COV >> 32) : sv;
        sv = lane == 4 ? base : sv;
        sv = lane == 5 ? r_out : sv;
        sv = lane == 6 ? ((by_copy ? 1u : 0u) | (long_copy ? 2u : 0u)) : sv;
        sv = lane == 7 ? ip : sv;
        sv = lane == 8 ? long_cand : sv;
        sv = lane == 9 ? long_len : sv;
        if (lane < 10u) r[64 + lane] = sv;
        __builtin_amdgcn_wave_barrier();
        ++head;
        if (lane == 0) box[kDuoSegHead] = head;                  // after the record (LDS operations of a wavefront stay in order)
        __builtin_amdgcn_wave_barrier();
    }
};

// ---------------------------------------------------------------------------
// the mate: emits the segments the parser publishes, analyses the windows it will need
// ---------------------------------------------------------------------------
// analyse one window for the parser and publish the result; returns the window's first dword per lane (byte 0 = the window's
// byte at that lane), which the emission of that window will want later
template <uint32_t kSlots>
__device__ __forceinline__ uint32_t duo_prepare(const uint8_t* __restrict__ blk, uint32_t base, uint32_t last16, uint32_t shift,
                                                uint32_t lane, lds_bytes_t dup_scratch, lds_words_t box)
{
    const uint32_t slot = (base >> 6) & 1u;
    StreamWindow w;
    w.base = base;
    const uint32_t q = base + lane;
    w.a = ld128(blk + (q < last16 ? q : last16));
    w.b = w.a;
    stream_hash_window(w, shift);
    StreamDup d;
    stream_analyse<kSlots>(d, w, dup_scratch, lane);
    lds_words_t a = box + kDuoAna + slot * kDuoAnaDwords;
    a[lane] = (d.j1 & 0xffu) | (d.extj << 8);
    uint32_t mv = (uint32_t)d.nf;
    mv = lane == 1 ? (uint32_t)(d.nf >> 32) : mv;
    mv = lane == 2 ? (uint32_t)d.hitj : mv;
    mv = lane == 3 ? (uint32_t)(d.hitj >> 32) : mv;
    mv = lane == 4 ? (uint32_t)d.deep : mv;
    mv = lane == 5 ? (uint32_t)(d.deep >> 32) : mv;
    mv = lane == 6 ? (uint32_t)d.cx : mv;
    mv = lane == 7 ? (uint32_t)(d.cx >> 32) : mv;
    if (lane < 8u) a[64 + lane] = mv;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) box[kDuoAnaBase + slot] = base + 1u;          // after the record
    __builtin_amdgcn_wave_barrier();
    return w.a.x;
}

template <uint32_t kSlots>
__device__ __forceinline__ void duo_mate_run(const uint8_t* __restrict__ blk, uint32_t avail, uint32_t n, uint32_t shift,
                                             uint8_t* __restrict__ dst, uint32_t lane, lds_bytes_t dup_scratch, lds_words_t box)
{
    const uint32_t limit = n - kInputMargin;
    const uint32_t last16 = avail - 16u;
    uint32_t tail = 0;                                           // segments emitted (the mate is the only writer of box[kDuoSegTail])
    // first dwords of the windows prepared last, by window parity: the emission of a window usually finds its bytes here
    uint32_t keep0 = 0, keep1 = 0, keep0_base = kDuoNoWindow, keep1_base = kDuoNoWindow;
    for (;;) {
        // one round trip for everything the mate decides on
        const uint32_t c_head = box[kDuoSegHead], c_want = box[kDuoWantBase], c_a0 = box[kDuoAnaBase], c_a1 = box[kDuoAnaBase + 1],
                       c_quit = box[kDuoQuit];
        const uint32_t head = uni(c_head), want = uni(c_want), have0 = uni(c_a0), have1 = uni(c_a1), quit = uni(c_quit);
        // ---- the window the parser is (or will be) waiting for comes first ----
        uint32_t prep = kDuoNoWindow;
        if (want != kDuoNoWindow && want + kStreamRoom <= limit) {
            const uint32_t have = ((want >> 6) & 1u) ? have1 : have0;
            if (have == 0) prep = want;
        }
        if (prep == kDuoNoWindow && head != tail) {
            // ---- a segment to emit ----
            lds_words_t r = box + kDuoSeg + (tail % kDuoSegSlots) * kDuoSegDwords;
            const uint32_t packed = r[lane];                     // one round trip for the record and the emission state
            const uint32_t srow = r[64 + (lane < 10u ? lane : 0u)];
            const uint32_t st0 = box[kDuoOp], st1 = box[kDuoNextEmit];
            const unsigned long long H = duo_lanes64(srow, 0), COV = duo_lanes64(srow, 2);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)srow, 4), r_out = (uint32_t)__builtin_amdgcn_readlane((int)srow, 5),
                           flags = (uint32_t)__builtin_amdgcn_readlane((int)srow, 6), ip = (uint32_t)__builtin_amdgcn_readlane((int)srow, 7),
                           long_cand = (uint32_t)__builtin_amdgcn_readlane((int)srow, 8),
                           long_len = (uint32_t)__builtin_amdgcn_readlane((int)srow, 9);
            uint32_t op = uni(st0), next_emit = uni(st1);
            uint32_t x0;                                         // byte 0 = the window's byte at this lane (literal payloads)
            if (base == keep0_base) x0 = keep0;
            else if (base == keep1_base) x0 = keep1;
            else x0 = ld32(blk + base + lane);
            stream_emit(dst, blk, op, next_emit, base, x0, packed & 0xffffu, packed >> 16, H, COV, (flags & 1u) != 0, r_out,
                        (flags & 2u) != 0, ip, long_cand, long_len, lane);
            __builtin_amdgcn_wave_barrier();
            ++tail;
            if (lane == 0) {
                box[kDuoOp] = op;
                box[kDuoNextEmit] = next_emit;
                box[kDuoSegTail] = tail;                         // frees the slot; the state above is written before it
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (prep == kDuoNoWindow && want != kDuoNoWindow && want + 64u + kStreamRoom <= limit) {
            // ---- nothing pressing: the window after the one asked for ----
            const uint32_t have = (((want + 64u) >> 6) & 1u) ? have1 : have0;
            if (have == 0) prep = want + 64u;
        }
        if (prep != kDuoNoWindow) {
            const uint32_t x0 = duo_prepare<kSlots>(blk, prep, last16, shift, lane, dup_scratch, box);
            if ((prep >> 6) & 1u) {
                keep1 = x0;
                keep1_base = prep;
            } else {
                keep0 = x0;
                keep0_base = prep;
            }
            continue;
        }
        if (quit && head == tail) break;
        duo_nap();
    }
}

// dynamic LDS of a duo workgroup: the u16 table, the mate's two tables of analyse(), the race tables of the parser's bulk
// steps (the two run at the same time, so they cannot share bytes as they do in the one-wavefront form), the mailbox
__host__ __device__ inline uint32_t duo_lds_bytes(uint32_t block_size)
{
    return 2u * lds_table_entries(block_size) + stream_scratch_bytes(kStreamSlotsLds) + kDupSlots + ((duo_box_bytes() + 15u) & ~15u);
}

// Workgroups of two wavefronts, one block at a time each; blocks are drawn from *next_block (shared with the kernels this one
// runs beside), or handed out grid-stride when it is null.
__global__ __launch_bounds__(128) void compress_blocks_duo_kernel(const K1Batch w, uint32_t block_size, uint32_t slot_stride,
                                                                  uint32_t* next_block)
{
    const uint32_t num_blocks = w.first_block[w.count];
    HIP_DYNAMIC_SHARED(uint8_t, lds_dyn)
    uint16_t* table = reinterpret_cast<uint16_t*>(lds_dyn);
    uint8_t* mate_scratch = lds_dyn + 2u * lds_table_entries(block_size);
    uint8_t* bulk_scratch = mate_scratch + stream_scratch_bytes(kStreamSlotsLds);
    lds_words_t box = (lds_words_t)(bulk_scratch + kDupSlots);
    __shared__ uint32_t drawn_s;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = uni(threadIdx.x >> 6);                 // (wave-uniform, and the compiler must know it: the roles branch on it)
#ifndef SNAPPY_EMU
    if (next_block) __builtin_amdgcn_s_setprio(3);               // few (LDS capacity) but free of table traffic: prefer them
#endif
    uint32_t b = blockIdx.x;
    for (;;) {
        if (next_block) {
            if (threadIdx.x == 0) drawn_s = atomicAdd(next_block, 1u);
            __syncthreads();
            b = uni(drawn_s);
        }
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
        if (threadIdx.x < kDuoAna) box[threadIdx.x] = (threadIdx.x == kDuoWantBase) ? kDuoNoWindow : 0u;   // the control words
        if (wave == 1)
            for (uint32_t i = lane; i < 2u * kStreamSlotsLds; i += kWave) ((lds_words_t)mate_scratch)[i] = 0;   // analyse() keeps them zeroed
        __syncthreads();
        if (wave == 0) {
            DuoParser mate{box, n >= kInputMargin ? n - kInputMargin : 0u};
            compress_one_block_stream<LdsTable, kStreamSlotsLds>(in, start, in_len, n, slot, LdsTable{table}, lane, bytes_out,
                                                                 (lds_bytes_t)bulk_scratch, mate);
            if (lane == 0) box[kDuoQuit] = 1u;
            if (next_block && lane == 0) atomicAdd(next_block + 4, 1u);   // statistics: blocks taken by LDS-table wavefronts
        } else if (n >= kInputMargin) {
            const uint32_t ts = table_entries_for(n);
            const uint32_t shift = (uint32_t)__builtin_clz(ts) + 1;
            const uint32_t avail = (left < 0x7fffffffull) ? (uint32_t)left : 0x7fffffffu;
            duo_mate_run<kStreamSlotsLds>(in + start, avail, n, shift, slot, lane, (lds_bytes_t)mate_scratch, box);
        }
        __syncthreads();
        b += gridDim.x;
    }
}

}  // namespace snappy_hip
