// k1_oracle_table.hpp -- CEILING EXPERIMENT for the two-pass K1 (gate (b)); ablation build only.
//
// Question: how fast is the parse (stream form, every decision and every output byte of the product) when the hash table
// costs nothing?  A host tool (tools/gate_b_records.c) runs the reference parse over the same container and records, per
// input position, one u32:
//     bits  0-15  what the window's gather would read from the table for this position (the slot's content at the moment
//                 the cursor first enters the position's 64-aligned window) -- the answer a table-free pass 2 would have
//                 to reconstruct from a parse-independent pass 1, handed over for free here;
//     bits 16-28  the duplicate analysis of the window, which IS parse-independent (what a pass 1 would deliver):
//                 16-21 nearest earlier lane of the window with the same hash, 22 there is one, 23 the 4-byte keys are
//                 equal, 24-27 bytes of the 8 behind the key that match the partner's, 28 the partner has a partner.
// OracleTable answers load_lane() from that array with ONE coalesced dword load per window and ignores every store;
// RecMate fills StreamDup from the same dword instead of running stream_analyse().  The kernel is then bit-exact with the
// product (tools/gate_b_ceiling.py compares the streams) and its time is an upper bound for ANY design that takes the
// table out of the serial kernel.  Not a product path: the records come from a CPU run of the parse itself.
#pragma once

namespace snappy_hip {

struct OracleTable {
    static constexpr bool kCollectiveStore = true;
    static constexpr bool kGathersOnce = true;   // the records hold ONE snapshot per window (its first probe): bulk_run hands over at window entries only
    const uint32_t* __restrict__ rec;      // records of this block (indexed by the position inside the block)
    __device__ __forceinline__ void init(uint32_t, uint32_t, uint32_t) const {}
    __device__ __forceinline__ static bool certain_miss(uint32_t, uint32_t) { return false; }
    __device__ __forceinline__ OracleTable with_empty(uint32_t) const { return *this; }
    __device__ __forceinline__ uint32_t load_lane(uint32_t, uint32_t mine) const
    {
        return (mine & 0xffff0000u) | (rec[mine & 0xffffu] & 0xffffu);
    }
    __device__ __forceinline__ void store_masked(unsigned long long, uint32_t, uint32_t, uint32_t) const {}
    __device__ __forceinline__ void put(uint32_t, uint32_t, uint32_t) const {}
};

struct RecMate {
    static constexpr bool kAnalysesInPlace = false;
    const uint32_t* __restrict__ rec;
    __device__ __forceinline__ void begin(uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ void drain(uint32_t&, uint32_t&, uint32_t) {}
    template <uint32_t kSlots>
    __device__ __forceinline__ void analysis(StreamDup& d, const StreamWindow& w, lds_bytes_t, uint32_t lane)
    {
        const uint32_t r = rec[w.base + lane] >> 16;
        d = StreamDup();
        d.j1 = r & 63u;
        d.nf = __ballot((r >> 6) & 1u);
        d.hitj = __ballot((r >> 7) & 1u);
        d.extj = (r >> 8) & 15u;
        d.deep = __ballot((r >> 12) & 1u);
    }
    __device__ __forceinline__ void emit(uint8_t* __restrict__ dst, const uint8_t* __restrict__ blk, uint32_t& op, uint32_t& next_emit,
                                         uint32_t base, uint32_t x0, uint32_t ent, uint32_t extv, unsigned long long H,
                                         unsigned long long COV, bool by_copy, uint32_t r_out, bool long_copy, uint32_t ip,
                                         uint32_t long_cand, uint32_t long_len, uint32_t lane)
    {
        stream_emit(dst, blk, op, next_emit, base, x0, ent, extv, H, COV, by_copy, r_out, long_copy, ip, long_cand, long_len, lane);
    }
};

// The same experiment WITH the costs a table-free pass 2 would have (kernel below, kWithCosts): the answers still come from
// the records, but every gather also does what reconstructing them would take -- read the position's parse-independent
// predecessor (`prevw`: nearest earlier position with the same hash before the window, a second host-made array, one
// coalesced u16 per lane), test its "inserted" bit in a 4 KiB LDS bitmap that the commits really maintain, and for the lanes
// whose predecessor was not inserted fetch the memoised candidate E[prevw] from a per-wavefront global array (a dependent
// random 2-byte read, awaited before the candidate bytes are asked for, as a real pass 2 would have to) -- and every window
// stores its 64 memoised candidates (one coalesced store) and pays two cross-lane permutes for resolving them.
struct OracleCostTable {
    static constexpr bool kCollectiveStore = true;
    static constexpr bool kGathersOnce = true;
    const uint32_t* __restrict__ rec;
    const uint16_t* __restrict__ prevw;
    uint16_t* __restrict__ memo;            // E[]: 64 KiB per wavefront in the scratch
    lds_words_t inserted;                   // one bit per position of the block
    __device__ __forceinline__ void init(uint32_t, uint32_t, uint32_t lane) const
    {
        for (uint32_t i = lane; i < 1024u; i += kWave) inserted[i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ static bool certain_miss(uint32_t, uint32_t) { return false; }
    __device__ __forceinline__ OracleCostTable with_empty(uint32_t) const { return *this; }
    __device__ __forceinline__ uint32_t load_lane(uint32_t, uint32_t mine) const
    {
        const uint32_t pos = mine & 0xffffu;
        const uint32_t c1 = prevw[pos];
        const bool ins = (inserted[c1 >> 5] >> (c1 & 31u)) & 1u;
        uint32_t e = 0;
        if (c1 != 0u && !ins) e = memo[c1];
#ifndef SNAPPY_EMU
        asm volatile("" ::"v"(e));          // the memoised candidate is needed HERE (the candidate bytes are fetched from it)
#endif
        return (mine & 0xffff0000u) | (rec[pos] & 0xffffu);
    }
    __device__ __forceinline__ void store_masked(unsigned long long m, uint32_t, uint32_t pos, uint32_t lane) const
    {
        const uint32_t base = (pos - lane) & 0xffffu;            // stream_store: the window's base; stream_put: garbage-free enough (lane 0 only)
        if (lane < 2u) lds_or(inserted + (((base >> 5) + lane) & 1023u), (uint32_t)(m >> (32u * lane)));
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void put(uint32_t, uint32_t, uint32_t) const {}
};

struct RecCostMate : RecMate {
    uint16_t* __restrict__ memo;
    template <uint32_t kSlots>
    __device__ __forceinline__ void analysis(StreamDup& d, const StreamWindow& w, lds_bytes_t scratch, uint32_t lane)
    {
        RecMate::analysis<kSlots>(d, w, scratch, lane);
        // once per window: resolve the memoised candidates along the in-window chains (two rounds of pointer jumping) and store them
        uint32_t v = d.j1 + w.base;
        v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((d.j1 & 63u) << 2), (int)v) + (v & 1u);
        v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((v & 63u) << 2), (int)v) + (v & 3u);
        memo[w.base + lane] = (uint16_t)v;
    }
};

// one container per launch; rec = records of the whole container (input_len + 64 entries)
template <bool kWithCosts>
__global__ __launch_bounds__(64) void compress_blocks_oracle_kernel(const K1Batch w, uint32_t block_size, uint32_t slot_stride,
                                                                    const uint32_t* __restrict__ rec, const uint16_t* __restrict__ prevw,
                                                                    uint16_t* __restrict__ memo_scratch, uint32_t* next_block)
{
    const uint32_t num_blocks = w.first_block[w.count];
    __shared__ __attribute__((aligned(16))) uint8_t dup_scratch[kDupSlots];     // the bulk form's race tables (block tails, stride > 1)
    __shared__ __attribute__((aligned(16))) uint32_t inserted_bits[kWithCosts ? 1024 : 4];   // 32768 positions
    const uint32_t lane = threadIdx.x;
    for (;;) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(next_block, 1u);
        b = uni(b);
        if (b >= num_blocks) break;
        const uint8_t* __restrict__ in = w.in[0];
        const uint64_t in_len = w.in_len[0];
        uint8_t* __restrict__ slot = w.slots[0] + (uint64_t)b * slot_stride;
        uint32_t* __restrict__ bytes_out = w.block_bytes[0] + b;
        const uint64_t start = (uint64_t)b * block_size;
        const uint64_t left = in_len - start;
        const uint32_t n = (left < block_size) ? (uint32_t)left : block_size;
        if constexpr (kWithCosts) {
            // 64 KiB of the scratch per wavefront (the launcher caps the grid at the device's wavefront slots, which is what the
            // scratch is sized for): 32768 u16 entries, one per position of a block of up to 32768 bytes -- the experiment's block size
            uint16_t* memo = memo_scratch + (size_t)blockIdx.x * 32768u;
            OracleCostTable table{rec + start, prevw + start, memo, (lds_words_t)inserted_bits};
            RecCostMate mate;
            mate.rec = rec + start;
            mate.memo = memo;
            compress_one_block_stream<OracleCostTable, kStreamSlotsGlobal>(in, start, in_len, n, slot, table, lane, bytes_out,
                                                                           (lds_bytes_t)dup_scratch, mate);
        } else {
            OracleTable table{rec + start};
            RecMate mate{rec + start};
            compress_one_block_stream<OracleTable, kStreamSlotsGlobal>(in, start, in_len, n, slot, table, lane, bytes_out,
                                                                       (lds_bytes_t)dup_scratch, mate);
        }
    }
}

}  // namespace snappy_hip
