// k1_tables.hpp -- hash-table variants of the global-table K1 that the product does not ship: the unfiltered tagged table and the tag-class filter.
// Ablation code: compiled only with -DSNAPPY_ABLATION (tools/build_ablation.py -> libsnappy_hip_ablation.so); the product
// library contains ONE K1 pair (bulk parse: global-table + LDS-table kernels), the two-wavefront LDS form, and one K2.
// Every form here is bit-exact with the product (tests/test_gpu_ablation.py, tests/test_emulated_kernels.py).
#pragma once

namespace snappy_hip {

// Tagged hash table of the windowed form: entry = tag << 16 | position (u32), where the tag is a 16-bit function
// of the 4 bytes at that position.  The positions stored and returned are exactly the reference's
// (snappy_compress.c:346-347, :392-397); the tag only lets a probe whose candidate has a DIFFERENT tag -- hence
// different 4 bytes, a certain miss in the reference's compare (:348, :398) -- skip fetching the candidate bytes.
// An empty slot reads "position 0" in the reference, so tables are initialised with position 0's own entry.
// The index is pinned into a VGPR so the access uses SGPR-base + VGPR-offset addressing, and every lane stores
// the same value to the same address (one write on the wire, no exec-mask save/restore).  uni() sits between the
// load and the store, so every lane has read before any lane writes.
struct TaggedGlobalTable {      // u32 entries in the global scratch: tag << 16 | position
    static constexpr bool kCollectiveStore = false;
    uint32_t* __restrict__ t;
    __device__ __forceinline__ void init(uint32_t entries, uint32_t entry_zero, uint32_t lane) const
    {
        uint4* q = reinterpret_cast<uint4*>(t);
        for (uint32_t i = lane; i < entries / 4; i += kWave) q[i] = make_uint4(entry_zero, entry_zero, entry_zero, entry_zero);
    }
    // returns the previous entry and stores `entry`
    __device__ __forceinline__ uint32_t exchange(uint32_t h, uint32_t entry, uint32_t) const
    {
        uint32_t hv = h;
        SNAPPY_PIN(hv);
        const uint32_t old = uni(t[hv]);
        t[hv] = entry;
        __builtin_amdgcn_wave_barrier();
        return old;
    }
    __device__ __forceinline__ void put(uint32_t h, uint32_t entry, uint32_t) const
    {
        uint32_t hv = h;
        SNAPPY_PIN(hv);
        t[hv] = entry;
        __builtin_amdgcn_wave_barrier();
    }
    // true when the candidate stored in `old` can be skipped without looking at its bytes
    __device__ __forceinline__ static bool certain_miss(uint32_t old, uint32_t entry) { return ((old ^ entry) >> 16) != 0; }
    // per-lane (divergent index) accessors for the look-ahead gather
    __device__ __forceinline__ uint32_t load_lane(uint32_t h, uint32_t = 0) const { return t[h]; }
    __device__ __forceinline__ void store_lane(uint32_t h, uint32_t entry) const { t[h] = entry; }
    __device__ __forceinline__ TaggedGlobalTable with_empty(uint32_t) const { return *this; }
};

// FilteredGlobalTable with two bits per slot (4 KiB of LDS per wavefront): 0 = not written in this block, 1..3 = a class of
// the 16-bit content tag of the entry the slot holds.  A probe whose own tag falls in a different class cannot match that
// entry (different tag => different 4 bytes, a certain miss in :348 / :398), so its table line is not read either; the
// caller gets an entry with the complemented tag, which certain_miss() rejects.  About 60 % of the probes of written
// slots end here.
struct ClassFilteredGlobalTable {
    static constexpr bool kCollectiveStore = false;
    uint32_t* __restrict__ t;
    lds_words_t cls;            // kMaxTableEntries / 16 words
    uint32_t empty;             // tag(position 0) << 16 | 0
    __device__ __forceinline__ static uint32_t class_of(uint32_t entry)
    {
        const uint32_t two = (entry >> 16) & 3u;
        return 1u + (two < 2u ? two : 2u);
    }
    __device__ __forceinline__ void init(uint32_t entries, uint32_t, uint32_t lane) const
    {
        for (uint32_t i = lane; i < entries / 16; i += kWave) cls[i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ uint32_t slot_class(uint32_t h) const { return (cls[h >> 4] >> ((h & 15u) * 2u)) & 3u; }
    __device__ __forceinline__ void set_class(uint32_t h, uint32_t entry) const
    {
        const uint32_t sh = (h & 15u) * 2u;
        lds_and(cls + (h >> 4), ~(3u << sh));
        lds_or(cls + (h >> 4), class_of(entry) << sh);
    }
    // what a probe carrying `probe_entry` needs to know about slot h
    __device__ __forceinline__ uint32_t load_lane(uint32_t h, uint32_t probe_entry) const
    {
        const uint32_t c = slot_class(h);
        if (c == 0) return empty;
        if (c != class_of(probe_entry)) return ~probe_entry & 0xffff0000u;     // some other tag: a certain miss
        return t[h];
    }
    __device__ __forceinline__ void store_lane(uint32_t h, uint32_t entry) const
    {
        t[h] = entry;
        set_class(h, entry);
    }
    __device__ __forceinline__ uint32_t exchange(uint32_t h, uint32_t entry, uint32_t lane) const
    {
        uint32_t hv = h;
        SNAPPY_PIN(hv);
        const uint32_t old = uni(load_lane(hv, entry));
        t[hv] = entry;
        if (lane == 0) set_class(h, entry);
        __builtin_amdgcn_wave_barrier();
        return old;
    }
    __device__ __forceinline__ void put(uint32_t h, uint32_t entry, uint32_t lane) const
    {
        uint32_t hv = h;
        SNAPPY_PIN(hv);
        t[hv] = entry;
        if (lane == 0) set_class(h, entry);
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ static bool certain_miss(uint32_t old, uint32_t entry) { return ((old ^ entry) >> 16) != 0; }
    __device__ __forceinline__ ClassFilteredGlobalTable with_empty(uint32_t e) const { return ClassFilteredGlobalTable{t, cls, e}; }
};


}  // namespace snappy_hip
