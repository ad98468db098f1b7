// snappy_kernels.hpp -- CDNA4 (gfx950) device code for the block-framed Snappy codec.
//
// One independent Snappy block per 64-lane wavefront (= one 64-thread workgroup).
// The greedy parse is inherently serial per block, so each wavefront runs the parse as
// wave-uniform scalar control flow (state lives in SGPRs), keeps the u16 hash table in
// LDS, and uses the 64 lanes for the data-parallel parts: table clear, match extension
// (ballot + ctz), literal payload copies, multi-piece copy emission, back-reference
// replication and the coalesced write-out of decoded blocks.
//
// Every cross-lane dependency through memory is separated by a wave collective or an
// explicit __builtin_amdgcn_wave_barrier() (free on hardware: LDS/VMEM of one wave are
// issued in order), and every collective sits in wave-uniform control flow.
//
// Bit-exactness target: the reference HOST path, snappy/snappy_compress.c:284-413 and
// snappy/snappy_decompress.c:218-289 (cited per function below).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#ifdef SNAPPY_EMU
#include <stdio.h>
#include <stdlib.h>
#endif

namespace snappy_hip {

constexpr uint32_t kWave = 64;
constexpr uint32_t kMaxTableEntries = 16384;   // snappy_compress.c:16-17
constexpr uint32_t kHashMul = 0x1e35a7bdu;     // snappy_compress.c:163
constexpr uint32_t kInputMargin = 15;          // snappy_compress.c:299

// Keeps a value's computation where it is written (stops LICM from hoisting a use of an in-flight load
// out of a loop, which would drag its s_waitcnt along).  No code is generated.
#if defined(__HIP_DEVICE_COMPILE__)
#define SNAPPY_PIN(x) asm volatile("" : "+v"(x))
#else
#define SNAPPY_PIN(x) ((void)0)
#endif

constexpr uint32_t kBlockOk = 0;
constexpr uint32_t kBlockInvalid = 1;

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------

// broadcast lane 0's value; marks the value wave-uniform for the compiler (SGPR)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// unaligned little-endian loads (gfx950 runs with unaligned VMEM/DS access enabled)
__device__ __forceinline__ uint32_t ld32(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld64(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// wave-uniform load of 4 bytes at a uniform address
__device__ __forceinline__ uint32_t uld32(const uint8_t* p) { return uni(ld32(p)); }
__device__ __forceinline__ uint64_t uld64(const uint8_t* p)
{
    const uint64_t v = ld64(p);
    return (uint64_t)uni((uint32_t)v) | ((uint64_t)uni((uint32_t)(v >> 32)) << 32);
}

// hash-table sizing rule, snappy_compress.c:139-146
__device__ __forceinline__ uint32_t table_entries_for(uint32_t n)
{
    uint32_t ts = 256;
    while (ts < kMaxTableEntries && ts < n) ts <<= 1;
    return ts;
}

// ---------------------------------------------------------------------------
// element emitters (snappy_compress.c:202-272); `op` is the byte offset in the slot
// ---------------------------------------------------------------------------

// snappy_compress.c:202-225
__device__ __forceinline__ uint32_t emit_literal(uint8_t* __restrict__ dst, uint32_t op,
                                                 const uint8_t* __restrict__ src, uint32_t len, uint32_t lane)
{
    const uint32_t n = len - 1;
    uint32_t hdr;
    if (n < 60) {
        hdr = 1;
        if (lane == 0) dst[op] = (uint8_t)(n << 2);
    } else {
        const uint32_t cnt = (n < 256u) ? 1u : ((n < 65536u) ? 2u : 3u);
        hdr = 1 + cnt;
        if (lane < hdr)
            dst[op + lane] = (lane == 0) ? (uint8_t)((59 + cnt) << 2) : (uint8_t)(n >> (8 * (lane - 1)));
    }
    uint8_t* d = dst + op + hdr;
    if (len <= kWave) {
        if (lane < len) d[lane] = src[lane];
    } else {
        uint32_t i = 4 * lane;
        for (; i + 4 <= len; i += 4 * kWave) st32(d + i, ld32(src + i));
        for (; i < len; ++i) d[i] = src[i];   // at most one lane, at most 3 bytes
    }
    return op + hdr + len;
}

// snappy_compress.c:234-245, one element of 4..64 bytes
__device__ __forceinline__ uint32_t emit_copy_piece(uint8_t* __restrict__ dst, uint32_t op, uint32_t off,
                                                    uint32_t len, uint32_t lane)
{
    if (len < 12 && off < 2048) {
        if (lane < 2)
            dst[op + lane] = (lane == 0) ? (uint8_t)(1 + ((len - 4) << 2) + ((off >> 8) << 5)) : (uint8_t)(off & 0xff);
        return op + 2;
    }
    if (lane < 3)
        dst[op + lane] = (lane == 0) ? (uint8_t)(2 + ((len - 1) << 2))
                                     : ((lane == 1) ? (uint8_t)(off & 0xff) : (uint8_t)(off >> 8));
    return op + 3;
}

// snappy_compress.c:254-272: split rule ">=68 -> 64", ">64 -> 60", rest
__device__ __forceinline__ uint32_t emit_copy(uint8_t* __restrict__ dst, uint32_t op, uint32_t off, uint32_t len,
                                              uint32_t lane)
{
    if (len > 64) {
        const uint32_t n64 = (len >= 68) ? ((len - 68) / 64 + 1) : 0;
        len -= 64 * n64;                      // now 4..67
        const uint32_t has60 = (len > 64) ? 1u : 0u;
        if (has60) len -= 60;                 // now 5..7
        const uint32_t nfull = n64 + has60;   // every one of these is a 3-byte COPY_2
        for (uint32_t k = lane; k < nfull; k += kWave) {
            const uint32_t plen = (k < n64) ? 64u : 60u;
            uint8_t* p = dst + op + 3 * k;
            p[0] = (uint8_t)(2 + ((plen - 1) << 2));
            p[1] = (uint8_t)(off & 0xff);
            p[2] = (uint8_t)(off >> 8);
        }
        op += 3 * nfull;
    }
    return emit_copy_piece(dst, op, off, len, lane);
}

// snappy_compress.c:176-193: number of equal bytes of blk[a..] and blk[b..], b bounded by n.
// 64 lanes x 4 bytes per round, first mismatch by ballot + ctz.
__device__ __forceinline__ uint32_t match_extend(const uint8_t* __restrict__ blk, uint32_t a, uint32_t b, uint32_t n,
                                                 uint32_t lane)
{
    uint32_t m = 0;
    for (;;) {
        const uint32_t pb = b + m + 4 * lane;
        const uint32_t pa = a + m + 4 * lane;
        uint32_t eq;   // equal leading bytes this lane can vouch for (0..4)
        if (pb + 4 <= n) {
            const uint32_t x = ld32(blk + pa) ^ ld32(blk + pb);
            eq = x ? ((uint32_t)__builtin_ctz(x) >> 3) : 4u;
        } else {
            eq = 0;
            for (uint32_t k = 0; pb + k < n; ++k) {   // <= 3 bytes, tail lanes only
                if (blk[pa + k] != blk[pb + k]) break;
                ++eq;
            }
        }
        const unsigned long long stop = __ballot(eq != 4);
        if (stop == 0) {
            m += 4 * kWave;
            continue;
        }
        const uint32_t first = (uint32_t)__builtin_ctzll(stop);
        return m + 4 * first + (uint32_t)__builtin_amdgcn_readlane((int)eq, (int)first);
    }
}

// ---------------------------------------------------------------------------
// K1, windowed form: one Snappy block per wavefront, the greedy parse of snappy_compress.c:284-413 as
// wave-uniform control flow, structured around what the PMC profiles showed to limit it: instruction
// issue on the scalar pipeline and serialized memory round trips per probe.
//
//  * Cursor side: a sliding register window.  Lane l holds x = le32(block + base + l) and its hash for
//    the 64 positions of the current 64-byte granule, plus a prefetch of the next granule.  The bytes
//    and the hash of the position being probed are one v_readlane each -- no load, no address math,
//    no multiply on the scalar unit.
//  * Candidate side: one 16-byte scalar load returns the 4 bytes for the hit test AND the next 8 bytes
//    for the match extension, so a match of up to 12 bytes costs no further round trip; longer matches
//    continue in the 64-lane extender.
//  * Hash tables: templated.  TaggedGlobalTable = one u32[16384] per resident wavefront in a global scratch,
//    so occupancy is bounded by registers (32 waves/CU), not LDS; LdsTable = the reference's u16[16384] in LDS
//    (5 workgroups/CU).  The default launch runs BOTH kernels concurrently on one container: the LDS-table
//    waves (lower table latency, no table traffic) take 4 of the 32 wave slots per CU, the global-table waves
//    the rest; every wavefront draws its next block from one atomic counter (*next_block zeroed per launch).
// Bit-exactness: identical decisions to snappy_compress.c:284-413; only where bytes are read from differs.
// ---------------------------------------------------------------------------
struct CursorWindow {
    const uint8_t* __restrict__ blk;   // block start
    uint32_t avail;                    // readable bytes from blk (clamped to 2^31)
    uint32_t shift;                    // hash shift of this block
    uint32_t base;                     // window base (multiple of 64)
    uint32_t x0, h0, e0;               // per lane: le32(blk + base + l), its hash, its content tag << 16
    uint32_t x1;                       // per lane: le32(blk + base + 64 + l) -- prefetch, possibly in flight

    // hash (snappy_compress.c:161-166) = top bits of x * kHashMul; the tag is the 16 bits right below them, so two
    // different 4-byte values that share a table slot also share a tag only once in 65536 times
    __device__ __forceinline__ void rehash()
    {
        const uint32_t prod = x0 * kHashMul;
        h0 = prod >> shift;
        e0 = (prod << (32 - shift)) & 0xffff0000u;
    }

    __device__ __forceinline__ uint32_t load_at(uint32_t pos, uint32_t lane) const
    {
        const uint32_t q = pos + lane;
        const uint32_t last = avail - 4;                         // avail >= 15 whenever the window is used
        return ld32(blk + ((q < last) ? q : last));              // lanes past the end read a clamped address
    }
    __device__ __forceinline__ void reset(uint32_t pos, uint32_t lane)
    {
        base = pos & ~63u;
        x0 = load_at(base, lane);
        rehash();
        __builtin_amdgcn_sched_barrier(0);
        x1 = load_at(base + 64, lane);
    }
    // make `pos` fall inside [base, base + 64)
    __device__ __forceinline__ bool ensure(uint32_t pos, uint32_t lane)   // true when the window moved
    {
        if (pos < base + 64) return false;
        if (pos < base + 128) {
            base += 64;
            x0 = x1;
            rehash();
            __builtin_amdgcn_sched_barrier(0);
            x1 = load_at(base + 64, lane);
        } else {
            reset(pos, lane);
        }
        return true;
    }
    __device__ __forceinline__ uint32_t bytes_at(uint32_t pos) const   // pos in [base, base+64)
    {
        return (uint32_t)__builtin_amdgcn_readlane((int)x0, (int)(pos - base));
    }
    __device__ __forceinline__ uint32_t hash_at(uint32_t pos) const
    {
        return (uint32_t)__builtin_amdgcn_readlane((int)h0, (int)(pos - base));
    }
    // table entry for inserting `pos`: tag << 16 | pos
    __device__ __forceinline__ uint32_t entry_at(uint32_t pos) const
    {
        return (uint32_t)__builtin_amdgcn_readlane((int)e0, (int)(pos - base)) | pos;
    }
    // le32 at pos for pos in [base, base+128): second granule forces the wait on the prefetch
    __device__ __forceinline__ uint32_t bytes_near(uint32_t pos) const
    {
        const uint32_t rel = pos - base;
        if (rel < 64) return (uint32_t)__builtin_amdgcn_readlane((int)x0, (int)rel);
        uint32_t v = x1;
        SNAPPY_PIN(v);
        return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(rel - 64));
    }
};

// LDS pointers keep their address space (ds_* instead of flat_*); the CPU emulator sees plain pointers.
#ifdef SNAPPY_EMU
typedef volatile uint8_t* lds_bytes_t;
typedef volatile uint32_t* lds_words_t;
__device__ __forceinline__ void lds_or(lds_words_t p, uint32_t v) { *p = *p | v; }
__device__ __forceinline__ void lds_and(lds_words_t p, uint32_t v) { *p = *p & v; }
#else
typedef volatile __attribute__((address_space(3))) uint8_t* lds_bytes_t;
typedef volatile __attribute__((address_space(3))) uint32_t* lds_words_t;
__device__ __forceinline__ void lds_or(lds_words_t p, uint32_t v)
{
    __hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void lds_and(lds_words_t p, uint32_t v)
{
    __hip_atomic_fetch_and((__attribute__((address_space(3))) uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
#endif

// unaligned dword access to LDS (gfx950 runs with unaligned DS access enabled, like its VMEM)
#ifdef SNAPPY_EMU
__device__ __forceinline__ uint32_t lds_ld32u(lds_bytes_t p)
{
    uint32_t v;
    __builtin_memcpy(&v, (const void*)p, 4);
    return v;
}
__device__ __forceinline__ void lds_st32u(lds_bytes_t p, uint32_t v) { __builtin_memcpy((void*)p, &v, 4); }
#else
typedef uint32_t __attribute__((aligned(1))) lds_u32_unaligned_t;
__device__ __forceinline__ uint32_t lds_ld32u(lds_bytes_t p)
{
    return *reinterpret_cast<volatile __attribute__((address_space(3))) lds_u32_unaligned_t*>(p);
}
__device__ __forceinline__ void lds_st32u(lds_bytes_t p, uint32_t v)
{
    *reinterpret_cast<volatile __attribute__((address_space(3))) lds_u32_unaligned_t*>(p) = v;
}
#endif

// Emulator only: aborts when two lanes of one (emulated) store instruction write the same table entry -- on the GPU it is
// not defined which of them lands last.  All lanes call, between two collectives.
#ifdef SNAPPY_EMU
inline uint32_t g_emu_store_index[16][kWave];                    // per emulated workgroup (grids of the emulator are tiny)
__device__ __forceinline__ void emu_check_distinct_stores(bool stores, uint32_t index, uint32_t lane)
{
    uint32_t* mine = g_emu_store_index[blockIdx.x & 15u];
    mine[lane] = stores ? index : 0xffffffffu;
    __builtin_amdgcn_wave_barrier();
    if (stores)
        for (uint32_t l = 0; l < kWave; ++l)
            if (l != lane && mine[l] == index) {
                fprintf(stderr, "emulator: lanes %u and %u store to table entry %u in one instruction\n", lane, l, index);
                abort();
            }
    __builtin_amdgcn_wave_barrier();
}
#else
__device__ __forceinline__ void emu_check_distinct_stores(bool, uint32_t, uint32_t) {}
#endif

// TaggedGlobalTable behind a one-bit-per-slot "written in this block" filter in LDS (2 KiB per wavefront).  Early in a
// block most slots still hold the initial entry (candidate position 0, :145 + :346), and the speculative gathers of the
// look-ahead forms read 64 slots per 64 input bytes -- one random 64-byte HBM line per input byte, which is what bounds
// those forms (~58 G random lines/s on MI355X).  With the filter a slot that was not written since the block started is
// answered from a register (the initial entry), its line is never fetched, and the table needs no per-block
// initialisation at all: whatever an earlier block left in the scratch is unreachable until this block overwrites it.
struct FilteredGlobalTable {
    static constexpr bool kCollectiveStore = false;
    uint32_t* __restrict__ t;
    lds_words_t written;        // kMaxTableEntries / 32 words
    uint32_t empty;             // tag(position 0) << 16 | 0
    __device__ __forceinline__ void init(uint32_t entries, uint32_t, uint32_t lane) const
    {
        for (uint32_t i = lane; i < entries / 32; i += kWave) written[i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ bool is_written(uint32_t h) const { return (written[h >> 5] >> (h & 31u)) & 1u; }
    __device__ __forceinline__ uint32_t exchange(uint32_t h, uint32_t entry, uint32_t lane) const
    {
        uint32_t hv = h;
        SNAPPY_PIN(hv);
        uint32_t old = empty;
        if (is_written(hv)) old = t[hv];
        old = uni(old);
        t[hv] = entry;
        if (lane == 0) lds_or(written + (h >> 5), 1u << (h & 31u));
        __builtin_amdgcn_wave_barrier();
        return old;
    }
    __device__ __forceinline__ void put(uint32_t h, uint32_t entry, uint32_t lane) const
    {
        uint32_t hv = h;
        SNAPPY_PIN(hv);
        t[hv] = entry;
        if (lane == 0) lds_or(written + (h >> 5), 1u << (h & 31u));
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ static bool certain_miss(uint32_t old, uint32_t entry) { return ((old ^ entry) >> 16) != 0; }
    __device__ __forceinline__ uint32_t load_lane(uint32_t h, uint32_t = 0) const { return is_written(h) ? t[h] : empty; }
    __device__ __forceinline__ void store_lane(uint32_t h, uint32_t entry) const
    {
        t[h] = entry;      // (write-through `sc1` and non-temporal stores measured the same: DESIGN 3.1)
        lds_or(written + (h >> 5), 1u << (h & 31u));
    }
    __device__ __forceinline__ FilteredGlobalTable with_empty(uint32_t e) const { return FilteredGlobalTable{t, written, e}; }
};

// The global table behind a write-back cache of slots in LDS (round 3).  The global-table kernel runs at the HBM's
// random-access rate: per 64-byte window 35 slot reads and 20 table stores, each a DRAM burst of its own
// (profiles/r03_k1_global_table_bound.txt).  Table stores have strong temporal locality -- a 4-gram that was just inserted is
// inserted or probed again soon -- so a direct-mapped cache of kSlots recent stores takes most of them
// (tools/slot_cache_sim.c on the benchmark data, 2,048 slots: 35.1 -> 10.2 slot reads and 19.9 -> 9.3 stores per window
// reach the global table).  Only stores allocate (a gather's reads are mostly speculation); a slot displaced from the cache
// is written to the global table then.  The slot's value is: its cache word if the cache holds the slot, else the global
// entry if the slot was ever written (`written` bit), else position 0 (:145).  Global entries are positions only (u16):
// a content tag would have to be recomputed at eviction, and candidates are recent input (L2 hits).
//   cache word = (slot + 1) << 16 | position; 0 = free
template <uint32_t kSlots>
struct CachedGlobalTable {
    static constexpr bool kCollectiveStore = true;
    uint16_t* __restrict__ t;   // kMaxTableEntries positions per wavefront
    lds_words_t written;        // kMaxTableEntries / 32 words
    lds_words_t cache;          // kSlots words
    __device__ __forceinline__ void init(uint32_t entries, uint32_t, uint32_t lane) const
    {
        for (uint32_t i = lane; i < entries / 32; i += kWave) written[i] = 0;
        for (uint32_t i = lane; i < kSlots; i += kWave) cache[i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ bool is_written(uint32_t h) const { return (written[h >> 5] >> (h & 31u)) & 1u; }
    __device__ __forceinline__ static bool certain_miss(uint32_t, uint32_t) { return false; }
    __device__ __forceinline__ CachedGlobalTable with_empty(uint32_t) const { return *this; }
    // the slot's position under the probing lane's own tag (every written slot's candidate is compared)
    __device__ __forceinline__ uint32_t load_lane(uint32_t h, uint32_t mine) const
    {
        const uint32_t w = cache[h & (kSlots - 1u)];
        const bool wr = is_written(h);
        uint32_t pos = w & 0xffffu;
        if ((w >> 16) != h + 1u) pos = wr ? (uint32_t)t[h] : 0u;
        return (mine & 0xffff0000u) | pos;
    }
    // All lanes call; the lanes in `m` (each its own slot) insert `pos`.  Lanes of one call that share a cache word: one
    // of them gets it (read back, not assumed), the others write through.
    // Round 4 tried ONE LDS atomic exchange per inserting lane instead (the word that comes back is displaced to the global
    // table): 10 instructions per call instead of ~25, bit-exact once the displaced words of earlier calls and those of the
    // call's own lanes went out in two store instructions -- in one, the old and the new position of a slot can meet at one
    // address, and which lands last is not defined (three lanes in a cache word, once per ~1,300 blocks of the benchmark data:
    // found on the GPU, now caught by emu_check_distinct_stores).  No faster on any workload
    // (profiles/r04_store_xchg_ab.txt), so the protocol that has three rounds of soak behind it stays.
    __device__ __forceinline__ void store_masked(unsigned long long m, uint32_t h, uint32_t pos, uint32_t lane) const
    {
        const bool s = __builtin_amdgcn_inverse_ballot_w64(m);
        const uint32_t idx = h & (kSlots - 1u), mine = ((h + 1u) << 16) | (pos & 0xffffu);
        const uint32_t old = cache[idx];                         // (every lane reads: no exec-mask region for a load)
        __builtin_amdgcn_wave_barrier();
        if (s) {
            cache[idx] = mine;
            lds_or(written + (h >> 5), 1u << (h & 31u));
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t now = cache[idx];
        const bool lost = s && now != mine;
        // the displaced slot goes out first, then the lanes that lost (one of them may be the displaced slot's new value).
        // Relied upon: two global_store instructions of ONE wavefront to the same address are performed in issue order, also
        // when different lanes issue them (vector memory operations of a wavefront complete in order -- global_*, not flat_*:
        // MI355X_MICROARCH.md; the same guarantee K2's back-references use).  tests/test_abi_symbols.py fails on a flat_*
        // instruction in any K1 kernel; the wave barrier below only keeps the compiler from reordering the two stores.
        // Within ONE store instruction no two lanes may share an address (not defined which would land last): the displaced
        // words belong to distinct cache words, hence distinct slots, and the lanes of `m` have slots of their own; the
        // emulator checks it (emu_check_distinct_stores).
        const bool evicts = s && !lost && old != 0u && (old >> 16) != h + 1u;
        if (evicts) t[(old >> 16) - 1u] = (uint16_t)old;
        emu_check_distinct_stores(evicts, (old >> 16) - 1u, lane);
        __builtin_amdgcn_wave_barrier();
        if (__ballot(lost)) {                                    // rare: two stores of one call met in a cache word
            if (lost) t[h] = (uint16_t)pos;
            emu_check_distinct_stores(lost, h, lane);
        }
        __builtin_amdgcn_wave_barrier();
    }
};

struct LdsTable {               // the reference's own layout: u16 positions, in LDS (no room for tags: 32 KiB per block)
    static constexpr bool kCollectiveStore = false;
    uint16_t* t;
    __device__ __forceinline__ void init(uint32_t entries, uint32_t, uint32_t lane) const
    {
        uint4* q = reinterpret_cast<uint4*>(t);
        for (uint32_t i = lane; i < entries / 8; i += kWave) q[i] = make_uint4(0, 0, 0, 0);
    }
    __device__ __forceinline__ uint32_t exchange(uint32_t h, uint32_t entry, uint32_t lane) const
    {
        const uint32_t old = uni((uint32_t)t[h]);
        if (lane == 0) t[h] = (uint16_t)entry;       // one lane: 64 same-address LDS writes would serialise
        __builtin_amdgcn_wave_barrier();
        return old;
    }
    __device__ __forceinline__ void put(uint32_t h, uint32_t entry, uint32_t lane) const
    {
        if (lane == 0) t[h] = (uint16_t)entry;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ static bool certain_miss(uint32_t, uint32_t) { return false; }
    __device__ __forceinline__ uint32_t load_lane(uint32_t h, uint32_t = 0) const { return t[h]; }
    __device__ __forceinline__ void store_lane(uint32_t h, uint32_t entry) const { t[h] = (uint16_t)entry; }
    __device__ __forceinline__ LdsTable with_empty(uint32_t) const { return *this; }
};

// Emitters of the windowed form.  Element headers are packed into one dword and stored by a single lane
// (the 1-2 bytes past the header are overwritten by whatever is emitted next; every slot has >= 32 bytes of
// slack, snappy_compress.c:55-60), and literal payloads are stored straight from the cursor window registers
// (byte 0 of lane l's x0 IS block byte base + l) -- no load, so no round trip on the emission path.

// snappy_compress.c:234-245 (len 4..64) / :254-272
__device__ __forceinline__ uint32_t emit_copy_packed(uint8_t* __restrict__ dst, uint32_t op, uint32_t off, uint32_t len,
                                                     uint32_t lane)
{
    if (len > 64) return emit_copy(dst, op, off, len, lane);     // multi-piece copies: generic path
    uint32_t word, bytes;
    if (len < 12 && off < 2048) {
        word = (1 + ((len - 4) << 2) + ((off >> 8) << 5)) | ((off & 0xff) << 8);
        bytes = 2;
    } else {
        word = (2 + ((len - 1) << 2)) | (off << 8);
        bytes = 3;
    }
    if (lane == 0) st32(dst + op, word);
    return op + bytes;
}

// snappy_compress.c:202-225 with the payload taken from the window when [from, from+len) lies inside it
__device__ __forceinline__ uint32_t emit_literal_windowed(uint8_t* __restrict__ dst, uint32_t op,
                                                          const uint8_t* __restrict__ blk, uint32_t from, uint32_t len,
                                                          uint32_t win_base, uint32_t win_x0, uint32_t lane)
{
    if (from < win_base || from + len > win_base + kWave) return emit_literal(dst, op, blk + from, len, lane);
    const uint32_t n = len - 1;                                  // len <= 64 here, so n < 64
    uint32_t hdr;
    if (n < 60) {
        hdr = 1;
        if (lane == 0) dst[op] = (uint8_t)(n << 2);
    } else {
        hdr = 2;
        if (lane == 0) {
            dst[op] = (uint8_t)(60 << 2);
            dst[op + 1] = (uint8_t)n;
        }
    }
    const uint32_t rel = from - win_base;
    if (lane >= rel && lane < rel + len) dst[op + hdr + lane - rel] = (uint8_t)win_x0;
    return op + hdr + len;
}

// floor(x / d) for x < 64, 1 <= d < 64:  (x * kRecip16[d]) >> 16  with kRecip16[d] = 65536/d + 1
__constant__ uint32_t kRecip16[64] = {
        0, 65537, 32769, 21846, 16385, 13108, 10923,  9363,  8193,  7282,  6554,  5958,  5462,  5042,  4682,  4370,
     4097,  3856,  3641,  3450,  3277,  3121,  2979,  2850,  2731,  2622,  2521,  2428,  2341,  2260,  2185,  2115,
     2049,  1986,  1928,  1873,  1821,  1772,  1725,  1681,  1639,  1599,  1561,  1525,  1490,  1457,  1425,  1395,
     1366,  1338,  1311,  1286,  1261,  1237,  1214,  1192,  1171,  1150,  1130,  1111,  1093,  1075,  1058,  1041};

// ---------------------------------------------------------------------------
// K1, masked form.  The PMC profile of the look-ahead form above shows it instruction-issue bound (SQ_WAIT_INST_ANY +
// SQ_ACTIVE_INST_ANY > 50 % of wave cycles, ~80 instructions per probe).  This form keeps the same speculative gather but
// resolves the probes of a window with 64-bit lane masks instead of one scalar round per probe:
//
//  * gather(r): lanes r .. r+kChunk-1 read their table slot, the tag-matching ones their 12 candidate bytes, and every
//    lane computes -- in parallel -- whether a probe at its position WOULD hit (HIT mask, :348 / :398) and how many of the
//    next 8 bytes would match (`extv`, the head of find_match_length, :176-193).
//  * The speculation is valid for a lane as long as no insert made after the gather went to its table slot.  Inserts
//    between a window's gather and the probe of one of its lanes are all positions of that same window, so the lanes at
//    risk are exactly those that share a slot with another lane of the window.  dup_slot_lanes() finds a superset of them
//    (DUP mask) with four LDS byte accesses; DUP lanes never use the cache: they take the serial exchange below.
//  * A scan run (:336-348, stride 1) is then: first set bit of (HIT | DUP) at or above the cursor.  The misses in front of
//    it are committed with ONE masked vector store (each lane writes its own entry to its own slot -- distinct slots, so
//    the order the reference made them in does not matter), and a resolved hit costs two v_readlane.
// Decisions, table contents after every step, and output bytes are those of snappy_compress.c:284-413.
// ---------------------------------------------------------------------------
constexpr uint32_t kDupSlots = 1024;   // bytes of LDS per wavefront for the duplicate-slot test (two tables of 512)

// index of the lowest set bit (0..31); all ones when x == 0 (v_ffbl_b32's own convention)
__device__ __forceinline__ uint32_t first_bit(uint32_t x) { return (uint32_t)__builtin_ffs((int)x) - 1u; }
__device__ __forceinline__ uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

__device__ __forceinline__ uint32_t ctz64_or(unsigned long long x, uint32_t if_zero)
{
    return x ? (uint32_t)__builtin_ctzll(x) : if_zero;
}
// lanes [lo, lo + cnt), 1 <= cnt, lo + cnt <= 64
__device__ __forceinline__ unsigned long long lane_range(uint32_t lo, uint32_t cnt)
{
    return (~0ull >> (64u - cnt)) << lo;
}

// Superset of the lanes whose table slot `h` is shared with another lane.  Two byte tables indexed by overlapping halves of the
// hash (bits 0-8 and 5-13): lanes race for a byte per slot; every loser, and every winner that a loser then marks, "has
// company" in that table.  Lanes with the same hash have company in both tables; two lanes with different hashes cannot
// collide in both (bits 0-8 and 5-13 equal means all 14 equal), so what is reported beyond the true sharers is only the
// rare lane that collides with one neighbour in the first table and with another in the second.
__device__ __forceinline__ unsigned long long dup_slot_lanes(lds_bytes_t scratch, uint32_t h, uint32_t lane)
{
    const uint32_t sa = h & (kDupSlots / 2 - 1);
    const uint32_t sb = kDupSlots / 2 + ((h >> 5) & (kDupSlots / 2 - 1));
    scratch[sa] = (uint8_t)lane;
    scratch[sb] = (uint8_t)lane;
    __builtin_amdgcn_wave_barrier();
    const uint32_t wa = scratch[sa];
    const uint32_t wb = scratch[sb];
    __builtin_amdgcn_wave_barrier();
    if (wa != lane) scratch[sa] = 0xff;              // lane ids are < 64
    if (wb != lane) scratch[sb] = 0xff;
    __builtin_amdgcn_wave_barrier();
    const uint32_t ma = scratch[sa];
    const uint32_t mb = scratch[sb];
    __builtin_amdgcn_wave_barrier();
    return __ballot(ma == 0xff && mb == 0xff);
}

template <class Table, uint32_t kChunk>
struct MaskedWindowState {
    uint32_t ent = 0;               // per lane: table slot content at gather time
    uint32_t extv = 0;              // per lane: matching bytes among the 8 after the 4-byte key (0..8), for HIT lanes
    unsigned long long hit = 0;     // wave-uniform: lanes whose probe would hit
    unsigned long long dup = 0;     // wave-uniform: lanes that must not use the cache
    unsigned long long longm = 0;   // wave-uniform: HIT lanes whose extv is saturated (the match goes on)
    unsigned long long deepm = 0;   // wave-uniform: lanes whose extv covers 24 bytes (saturates at 24) instead of 8
    unsigned long long inserted = 0;// wave-uniform: window lanes whose position has been inserted (bulk form)
    uint32_t xa = 0, xb = 0;        // per lane: le32 at position + 4 / + 8
    uint32_t cov_end = 0;           // lanes in [gather start, cov_end) are resolved
    bool dup_valid = false;

    __device__ __forceinline__ void invalidate()
    {
        cov_end = 0;
        hit = 0;
        dup_valid = false;
        inserted = 0;
    }

    // le32(block + window base + lane + d), d in {4, 8}: from the current or the prefetched granule
    __device__ __forceinline__ static uint32_t bytes_ahead(const CursorWindow& win, uint32_t lane, uint32_t d)
    {
        const uint32_t src = (lane + d) & 63u;
        const uint32_t a = (uint32_t)__shfl((int)win.x0, (int)src);
        const uint32_t b = (uint32_t)__shfl((int)win.x1, (int)src);
        return (lane + d < kWave) ? a : b;
    }

    // kWithDup: also read the slots of DUP lanes (their values hold only while no same-hash lane of the window is inserted)
    // kDeep: compare 28 candidate bytes instead of 12 (where the block has that many), so that matches of up to 27 bytes
    // are fully sized by the gather
    template <bool kWithDup = false, bool kDeep = false>
    __device__ __forceinline__ void gather(const Table& table, const CursorWindow& win, lds_bytes_t scratch, uint32_t r,
                                           uint32_t span, uint32_t lane, uint32_t block_len = 0)
    {
        const uint32_t e = (r + span < kWave) ? r + span : kWave;
        if (!kWithDup && !dup_valid) {                 // the gather mask needs DUP first
            dup = dup_slot_lanes(scratch, win.h0, lane);
            dup_valid = true;
        }
        const unsigned long long gm = kWithDup ? lane_range(r, e - r) : (lane_range(r, e - r) & ~dup);
        const bool g = __builtin_amdgcn_inverse_ballot_w64(gm);
        const uint32_t mine_l = win.e0 | (win.base + lane);
        if (g) ent = table.load_lane(win.h0, mine_l);
        // The LDS duplicate test (three dependent LDS round trips) runs underneath the longest loads of the gather: the table
        // loads when the table is in global memory, the candidate loads when it is in LDS (its entries arrive at once).
        constexpr bool kDupUnderCandidates = std::is_same<Table, LdsTable>::value;
        if (kWithDup && !dup_valid && !kDupUnderCandidates) {
            __builtin_amdgcn_sched_barrier(0);
            dup = dup_slot_lanes(scratch, win.h0, lane);
            dup_valid = true;
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool worth = g && !Table::certain_miss(ent, mine_l);
        uint32_t k0 = 0, k1 = 0, k2 = 0;
        const bool deep = kDeep && worth && (win.base + lane + 28u <= block_len);   // candidate < position, so it has 28 too
        uint32_t k3 = 0, k4 = 0, k5 = 0, k6 = 0, o3 = 1, o4 = 1, o5 = 1, o6 = 1;
        if (worth) {                                  // every stored position p has p + 16 <= block length
            const uint8_t* __restrict__ c = win.blk + (ent & 0xffffu);
            k0 = ld32(c);
            k1 = ld32(c + 4);
            k2 = ld32(c + 8);
            if (deep) {
                k3 = ld32(c + 12);
                k4 = ld32(c + 16);
                k5 = ld32(c + 20);
                k6 = ld32(c + 24);
                const uint8_t* __restrict__ o = win.blk + win.base + lane;
                o3 = ld32(o + 12);
                o4 = ld32(o + 16);
                o5 = ld32(o + 20);
                o6 = ld32(o + 24);
            }
        }
        if (kWithDup && !dup_valid && kDupUnderCandidates) {
            __builtin_amdgcn_sched_barrier(0);
            dup = dup_slot_lanes(scratch, win.h0, lane);
            dup_valid = true;
            __builtin_amdgcn_sched_barrier(0);
        }
        xa = bytes_ahead(win, lane, 4);
        xb = bytes_ahead(win, lane, 8);
        // Matching bytes behind the key = the first differing bit of the compared words, found without a branch (the ternary
        // cascade compiles to five nested exec-mask regions).  first_bit: 0..31, all ones for a zero word, which the ORs keep as
        // "none" and the minimum ignores.  A lane that is not deep has k3..k6 = 0 and o3..o6 = 1: its second word group differs
        // at bit 0, which caps it at 8 bytes.
        const uint32_t t1 = min_u32(first_bit(k1 ^ xa), first_bit(k2 ^ xb) | 32u);
        uint32_t bits = min_u32(t1, 64u);
        if (kDeep) {
            const uint32_t t2 = min_u32(first_bit(k3 ^ o3), first_bit(k4 ^ o4) | 32u) | 64u;
            const uint32_t t3 = min_u32(first_bit(k5 ^ o5), first_bit(k6 ^ o6) | 32u) | 128u;
            bits = min_u32(min_u32(t1, t2), min_u32(t3, 192u));
        }
        extv = bits >> 3;
        // lane masks from single compares, combined with scalar logic (a ballot of a compound bool costs a v_cndmask + v_cmp more)
        const unsigned long long worthm = __ballot(worth);
        deepm = kDeep ? (worthm & __ballot(win.base + lane + 28u <= block_len)) : 0ull;
        hit = worthm & __ballot(k0 == win.x0);         // lanes below r are behind the cursor, lanes >= e not covered
        longm = hit & ((deepm & __ballot(extv == 24u)) | (~deepm & __ballot(extv == 8u)));   // hits whose match goes on past the compared bytes
        cov_end = e;
    }

    // insert the positions of the lanes in `m` (each its own slot)
    __device__ __forceinline__ static void commit(const Table& table, const CursorWindow& win, unsigned long long m, uint32_t lane)
    {
        if constexpr (Table::kCollectiveStore) {
            table.store_masked(m, win.h0, win.base + lane, lane);
        } else {
            if (__builtin_amdgcn_inverse_ballot_w64(m)) table.store_lane(win.h0, win.e0 | (win.base + lane));
            __builtin_amdgcn_wave_barrier();
        }
    }
};

// ---------------------------------------------------------------------------
// K1, bulk form: the masked form with the per-match work moved off the scalar chain.  Within one window, for as long as
// the parse only meets resolved lanes (not DUP), short matches (< 12 bytes, length known from `extv`) and stride-1
// scanning, the walk is pure mask arithmetic -- per match: first set bit of HIT above the cursor, one v_readlane for the
// match length, four mask updates.  What the reference does per step is then done once per such SEGMENT by all lanes:
//   * table inserts (:346-347, :391-392, :397): one masked vector store for every probed lane and every "ip - 1" lane;
//   * emission (:355, :202-245): each lane derives its own output offset from the segment's masks with v_mbcnt --
//       P(l) = op + #literal bytes + 2 * #copies + #3-byte copies + #literal headers   (all counted below lane l)
//     literal lanes store their byte, run-start lanes the literal header, hit lanes the 2- or 3-byte copy element.
// Anything else (DUP lanes, matches of 12+ bytes, window edges) ends the segment and takes a single step; a DUP lane is
// resolved there from the window registers (the latest inserted lane with the same hash IS the table slot's content).  Same decisions, same table contents at every read, same bytes as snappy_compress.c:284-413.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m, uint32_t add)
{
    return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, add));
}
// lanes [0, n), 1 <= n <= 64
__device__ __forceinline__ unsigned long long lanes_below(uint32_t n) { return ~0ull >> (64u - n); }

// Emission of one segment (:355, :202-245) by the lanes themselves.  H = lanes whose probe hit (their copy: candidate in
// `ent`, length 4 + extv), COV = lanes covered by those copies, r = where the cursor stands afterwards relative to `base`
// (by_copy: right behind the last copy, possibly beyond the window; otherwise the literal after the last copy stays
// pending).  Each lane derives its own output offset from the masks with v_mbcnt:
//     P(l) = op + #literal bytes + 2 * #copies + #3-byte copies + #literal headers   (all counted below lane l)
// literal lanes store their byte (byte 0 of x0 IS block byte base + l), run-start lanes the literal header, hit lanes the
// 2- or 3-byte copy element.
__device__ __forceinline__ void emit_segment(uint8_t* __restrict__ dst, const uint8_t* __restrict__ blk, uint32_t& op, uint32_t& next_emit,
                                             uint32_t base, uint32_t x0, uint32_t ent, uint32_t extv, unsigned long long H,
                                             unsigned long long COV, bool by_copy, uint32_t r, uint32_t lane)
{
    const uint32_t r_end = r < kWave ? r : kWave;
    const uint32_t first_hit = (uint32_t)__builtin_ctzll(H);
    const uint32_t last_end = by_copy ? r_end : 64u - (uint32_t)__builtin_clzll(COV);
    uint32_t s0 = first_hit;
    const uint32_t p0 = base + first_hit;
    if (next_emit >= base && p0 - next_emit <= 60u) {
        s0 = next_emit - base;                       // the first run is inside the window too
    } else if (p0 > next_emit) {                     // it started in an earlier window (or is 61+ bytes)
        op = emit_literal_windowed(dst, op, blk, next_emit, p0 - next_emit, base, x0, lane);
    }
    const unsigned long long LIT = ((~0ull << s0) & lanes_below(last_end)) & ~COV;
    const unsigned long long LS = LIT & ~(LIT << 1);            // first lane of each literal run
    const uint32_t off = base + lane - (ent & 0xffffu);         // meaningful in H lanes
    const uint32_t len = 4u + extv;
    const bool is_hit = __builtin_amdgcn_inverse_ballot_w64(H);
    const bool three = off >= 2048u || len >= 12u;                 // :234-245
    const unsigned long long H3 = __ballot(is_hit && three);
    uint32_t P = mbcnt64(H, 0);
    P = mbcnt64(LIT, op + 2u * P);
    P = mbcnt64(H3, P);
    P = mbcnt64(LS, P);
    const bool is_ls = __builtin_amdgcn_inverse_ballot_w64(LS);
    if (__builtin_amdgcn_inverse_ballot_w64(LIT)) dst[P + (is_ls ? 1u : 0u)] = (uint8_t)x0;
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
    op += (uint32_t)__builtin_popcountll(LIT) + 2u * (uint32_t)__builtin_popcountll(H) + (uint32_t)__builtin_popcountll(H3) +
          (uint32_t)__builtin_popcountll(LS);
    next_emit = base + (by_copy ? r : last_end);
}

// The walk of one segment.  `inter` = HIT | stop lanes (stop = DUP, long matches, and every lane >= hi), `lenv` = per-lane
// match length for HIT lanes (4..63), r < hi <= 64 the cursor lane, B the number of stride-1 probes still allowed (:339).
// Per match: skip the misses (first set bit of inter), take the hit (H), cover its lanes (COV), continue behind it.
// Returns why it stopped: 0 = behind a copy at a lane >= hi, 1 = no hit within the next B lanes (r NOT advanced),
// 2 = r is a stop lane (not probed).  On gfx950 this is 16 instructions per match, hand-scheduled; the C++ body is
// the same algorithm for the CPU emulator.
__device__ __forceinline__ uint32_t segment_walk(unsigned long long inter, unsigned long long stopm, uint32_t lenv, uint32_t hi,
                                                 uint32_t& r, uint32_t& B, unsigned long long& H, unsigned long long& COV)
{
    uint32_t why;
#ifdef SNAPPY_EMU
    for (;;) {
        const unsigned long long m = inter >> r;
        const uint32_t f = m ? (uint32_t)__builtin_ctzll(m) : 0xffffffffu;
        if (f >= B) {
            why = 1;
            break;
        }
        r += f;
        B -= f;
        if ((stopm >> r) & 1ull) {
            why = 2;
            break;
        }
        H |= 1ull << r;
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)lenv, (int)r);
        COV |= ((1ull << e) - 1ull) << r;
        r += e;
        B = 33;
        if (r < hi) continue;
        why = 0;
        break;
    }
#else
    unsigned long long m;
    uint32_t f, e;
    asm volatile(
        "1:\n"
        "  s_lshr_b64 %[m], %[inter], %[r]\n"
        "  s_ff1_i32_b64 %[f], %[m]\n"          // -1 when no bit is set
        "  s_cmp_ge_u32 %[f], %[B]\n"
        "  s_cbranch_scc1 2f\n"
        "  s_add_u32 %[r], %[r], %[f]\n"
        "  s_sub_u32 %[B], %[B], %[f]\n"
        "  s_bitcmp1_b64 %[stopm], %[r]\n"
        "  s_cbranch_scc1 3f\n"
        "  s_bitset1_b64 %[H], %[r]\n"
        "  v_readlane_b32 %[e], %[lenv], %[r]\n"
        "  s_bfm_b64 %[m], %[e], %[r]\n"        // ((1 << e) - 1) << r
        "  s_or_b64 %[COV], %[COV], %[m]\n"
        "  s_add_u32 %[r], %[r], %[e]\n"
        "  s_movk_i32 %[B], 33\n"
        "  s_cmp_lt_u32 %[r], %[hi]\n"
        "  s_cbranch_scc1 1b\n"
        "  s_mov_b32 %[why], 0\n"
        "  s_branch 4f\n"
        "2:\n"
        "  s_mov_b32 %[why], 1\n"
        "  s_branch 4f\n"
        "3:\n"
        "  s_mov_b32 %[why], 2\n"
        "4:\n"
        : [r] "+s"(r), [B] "+s"(B), [H] "+s"(H), [COV] "+s"(COV), [m] "=&s"(m), [f] "=&s"(f), [e] "=&s"(e), [why] "=&s"(why)
        : [inter] "s"(inter), [stopm] "s"(stopm), [lenv] "v"(lenv), [hi] "s"(hi)
        : "scc");
#endif
    return why;
}

// The scan state of one block between two steps of the parse (snappy_compress.c:284-413): the cursor, the reference's skip
// counter (:333, :339; 31 marks "the next probe is the one right after a copy", :393-398, which moves on by one position
// like a stride-1 scan probe and leaves skip = 32 behind), the output offset in the slot and the start of the pending literal.
struct ParseState {
    uint32_t ip = 1;          // :305
    uint32_t skip = 32;       // :333
    uint32_t op = 4;          // :291
    uint32_t next_emit = 0;   // :298
};

// The bulk parse from `ps` on: steps until the scan is over (returns true: the caller emits the remainder, :405-410) or --
// after at least one step -- the cursor has reached stop_ip at stride 1 (returns false; the table, the slot and `ps` are
// exactly the reference's state in front of the probe at ps.ip, so any form of the parse can take over).
// kAtWindowEntry (the ceiling experiment's table, csrc/ablation/k1_oracle_table.hpp): hand over only where the cursor has just
// entered a window this form has not gathered, so that every window is gathered once, at its first probe.
template <class Table, uint32_t kChunk, bool kAtWindowEntry = false>
__device__ __forceinline__ bool bulk_run(const uint8_t* __restrict__ blk, uint32_t avail, uint32_t n, uint32_t shift,
                                         uint8_t* __restrict__ dst, const Table table, uint32_t lane, lds_bytes_t dup_scratch,
                                         ParseState& ps, uint32_t stop_ip)
{
    using State = MaskedWindowState<Table, kChunk>;
    uint32_t op = ps.op;
    uint32_t next_emit = ps.next_emit;
    bool finished = true;
    {
        const uint32_t limit = n - kInputMargin;
        CursorWindow win;
        win.blk = blk;
        win.avail = avail;
        win.shift = shift;
        win.reset(ps.ip, lane);
        State st;
        uint32_t ip = ps.ip;
        uint32_t skip = ps.skip;
        for (uint32_t iter = 0;; ++iter) {
            const uint32_t stride = skip >> 5;
            const uint32_t step = stride ? stride : 1u;
            if (ip + step > limit) break;                        // :342-343 / :388-389
            if (iter && ip >= stop_ip && skip < 64u && (!kAtWindowEntry || ip >= win.base + 64u)) {
                finished = false;
                break;
            }
            if (win.ensure(ip, lane)) st.invalidate();
            uint32_t r = ip - win.base;
            if (r >= uni(st.cov_end)) st.template gather<true, true>(table, win, dup_scratch, r, kChunk, lane, n);
            unsigned long long stopm = st.dup | st.longm;

            bool need_single = true;
            if (stride <= 1 || !((stopm >> r) & 1ull)) {
                // ---------------- segment ----------------
                uint32_t hi = uni(st.cov_end);
                const uint32_t lim = limit - win.base;           // lanes below may be probed (position + 1 <= limit)
                hi = lim < hi ? lim : hi;
                if (hi < kWave) stopm |= ~0ull << hi;
                unsigned long long inter = st.hit | stopm;
                unsigned long long pre = 0;                      // lanes probed by the strided prefix
                uint32_t B = 64u - skip;                         // stride-1 probes left before :339 widens the stride
                if (stride > 1) {
                    // The scan is at stride s (:339): lanes r, r+s, ... are probed until one hits, the stride level is used
                    // up (32 probes per level), or the window / limit ends.  Find the first interesting one with a mask.
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
                        State::commit(table, win, pre, lane);                // (no DUP lane among them: those are stops)
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
                for (;;) {
                    why = segment_walk(inter, stopm, 4u + st.extv, hi, r, B, H, COV);
                    if (why != 2 || r >= hi) break;
                    // The walk stands on a DUP lane or on a hit of 12+ bytes.  Settle that one probe here -- from registers
                    // for a DUP lane, with the 64-lane extender for a long match -- patch the lane's cached entry / match
                    // length, and let the walk go on: the segment, its one table commit and its one emission continue.
                    bool hit_r;
                    uint32_t cand_r, ext_r;
                    uint32_t sat_r = 8;                          // where ext_r saturates: 8, or 24 for a deep lane's own result
                    if ((st.dup >> r) & 1ull) {
                        const uint32_t hr = (uint32_t)__builtin_amdgcn_readlane((int)win.h0, (int)r);
                        const unsigned long long inner = COV & ~H;
                        const unsigned long long so_far = st.inserted | pre | (((~0ull << r0) & ((1ull << r) - 1ull)) & ~inner) |
                                                          (COV & ~(inner >> 1));
                        const unsigned long long J = __ballot(win.h0 == hr) & so_far & ((1ull << r) - 1ull);
                        if (J) {                                 // the slot holds the latest inserted lane with this hash
                            const uint32_t j = 63u - (uint32_t)__builtin_clzll(J);
                            cand_r = win.base + j;
                            hit_r = (uint32_t)__builtin_amdgcn_readlane((int)win.x0, (int)r) ==
                                    (uint32_t)__builtin_amdgcn_readlane((int)win.x0, (int)j);
                            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)r) ^
                                                (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)j);
                            const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)r) ^
                                                (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)j);
                            ext_r = d0 ? ((uint32_t)__builtin_ctz(d0) >> 3) : (d1 ? 4u + ((uint32_t)__builtin_ctz(d1) >> 3) : 8u);
                        } else {                                 // nothing inserted since the gather: its result stands
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
                    if (hit_r && ext_r == sat_r)
                        len_r = 4u + sat_r + match_extend(blk, cand_r + 4u + sat_r, win.base + r + 4u + sat_r, n, lane);
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
                if (why == 1) {                                  // B (or all remaining) lanes of misses
                    const uint32_t room = hi - r;
                    const uint32_t adv = B < room ? B : room;
                    r += adv;
                    B -= adv;
                }
                ip = win.base + r;
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
                State::commit(table, win, C & ~st.dup, lane);
                st.inserted |= C;
                for (unsigned long long d = C & st.dup; d; d &= d - 1)                         // shared slots: in position order
                    State::commit(table, win, d & (~d + 1), lane);

                if (H) emit_segment(dst, blk, op, next_emit, win.base, win.x0, st.ent, st.extv, H, COV, why == 0, r, lane);
                if (done) break;
                if (why == 0 && r > kWave) {                     // :391-392 for a copy that ended in a later window
                    if (win.ensure(ip - 1, lane)) st.invalidate();
                    State::commit(table, win, 1ull << (ip - 1 - win.base), lane);
                    st.inserted |= 1ull << (ip - 1 - win.base);
                }
                if (!need_single) continue;
                r = ip - win.base;                               // a copy of 64+ bytes starts here
            }

            // ---------------- single step: DUP lane, long match, or stride > 1 ----------------
            uint32_t cand = 0, ext = 0, sat = 8;
            bool hit;
            // A DUP lane shares its hash with other lanes of the window.  If one of them (below r) has been inserted since
            // the gather, the table slot holds the LATEST such lane j: candidate, hit test and match head all come from
            // window registers.  Otherwise the gathered entry still stands, exactly as for a lane that shares nothing.
            unsigned long long J = 0;
            if ((st.dup >> r) & 1ull) {
                const uint32_t hr = win.hash_at(ip);
                J = __ballot(win.h0 == hr) & st.inserted & ((1ull << r) - 1ull);
            }
            State::commit(table, win, 1ull << r, lane);
            st.inserted |= 1ull << r;
            if (J) {
                const uint32_t j = 63u - (uint32_t)__builtin_clzll(J);
                cand = win.base + j;
                hit = win.bytes_at(ip) == (uint32_t)__builtin_amdgcn_readlane((int)win.x0, (int)j);
                if (hit) {
                    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)r) ^
                                        (uint32_t)__builtin_amdgcn_readlane((int)st.xa, (int)j);
                    const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)r) ^
                                        (uint32_t)__builtin_amdgcn_readlane((int)st.xb, (int)j);
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
            if (ip > next_emit) op = emit_literal_windowed(dst, op, blk, next_emit, ip - next_emit, win.base, win.x0, lane);   // :355
            const uint32_t mbase = ip;
            uint32_t matched = 4 + ext;                          // find_match_length (:176-193)
            if (ext == sat) matched = 4 + sat + match_extend(blk, cand + 4 + sat, ip + 4 + sat, n, lane);
            ip += matched;
            op = emit_copy_packed(dst, op, mbase - cand, matched, lane);
            next_emit = ip;
            if (ip >= limit) break;                              // :388-389
            if (win.ensure(ip - 1, lane)) st.invalidate();
            State::commit(table, win, 1ull << (ip - 1 - win.base), lane);   // :391-392
            st.inserted |= 1ull << (ip - 1 - win.base);
            skip = 31;
        }
        ps.ip = ip;
        ps.skip = skip;
    }
    ps.op = op;
    ps.next_emit = next_emit;
    return finished;
}

// emit_remainder (:405-410) and the size prefix (:412)
__device__ __forceinline__ void finish_block(const uint8_t* __restrict__ blk, uint32_t n, uint8_t* __restrict__ dst, const ParseState& ps,
                                             uint32_t lane, uint32_t* __restrict__ block_bytes_out)
{
    uint32_t op = ps.op;
    if (ps.next_emit < n) op = emit_literal(dst, op, blk + ps.next_emit, n - ps.next_emit, lane);
    if (lane == 0) {
        st32(dst, op - 4);
        *block_bytes_out = op;
    }
    __builtin_amdgcn_wave_barrier();
}

template <class Table, uint32_t kChunk>
__device__ __forceinline__ void compress_one_block_bulk(const uint8_t* __restrict__ base16, uint64_t start, uint64_t in_len,
                                                        uint32_t n, uint8_t* __restrict__ dst, const Table table_in, uint32_t lane,
                                                        uint32_t* __restrict__ block_bytes_out, lds_bytes_t dup_scratch)
{
    const uint8_t* __restrict__ blk = base16 + start;
    const uint32_t ts = table_entries_for(n);                    // get_hash_table, :139-146 (+ shift, :288)
    const uint32_t shift = (uint32_t)__builtin_clz(ts) + 1;
    // "empty" = candidate position 0 (:346 on a zeroed table), carrying position 0's tag
    const uint32_t e_zero = (n >= kInputMargin) ? (((uld32(blk) * kHashMul) << (32 - shift)) & 0xffff0000u) : 0u;
    const Table table = table_in.with_empty(e_zero);
    if (n >= kInputMargin) table.init(ts, e_zero, lane);
    __builtin_amdgcn_wave_barrier();
    ParseState ps;
    if (n >= kInputMargin) {  // :301
        const uint64_t left = in_len - start;
        const uint32_t avail = (left < 0x7fffffffull) ? (uint32_t)left : 0x7fffffffu;
        bulk_run<Table, kChunk>(blk, avail, n, shift, dst, table, lane, dup_scratch, ps, 0xffffffffu);
    }
    finish_block(blk, n, dst, ps, lane, block_bytes_out);
}

}  // namespace snappy_hip
#include "snappy_k1_stream.hpp"
namespace snappy_hip {

// dynamic LDS of the LDS-table kernel in its stream form: the u16 table + the two tables of analyse()
__host__ __device__ inline uint32_t lds_table_stream_lds_bytes(uint32_t block_size);

// ~3.4 us per iteration (s_sleep 127 = 127 x 64 cycles).  The hybrid K1 launch puts a few of these in front of the
// global-table kernel so that the LDS-table workgroups of the co-running kernel (33 KiB of LDS each) are placed first:
// if the small LDS allocations of the global-table wavefronts land first they fragment the LDS and only two of the three
// LDS-table workgroups per CU fit (block share 13-16 % instead of 22 %).
__global__ __launch_bounds__(64) void delay_kernel(uint32_t iters)
{
#ifndef SNAPPY_EMU
    for (uint32_t i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
#else
    (void)iters;
#endif
}

// One K1 launch can serve several containers (independent inputs with their own slot / size arrays): the persistent
// wavefronts draw GLOBAL block numbers and map them to (container, block) here, so a batch has one tail instead of one per
// container.  Passed by value; the kernel argument segment is indexed with scalar loads.
constexpr uint32_t kMaxBatch = 8;
struct K1Batch {
    uint32_t count;
    uint32_t first_block[kMaxBatch + 1];   // first_block[count] = total number of blocks
    const uint8_t* in[kMaxBatch];
    uint64_t in_len[kMaxBatch];
    uint8_t* slots[kMaxBatch];
    uint32_t* block_bytes[kMaxBatch];
};
__device__ __forceinline__ uint32_t batch_container_of(const K1Batch& w, uint32_t b)
{
    uint32_t c = 0;
    while (c + 1 < w.count && b >= w.first_block[c + 1]) ++c;
    return c;
}

// u16 table entries the LDS-table kernels reserve for blocks of up to block_size bytes (= table_entries_for(block_size))
__host__ __device__ inline uint32_t lds_table_entries(uint32_t block_size)
{
    uint32_t ts = 256;
    while (ts < kMaxTableEntries && ts < block_size) ts <<= 1;
    return ts;
}
__host__ __device__ inline uint32_t lds_table_kernel_lds_bytes(uint32_t block_size, bool with_dup_scratch)
{
    return 2u * lds_table_entries(block_size) + (with_dup_scratch ? kDupSlots : 16u);
}
__host__ __device__ inline uint32_t lds_table_stream_lds_bytes(uint32_t block_size)
{
    return 2u * lds_table_entries(block_size) + stream_scratch_bytes(kStreamSlotsLds);
}

// next_block == nullptr: static grid-stride assignment; otherwise blocks are drawn from the shared atomic counter,
// which lets this kernel run CONCURRENTLY with compress_blocks_global_table_kernel on the same container (the
// LDS-table waves fill 5 wave slots per CU with low-latency tables, the global-table waves the other 27).
template <uint32_t kAhead, int kForm = 0>
__global__ __launch_bounds__(64) void compress_blocks_lds_table_kernel(const K1Batch w, uint32_t block_size, uint32_t slot_stride,
                                                                       uint32_t* next_block)
{
    const uint32_t num_blocks = w.first_block[w.count];
    // Dynamic LDS, sized by the launch from the block size (lds_table_kernel_lds_bytes): the u16 table of
    // table_entries_for(block_size) entries -- 512 B for -b 256 ... 32 KiB from -b 16384 up, as the reference sizes its
    // table to the block (snappy_compress.c:139-146; dpu_compress.c:472-476 to the tasklet's memory) -- then the 1 KiB
    // duplicate-slot scratch of the masked / bulk forms.  Small block sizes therefore fit many more of these wavefronts per CU.
    HIP_DYNAMIC_SHARED(uint8_t, lds_dyn)
    uint16_t* table = reinterpret_cast<uint16_t*>(lds_dyn);
    uint8_t* dup_scratch = lds_dyn + 2u * lds_table_entries(block_size);
    const uint32_t lane = threadIdx.x;
#ifndef SNAPPY_EMU
    // The LDS-table wavefronts are few (LDS capacity) but cost no table traffic: let the instruction arbiter prefer them
    // over the global-table wavefronts they share a SIMD with.
    if (next_block) __builtin_amdgcn_s_setprio(3);
#endif
    uint32_t b = blockIdx.x;
    for (;;) {
        if (next_block) {
            uint32_t drawn = 0;
            if (lane == 0) drawn = atomicAdd(next_block, 1u);
            b = uni(drawn);
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
        static_assert(kForm == 2 || kForm == 3, "the bulk (2) and the stream (3) form of the parse exist; rounds 1-2's other forms are in the history (profiles/HISTORY.md)");
        if constexpr (kForm == 3)   // (launched with lds_table_stream_lds_bytes(block_size) of dynamic LDS)
        {
            SoloMate solo;
            compress_one_block_stream<LdsTable, kStreamSlotsLds>(in, start, in_len, n, slot, LdsTable{table}, lane, bytes_out,
                                                                 (lds_bytes_t)dup_scratch, solo);
        }
        else
            compress_one_block_bulk<LdsTable, kAhead>(in, start, in_len, n, slot, LdsTable{table}, lane, bytes_out,
                                                      (lds_bytes_t)dup_scratch);
        if (next_block && lane == 0) atomicAdd(next_block + 4, 1u);   // statistics: blocks taken by the LDS-table form
        __syncthreads();
        b += gridDim.x;
    }
}

template <uint32_t kAhead, int kForm = 0, int kFilter = 0, uint32_t kCacheSlots = 0>
__global__ __launch_bounds__(64) void compress_blocks_global_table_kernel(const K1Batch w, uint32_t block_size, uint32_t slot_stride,
                                                                          uint32_t* table_scratch, uint32_t* next_block)
{
    const uint32_t num_blocks = w.first_block[w.count];
    __shared__ __attribute__((aligned(16))) uint8_t dup_scratch[kForm == 3 ? stream_scratch_bytes(kStreamSlotsGlobal) : (kForm ? kDupSlots : 16)];
    __shared__ __attribute__((aligned(16))) uint32_t slot_state[kMaxTableEntries / 32];   // the "slot written in this block" filter
    __shared__ __attribute__((aligned(16))) uint32_t slot_cache[kCacheSlots ? kCacheSlots : 4];
    const uint32_t lane = threadIdx.x;
    static_assert((kForm == 2 || kForm == 3) && kFilter == 1, "the bulk (2) and the stream (3) form behind the slot filter exist; rounds 1-2's other forms are in the history (profiles/HISTORY.md)");
    using Narrow = FilteredGlobalTable;
    using Table = typename std::conditional<kCacheSlots != 0, CachedGlobalTable<(kCacheSlots ? kCacheSlots : 1024u)>, Narrow>::type;
    Table table;
    if constexpr (kCacheSlots != 0) {
        table.t = (uint16_t*)table_scratch + (size_t)blockIdx.x * kMaxTableEntries;
        table.written = (lds_words_t)slot_state;
        table.cache = (lds_words_t)slot_cache;
    } else {
        table.t = table_scratch + (size_t)blockIdx.x * kMaxTableEntries;
    }
    if constexpr (kFilter == 1 && kCacheSlots == 0) {
        table.written = (lds_words_t)slot_state;
        table.empty = 0;
    }
    for (;;) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(next_block, 1u);
        b = uni(b);
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
        if constexpr (kForm == 3)
        {
            SoloMate solo;
            compress_one_block_stream<Table, kStreamSlotsGlobal>(in, start, in_len, n, slot, table, lane, bytes_out, (lds_bytes_t)dup_scratch,
                                                                 solo);
        }
        else
            compress_one_block_bulk<Table, kAhead>(in, start, in_len, n, slot, table, lane, bytes_out, (lds_bytes_t)dup_scratch);
    }
}

#ifdef SNAPPY_ABLATION
}  // namespace snappy_hip
#include "ablation/k1_oracle_table.hpp"
namespace snappy_hip {
#endif

// Single-workgroup exclusive scan (<= 131072 blocks per 4 GiB container at 32 KiB; any count
// works, it loops).  Also writes the two header varints (snappy_compress.c:461-465).
__global__ __launch_bounds__(1024) void scan_block_bytes_kernel(const uint32_t* __restrict__ block_bytes,
                                                                uint32_t num_blocks, uint32_t total_len,
                                                                uint32_t block_size, uint8_t* __restrict__ stream,
                                                                uint64_t* __restrict__ offsets,
                                                                uint64_t* __restrict__ stream_len)
{
    __shared__ uint64_t wave_sums[16];
    __shared__ uint64_t carry_s;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;

    // header: at most 5 + 5 bytes, thread 0
    uint32_t hdr_len = 0;
    {
        uint8_t hb[10];
        uint32_t v = total_len;
        while (v >= 0x80) { hb[hdr_len++] = (uint8_t)(v | 0x80); v >>= 7; }
        hb[hdr_len++] = (uint8_t)v;
        v = block_size;
        while (v >= 0x80) { hb[hdr_len++] = (uint8_t)(v | 0x80); v >>= 7; }
        hb[hdr_len++] = (uint8_t)v;
        if (tid == 0)
            for (uint32_t i = 0; i < hdr_len; ++i) stream[i] = hb[i];
    }
    if (tid == 0) carry_s = hdr_len;
    __syncthreads();

    for (uint32_t base = 0; base < num_blocks; base += 1024) {
        const uint32_t i = base + tid;
        const uint64_t mine = (i < num_blocks) ? (uint64_t)block_bytes[i] : 0;
        // inclusive scan inside the wave
        uint64_t x = mine;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, (int)d);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), (int)d);
            if (lane >= d) x += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) wave_sums[wave] = x;
        __syncthreads();
        uint64_t before = carry_s;
        for (uint32_t w = 0; w < wave; ++w) before += wave_sums[w];
        if (i < num_blocks) offsets[i] = before + x - mine;
        __syncthreads();
        if (tid == 1023) carry_s = before + x;
        __syncthreads();
    }
    if (tid == 0) {
        offsets[num_blocks] = carry_s;
        if (stream_len) *stream_len = carry_s;
    }
}

// One 256-thread workgroup per block: copy block_bytes[b] bytes from the slot (16-byte aligned)
// to stream + offsets[b] (arbitrary alignment).  dword stores on the aligned middle.
__global__ __launch_bounds__(256) void gather_slots_kernel(const uint8_t* __restrict__ slots, uint32_t slot_stride,
                                                           const uint32_t* __restrict__ block_bytes,
                                                           const uint64_t* __restrict__ offsets,
                                                           uint8_t* __restrict__ stream, uint32_t num_blocks)
{
    for (uint32_t b = blockIdx.x; b < num_blocks; b += gridDim.x) {
        const uint8_t* __restrict__ src = slots + (uint64_t)b * slot_stride;
        uint8_t* __restrict__ dst = stream + offsets[b];
        const uint32_t len = block_bytes[b];
        const uint32_t head = (uint32_t)((4 - ((uintptr_t)dst & 3)) & 3);   // bytes until dst is dword aligned
        const uint32_t h = head < len ? head : len;
        if (threadIdx.x < h) dst[threadIdx.x] = src[threadIdx.x];
        const uint32_t body = (len - h) & ~3u;
        uint32_t* __restrict__ d32 = reinterpret_cast<uint32_t*>(dst + h);
        for (uint32_t i = threadIdx.x * 4; i < body; i += 256 * 4) d32[i >> 2] = ld32(src + h + i);
        const uint32_t done = h + body;
        if (done + threadIdx.x < len) dst[done + threadIdx.x] = src[done + threadIdx.x];
    }
}

// ---------------------------------------------------------------------------
// size-chain index (device form of snappy_decompress.c:317-340); one wave per stream
// ---------------------------------------------------------------------------
struct StreamDesc {            // must match snappy_hip_stream_desc (include/snappy_hip.h)
    const uint8_t* stream;
    uint64_t stream_len;
    uint64_t* block_offsets;
    uint32_t* result;
    uint32_t total_len;
    uint32_t block_size;
    uint32_t header_len;
    uint32_t num_blocks;
};

// The chain of u32 block sizes is serial by format: 65536 dependent hops for a 2 GiB container, 950 cycles per hop when the
// size field comes from HBM, 200 when it is in L2 (tools/microbench/l2_warm_probe).  So each stream gets a GROUP of
// workgroups on one XCD (workgroup i runs on XCD i mod 8): one walks -- wave 0 of the first -- and the others only read
// ahead, streaming the compressed bytes up to kIndexAheadSupers x 32 KiB in front of the walker through that XCD's L2 (one
// CU cannot stream fast enough: ~40 GB/s).  The walker publishes its position in result[0] (top bit = "running"); the
// readers stop when it stores the final status there.  If the placement assumption does not hold the walk is merely as
// slow as without readers.
constexpr uint32_t kIndexGroup = 64;           // workgroups launched per stream; those on the stream's XCD take part
constexpr uint32_t kIndexReaderWgs = 7;
constexpr uint32_t kIndexWgWaves = 4;
constexpr uint32_t kIndexReaders = kIndexReaderWgs * kIndexWgWaves;
constexpr uint32_t kIndexSuper = 8 * 4096;     // one read-ahead step of a wave: 8 loads x 64 lanes x 64 bytes apart
constexpr uint32_t kIndexAheadSupers = 64;     // stay at most 2 MiB in front of the walker
constexpr uint32_t kIndexRunning = 0x80000000u;
__global__ __launch_bounds__(64 * kIndexWgWaves) void index_streams_kernel(const StreamDesc* __restrict__ descs, uint32_t count,
                                                                           uint32_t group, const uint32_t* __restrict__ resolved = nullptr)
{
    // group == 1: one workgroup per stream, no readers (also what the CPU emulator runs)
    const uint32_t s = blockIdx.x / group;
    const uint32_t j = blockIdx.x % group;
    if (s >= count) return;
    if (resolved && resolved[s]) return;        // the parallel segments (chain_*_kernel, below) have laid the chain out already
    uint32_t role = 0;                          // 0 = walker, 1.. = reader workgroup
    if (group > 1) {
        if ((j & 7u) != (s & 7u)) return;       // not on this stream's XCD
        role = j >> 3;
        if (role > kIndexReaderWgs) return;
    }
    const StreamDesc d = descs[s];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    volatile uint32_t* ctl = reinterpret_cast<volatile uint32_t*>(d.result);
    if (role == 0) {
        if (wave != 0) return;
        uint64_t at = d.header_len;
        uint32_t status = kBlockOk;
        uint32_t i = 0;
        if (group > 1 && lane == 0) ctl[0] = kIndexRunning;
        // offsets are collected in registers (lane i & 63 keeps hop i's) and stored 64 at a time, so that a hop is the
        // dependent load and nothing else: a store per hop would put its acknowledgement on every hop's wait
        uint64_t mine_off = 0;
        for (; i < d.num_blocks; ++i) {
            if (at + 4 > d.stream_len) {
                status = kBlockInvalid;
                break;
            }
            if (lane == (i & 63u)) mine_off = at;
            if ((i & 63u) == 63u) {
                d.block_offsets[(i & ~63u) + lane] = mine_off;
                if (group > 1 && lane == 0) ctl[0] = kIndexRunning | (uint32_t)(at / kIndexSuper);
            }
            at += 4 + (uint64_t)uld32(d.stream + at);
        }
        if ((i & 63u) != 0 && lane < (i & 63u)) d.block_offsets[(i & ~63u) + lane] = mine_off;   // the last partial group
        if (status == kBlockOk && at != d.stream_len) status = kBlockInvalid;
        if (lane == 0) {
            d.result[1] = i;
            __threadfence();
            ctl[0] = status;                    // also tells the readers to stop
        }
        return;
    }
    // reader wave h of kIndexReaders takes the super-chunks c = h (mod kIndexReaders)
    const uint32_t mine = (role - 1u) * kIndexWgWaves + wave;
    // super-chunk c is read at byte offsets below (c + 1) * kIndexSuper: it must lie inside the stream entirely
    const uint64_t whole = d.stream_len / kIndexSuper;
    if (whole == 0) return;                     // a stream shorter than one super-chunk: nothing to read ahead
    const uint32_t last = (uint32_t)(whole - 1u);
    uint32_t c = mine;
    uint32_t sink = 0;
    bool seen_running = false;
    uint32_t patience = 4096;                   // bounded wait for the walker to start (it may be scheduled later)
    for (;;) {
        const uint32_t w = uni(ctl[0]);
        if (!(w & kIndexRunning)) {
            if (seen_running || --patience == 0) break;
            __builtin_amdgcn_s_sleep(16);
            continue;
        }
        seen_running = true;
        const uint32_t lo = w & ~kIndexRunning;
        if (c < lo) c = lo + ((mine + kIndexReaders - (lo % kIndexReaders)) % kIndexReaders);   // catch up with the walker
        if (c > last) break;                    // the tail of the stream: nothing more to warm
        if (c <= lo + kIndexAheadSupers) {
            const uint8_t* __restrict__ q = d.stream + (uint64_t)c * kIndexSuper + (uint64_t)lane * 64u;
            const uint32_t v0 = ld32(q), v1 = ld32(q + 4096), v2 = ld32(q + 8192), v3 = ld32(q + 12288);
            const uint32_t v4 = ld32(q + 16384), v5 = ld32(q + 20480), v6 = ld32(q + 24576), v7 = ld32(q + 28672);
            sink ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
            c += kIndexReaders;
        } else {
            __builtin_amdgcn_s_sleep(16);
        }
    }
    // keeps the loads alive: a condition the compiler cannot decide (a stream never has 2^32 - 1 blocks)
    if (sink == 0x9e3779b9u && d.num_blocks == 0xffffffffu) d.block_offsets[0] = sink;
}

// ---------------------------------------------------------------------------
// The size chain in parallel segments (round 4).  The walk above is one chain of dependent loads, 162 ns per hop even with
// the bytes read ahead into L2: 5.4 ms per 1 GiB container.  Nothing in the FORMAT marks a block boundary, but a boundary
// can be recognised with near certainty: the u32 there is a plausible compressed size (1 .. 32 + BS + BS/6, the most a
// conforming compressor emits for BS bytes), the element behind it is a literal (a block cannot start with a copy), the
// chain that starts there keeps landing on such positions, and some such position within one maximal block BEFORE it points
// exactly at it.  So kChainSegments walkers per stream each look for the first boundary at or behind their share of the
// stream (chain_anchor_kernel) and walk from there to the next walker's starting point (chain_walk_kernel); the segments
// are then laid end to end (chain_finish_kernel).
// Exactness does not rest on the recognition: segment 0 starts at the header's end, which IS a boundary, and a walk that
// starts on a boundary and ends EXACTLY on the next walker's starting point proves that point to be a boundary too (it is
// on the chain) -- by induction all recorded hops are the chain of snappy_decompress.c:317-340 iff every segment ended on
// the next one's start, the last on the stream's end, and the hops number num_blocks.  If any of that fails (a misjudged
// starting point: the walker before it runs past it; blocks too small for the segment buffers; a damaged stream) the
// stream stays unresolved and index_streams_kernel walks it as before.
// ---------------------------------------------------------------------------
constexpr uint32_t kChainSegments = 256;            // walkers per stream
constexpr uint32_t kChainSegCap = 2048;             // hops a walker can record (segments of ~2 MB hold ~130 blocks of 32 KiB)
constexpr uint32_t kChainMinSegBytes = 1u << 17;    // shorter streams use fewer walkers
constexpr uint32_t kChainNone = 0xffffffffu;
constexpr uint32_t kChainTries = 64;                // candidates a walker examines before it gives up its share
#ifdef SNAPPY_EMU
constexpr uint32_t kChainMinBlocks = 1;             // (the CPU emulator's inputs are small: it always exercises the walkers)
#else
constexpr uint32_t kChainMinBlocks = 2048;          // shorter chains go to the serial walk (0.17 us per hop)
#endif
struct ChainWork {                                  // device workspace of one index_streams call: [stream][segment]
    uint32_t* anchor;                               // where the segment's walker starts (kChainNone: no walker)
    uint32_t* seg_hops;                             // hops it recorded
    uint32_t* seg_ok;                               // it ended exactly on the next walker's start (or the stream's end)
    uint32_t* hops;                                 // [stream][segment][kChainSegCap] offsets
    uint32_t* resolved;                             // [stream]: 1 = block_offsets / result hold the chain
};

// the most compressed bytes a conforming compressor emits for a block of up to block_size bytes (snappy_compress.c:55-60)
__device__ __forceinline__ uint32_t chain_max_block(uint32_t block_size) { return 32u + block_size + block_size / 6u; }

// Could a block start at stream offset o?  (per lane; every byte read lies inside the stream)
__device__ __forceinline__ bool chain_plausible(const uint8_t* __restrict__ st, uint64_t len, uint32_t header_len, uint32_t maxc, uint64_t o)
{
    if (o < header_len || o + 5u > len) return false;
    const uint32_t size = ld32(st + o);
    return size - 1u < maxc && o + 4u + size <= len && (st[o + 4u] & 3u) == 0u;
}
// wave-uniform form of the same test; also hands back where the chain goes from there
__device__ __forceinline__ bool chain_plausible_uni(const uint8_t* __restrict__ st, uint64_t len, uint32_t header_len, uint32_t maxc, uint64_t o,
                                                    uint64_t& next)
{
    if (o < header_len || o + 5u > len) return false;
    const uint32_t size = uld32(st + o);
    next = o + 4u + size;
    return size - 1u < maxc && next <= len && (uni((uint32_t)st[o + 4u]) & 3u) == 0u;
}

__device__ __forceinline__ uint32_t chain_segments_of(uint64_t len, uint32_t header_len)
{
    const uint64_t body = len > header_len ? len - header_len : 0;
    const uint64_t want = body / kChainMinSegBytes;
    return want < 1 ? 1u : (want > kChainSegments ? kChainSegments : (uint32_t)want);
}

// grid: count x kChainSegments workgroups of one wavefront
__global__ __launch_bounds__(64) void chain_anchor_kernel(const StreamDesc* __restrict__ descs, uint32_t count, ChainWork w)
{
    const uint32_t s = blockIdx.x / kChainSegments, k = blockIdx.x % kChainSegments;
    if (s >= count) return;
    const StreamDesc d = descs[s];
    const uint32_t lane = threadIdx.x;
    uint32_t found = kChainNone;
    const uint64_t len = d.stream_len;
    // (offsets are kept in 32 bits here: longer streams, and streams without blocks, are left to the serial walk)
    // and a chain of fewer than kChainMinBlocks hops is quicker walked than shared out: 312 blocks 0.06 ms serial against 0.18,
    // 1,564 blocks 0.27 either way, 2,571 blocks 0.44 against 0.22 -- profiles/r04_small_chain_threshold.txt)
    if (len < 0xffffffffull && d.num_blocks >= kChainMinBlocks && len > d.header_len) {
        const uint32_t segs = chain_segments_of(len, d.header_len);
        const uint32_t maxc = chain_max_block(d.block_size);
        if (k == 0) {
            found = d.header_len;
        } else if (k < segs) {
            const uint64_t body = len - d.header_len;
            const uint64_t lo = d.header_len + body * k / segs, hi = d.header_len + body * (k + 1) / segs;
            // (bounded work whatever the bytes are: a share whose first kChainTries candidates all fail gets no walker --
            // the walker before it carries on through this share, or the stream goes to the serial walk)
            uint32_t tries = 0;
            for (uint64_t pos = lo; pos < hi && found == kChainNone && tries < kChainTries; pos += kWave) {
                unsigned long long m = __ballot(pos + lane < hi && chain_plausible(d.stream, len, d.header_len, maxc, pos + lane));
                while (m && found == kChainNone && tries < kChainTries) {
                    const uint64_t a = pos + (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    ++tries;
                    // forward: the chain from `a` keeps landing on plausible starts (or reaches the end)
                    uint64_t at = a, nx = 0;
                    bool good = true;
                    for (uint32_t hop = 0; hop < 4 && good && at != len; ++hop) {
                        good = chain_plausible_uni(d.stream, len, d.header_len, maxc, at, nx);
                        at = nx;
                    }
                    if (!good) continue;
                    // backward: some plausible start within one maximal block before `a` points exactly at it
                    const uint64_t reach = (uint64_t)maxc + 4u;
                    const uint64_t from = a > reach + d.header_len ? a - reach : d.header_len;
                    bool pointed = false;
                    for (uint64_t c = from; c + 5u <= a && !pointed; c += kWave) {
                        const uint64_t o = c + lane;
                        bool hit = false;
                        if (o + 5u <= a && chain_plausible(d.stream, len, d.header_len, maxc, o)) hit = o + 4u + ld32(d.stream + o) == a;
                        pointed = __ballot(hit) != 0;
                    }
                    if (pointed) found = (uint32_t)a;
                }
            }
        }
    }
    if (lane == 0) w.anchor[s * kChainSegments + k] = found;
}

// grid: count x kChainSegments workgroups of one wavefront; after chain_anchor_kernel
__global__ __launch_bounds__(64) void chain_walk_kernel(const StreamDesc* __restrict__ descs, uint32_t count, ChainWork w)
{
    const uint32_t s = blockIdx.x / kChainSegments, k = blockIdx.x % kChainSegments;
    if (s >= count) return;
    const StreamDesc d = descs[s];
    const uint32_t lane = threadIdx.x;
    const uint32_t* __restrict__ anchors = w.anchor + s * kChainSegments;
    const uint32_t a = anchors[k];
    uint32_t n = 0, ok = 1;
    if (a != kChainNone) {
        // the next walker that has a starting point, else the stream's end
        uint64_t next = d.stream_len;
        for (uint32_t j = k + 1; j < kChainSegments; ++j) {
            const uint32_t aj = anchors[j];
            if (aj != kChainNone) {
                next = aj;
                break;
            }
        }
        uint32_t* __restrict__ out = w.hops + ((size_t)s * kChainSegments + k) * kChainSegCap;
        uint64_t at = a;
        uint32_t mine = 0;                                       // lane i & 63 keeps hop i until 64 are stored at once
        while (at < next) {
            if (n == kChainSegCap || at + 4u > d.stream_len) {
                ok = 0;
                break;
            }
            if (lane == (n & 63u)) mine = (uint32_t)at;
            if ((n & 63u) == 63u) out[(n & ~63u) + lane] = mine;
            ++n;
            at += 4u + (uint64_t)uld32(d.stream + at);
        }
        if (ok && (n & 63u) != 0 && lane < (n & 63u)) out[(n & ~63u) + lane] = mine;
        if (at != next) ok = 0;
    }
    if (lane == 0) {
        w.seg_hops[s * kChainSegments + k] = n;
        w.seg_ok[s * kChainSegments + k] = ok;
    }
}

// grid: count workgroups of 1024; after chain_walk_kernel.  Lays the segments end to end into block_offsets (num_blocks
// entries, as index_streams_kernel leaves them) and sets result / resolved -- or leaves the stream to the serial walk.
__global__ __launch_bounds__(1024) void chain_finish_kernel(const StreamDesc* __restrict__ descs, uint32_t count, ChainWork w)
{
    __shared__ uint32_t seg_start[kChainSegments + 1];
    __shared__ uint32_t all_ok;
    const uint32_t s = blockIdx.x;
    if (s >= count) return;
    const StreamDesc d = descs[s];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        uint32_t ok = w.anchor[s * kChainSegments] == d.header_len ? 1u : 0u;
        uint64_t sum = 0;                                        // (at most 256 x 2048 hops: no wrap)
        for (uint32_t k = 0; k < kChainSegments; ++k) {
            seg_start[k] = (uint32_t)sum;
            ok &= w.seg_ok[s * kChainSegments + k];
            sum += w.seg_hops[s * kChainSegments + k];
        }
        seg_start[kChainSegments] = (uint32_t)sum;
        all_ok = (ok && sum == d.num_blocks) ? 1u : 0u;
    }
    __syncthreads();
    if (!all_ok) {
        if (tid == 0) w.resolved[s] = 0;
        return;
    }
    for (uint32_t k = 0; k < kChainSegments; ++k) {
        const uint32_t base = seg_start[k], c = seg_start[k + 1] - base;
        const uint32_t* __restrict__ in = w.hops + ((size_t)s * kChainSegments + k) * kChainSegCap;
        for (uint32_t i = tid; i < c; i += 1024) d.block_offsets[base + i] = in[i];
    }
    if (tid == 0) {
        d.result[0] = kBlockOk;
        d.result[1] = d.num_blocks;
        w.resolved[s] = 1;
    }
}

// ---------------------------------------------------------------------------
// Verified side index.  The size chain is serial only when nothing is known about it.  When the caller already holds a
// candidate index -- the offsets the compressor's own scan produced, or an index stored beside the file -- every link of
// the chain can be CHECKED independently: offsets[0] is the header length, and for every block b the u32 stored at
// offsets[b] must lead exactly to offsets[b + 1], the last one to the end of the stream.  By induction that is the walk of
// snappy_decompress.c:317-340 with the same result, in one parallel pass over num_blocks + 1 words instead of num_blocks
// dependent loads.  A candidate that fails any link is rejected as a whole (result[0] = invalid); the caller then walks
// the chain with index_streams_kernel.  descs[s].block_offsets holds num_blocks + 1 entries here.
//   result[0] = SNAPPY_HIP_BLOCK_OK / _INVALID, result[1] = number of links that hold (num_blocks when the index is right)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void verify_index_begin_kernel(const StreamDesc* __restrict__ descs, uint32_t count)
{
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= count) return;
    const StreamDesc d = descs[s];
    uint32_t st = kBlockOk;
    if (d.num_blocks == 0) {
        if (d.stream_len != d.header_len) st = kBlockInvalid;
    } else if (d.block_offsets[0] != d.header_len || d.block_offsets[d.num_blocks] != d.stream_len) {
        st = kBlockInvalid;
    }
    d.result[0] = st;
    d.result[1] = 0;
}

constexpr uint32_t kVerifyGroup = 64;          // workgroups per stream
__global__ __launch_bounds__(256) void verify_index_kernel(const StreamDesc* __restrict__ descs, uint32_t count)
{
    const uint32_t s = blockIdx.x / kVerifyGroup;
    if (s >= count) return;
    const StreamDesc d = descs[s];
    uint32_t good = 0, bad = 0;
    for (uint32_t b = (blockIdx.x % kVerifyGroup) * 256 + threadIdx.x; b < d.num_blocks; b += kVerifyGroup * 256) {
        const uint64_t at = d.block_offsets[b];
        const uint64_t next = d.block_offsets[b + 1];
        // at + 4 <= next <= stream_len keeps the load inside the stream whatever the candidate holds
        if (next > d.stream_len || at > next || next - at < 4 || next - at - 4 != (uint64_t)ld32(d.stream + at))
            ++bad;
        else
            ++good;
    }
    // one atomic per wavefront, not per thread: 16384 threads per stream adding to one word take 1.4 ms for 20 us of checking
    const uint32_t wl = threadIdx.x & (kWave - 1);
    for (uint32_t m = kWave / 2; m; m >>= 1) good += (uint32_t)__shfl((int)good, (int)(wl ^ m));
    const unsigned long long any_bad = __ballot(bad != 0);
    if (wl == 0) {
        if (any_bad) atomicOr(d.result, kBlockInvalid);
        if (good) atomicAdd(d.result + 1, good);
    }
}

// ---------------------------------------------------------------------------
// K2: decompress.  One wavefront per block.
//
//  * The compressed stream is consumed through a 128-byte register window: lane l holds the 8 bytes at
//    compressed offset g+l (W0) and g+64+l (W1); W1 is the prefetch for the next 64 bytes.
//  * Every lane pre-decodes "the element that would start at my byte" (type, header size, output length,
//    offset) -- 64 candidate starts per window in a handful of VALU instructions.  The real element chain
//    is then followed with v_readlane only: no memory access to parse a tag.
//  * Literal payloads are stored to the output window straight from the window registers (lane l's
//    byte 0 IS compressed byte g+l); back-references are window->window replications (`lane % offset` by
//    reciprocal multiply), one gather/scatter per copy element.  (Packing several independent copies into one
//    64-lane gather/scatter was measured slower in both window forms: its lane bookkeeping costs more issue
//    slots than the round trips it saves once 24-32 wavefronts per CU hide them.)
//
// Semantics: snappy_decompress.c:232-285 on well-formed streams, strict otherwise (per-block status).
// ---------------------------------------------------------------------------


// Window loads: lane value = the 8 bytes at src[pos..pos+8), zero-filled at and beyond `avail`.
// Split in two so that a prefetch can stay in flight: window_issue() only issues the (clamped-address)
// load, window_value() applies the tail shift at the point of use.  For avail >= 8 (wave-uniform test)
// this is branch-free per lane; streams shorter than 8 bytes take a byte-wise path.
struct WindowLoad {
    uint64_t raw;
    uint32_t shift;   // bits; >= 64 means "all zero"
};
__device__ __forceinline__ WindowLoad window_issue(const uint8_t* __restrict__ src, uint64_t pos, uint64_t avail)
{
    WindowLoad r;
    if (avail >= 8) {
        const uint64_t last = avail - 8;
        const uint64_t a = (pos < last) ? pos : last;
        r.raw = ld64(src + a);
        const uint64_t sh = 8 * (pos - a);
        r.shift = (sh < 64) ? (uint32_t)sh : 64u;
    } else {
        r.raw = 0;
        r.shift = 0;
        for (uint32_t k = 0; k < 8; ++k)
            if (pos + k < avail) r.raw |= (uint64_t)src[pos + k] << (8 * k);
    }
    return r;
}
__device__ __forceinline__ uint64_t window_value(const WindowLoad& r) { return (r.shift < 64) ? (r.raw >> r.shift) : 0; }

// The same field rules for the per-window batch decoder, which wants the fields themselves (no packing) and is bounded by
// VALU issue: branch-free selects, 32-bit arithmetic only.  consumed = compressed bytes the element takes; `rejected` = the
// lanes whose element cannot be valid here (over-long literal, header or literal payload running past the block's compressed
// size).  Per-lane conditions of this decoder are kept as LANE MASKS built from single compares and combined with scalar
// logic: a ballot of a compound bool costs a v_cndmask + v_cmp on top of the compares.
__device__ __forceinline__ void predecode_window(uint64_t w, uint32_t pos, uint32_t csz, uint32_t& type, uint32_t& hdr, uint32_t& olen,
                                                 uint32_t& off, uint32_t& consumed, unsigned long long& rejected)
{
    const uint32_t tag = (uint32_t)w & 0xffu;
    type = tag & 3u;
    const uint32_t v = tag >> 2;
    const uint32_t next4 = (uint32_t)(w >> 8);
    const bool lit = type == 0;
    olen = (type == 1) ? (v & 7u) + 4u : v + 1u;                                // :264-266 / :271-273, :278-280, :249
    hdr = lit ? 1u : ((type == 3) ? 5u : type + 1u);
    const uint32_t c1_off = ((tag << 3) & 0x700u) | (next4 & 0xffu);
    off = lit ? 0u : ((type == 1) ? c1_off : ((type == 2) ? (next4 & 0xffffu) : next4));
    const bool long_lit = lit && v >= 60u;                                       // :250-255, :64-74: v - 59 = 1..4 length bytes
    const uint32_t raw = next4 & (0xffffffffu >> ((63u - v) * 8u & 31u));       // shift 24, 16, 8, 0 (only read when long_lit)
    olen = long_lit ? ((raw < 65536u) ? raw + 1u : 0u) : olen;                   // blocks are < 64 KiB
    hdr = long_lit ? v - 58u : hdr;
    consumed = hdr + (lit ? olen : 0u);
    rejected = __ballot(olen == 0) | __ballot(pos + consumed > csz);
}


// amdgpu_num_sgpr: measured on gfx950, the 81st SGPR costs the eighth wavefront per SIMD (8.3 -> 9.0 ms per container).
// ---------------------------------------------------------------------------
// K2, per-window batch (default).  PMC on the element-at-a-time loop showed K2 bound by the scalar unit (3.7e9 SALU
// instructions per 2 GiB container, 83 % of what the chip's scalar units can issue in the kernel's time, against 2.4e9 VALU).
// Here the only per-element scalar work left is following the element chain, and that over PAIRS of elements (k2_chain_walk
// on a doubled jump vector, the skipped starts filled in with one ds_permute); everything else is done for all elements of a
// 64-byte window at once, and the window's OUTPUT -- at most 1408 bytes -- is assembled in a per-wavefront LDS stage and
// flushed to the block's place in global memory once:
//   * output offsets: exclusive prefix sum of the elements' lengths (DPP row scans), one bounds check for the window;
//   * literals: every payload byte of the window finds its element (the last element start at or below its lane) and its
//     place in the stage with one ds_bpermute, ONE byte write for all literals of the window;
//   * "far" copies (source wholly before this window's output, no overlap with their destination, >= 4 bytes): one LANE per
//     copy, unaligned dword steps global -> stage, all of them together;
//   * "near" copies (source wholly inside the stage): the same steps LDS -> LDS, in dependency rounds;
//   * the other copies (overlapping, shorter than 4, straddling the start of the stage): in element order, a lane per byte;
//   * flush: a dword per lane, stage -> global.  Loads of later windows see it: same-wavefront global_* operations complete
//     in order (tests/test_abi_symbols.py keeps flat_* out of this kernel).
// DESIGN.md 3.2 has the measurements behind each choice.
// Same strictness as the element loop: any invalid element, overrun of the block's output, zero offset or reach before
// the block start makes the block invalid.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
#ifdef SNAPPY_EMU
    for (uint32_t d = 1; d < kWave; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, (int)d);
        if (lane >= d) v += t;
    }
#else
    (void)lane;
    // Hillis-Steele inside each row of 16 lanes (row_shr:1,2,4,8; lanes without a source keep the 0 of `old`), then the row
    // totals travel up: lane 15 of rows 0 and 2 into rows 1 and 3, lane 31 into rows 2 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
#endif
    return v;
}

// Follow the element chain of one window: E collects the lanes where an element starts, s ends at or beyond wlim.
// advv = compressed bytes the element that starts at a lane takes (64 for a start predecode rejected: ends the walk).
__device__ __forceinline__ void k2_chain_walk(uint32_t advv, uint32_t wlim, uint32_t& s, unsigned long long& E)
{
#ifdef SNAPPY_EMU
    do {
        E |= 1ull << s;
        s += (uint32_t)__builtin_amdgcn_readlane((int)advv, (int)s);
    } while (s < wlim);
#else
    uint32_t a;
    // Four elements per trip (a branch that falls through is cheaper than one that is taken).  A full window (wlim == 64, all
    // but a block's last) needs no compare: with t = s - 64 (mod 2^32) "t += a" carries out exactly when s reaches 64, and
    // s_bitset1_b64 / v_readlane_b32 only look at the low six bits of their index, which t and s share.
    if (wlim == 64u) {
        uint32_t t = s - 64u;
        asm volatile(
            "1:\n"
            "  s_bitset1_b64 %[E], %[t]\n"
            "  v_readlane_b32 %[a], %[advv], %[t]\n"
            "  s_add_u32 %[t], %[t], %[a]\n"
            "  s_cbranch_scc1 2f\n"
            "  s_bitset1_b64 %[E], %[t]\n"
            "  v_readlane_b32 %[a], %[advv], %[t]\n"
            "  s_add_u32 %[t], %[t], %[a]\n"
            "  s_cbranch_scc1 2f\n"
            "  s_bitset1_b64 %[E], %[t]\n"
            "  v_readlane_b32 %[a], %[advv], %[t]\n"
            "  s_add_u32 %[t], %[t], %[a]\n"
            "  s_cbranch_scc1 2f\n"
            "  s_bitset1_b64 %[E], %[t]\n"
            "  v_readlane_b32 %[a], %[advv], %[t]\n"
            "  s_add_u32 %[t], %[t], %[a]\n"
            "  s_cbranch_scc0 1b\n"
            "2:\n"
            : [t] "+s"(t), [E] "+s"(E), [a] "=&s"(a)
            : [advv] "v"(advv)
            : "scc");
        s = t + 64u;
        return;
    }
    asm volatile(
        "1:\n"
        "  s_bitset1_b64 %[E], %[s]\n"
        "  v_readlane_b32 %[a], %[advv], %[s]\n"
        "  s_add_u32 %[s], %[s], %[a]\n"
        "  s_cmp_lt_u32 %[s], %[wlim]\n"
        "  s_cbranch_scc1 1b\n"
        : [s] "+s"(s), [E] "+s"(E), [a] "=&s"(a)
        : [advv] "v"(advv), [wlim] "s"(wlim)
        : "scc");
#endif
}

// Output one 64-byte window of compressed data can produce without its last literal's run-on: 22 three-byte copies of 64 bytes
// (starts 0, 3, .. 63) = 1408; rounded up.
constexpr uint32_t kK2StageBytes = 1536;

// levels of doubling in front of K2's serial chain walk: the walk visits every 2^L-th element
#ifndef SNAPPY_K2_WALK_LEVELS
#define SNAPPY_K2_WALK_LEVELS 2
#endif
constexpr uint32_t kK2WalkLevels = SNAPPY_K2_WALK_LEVELS;

// One K2 launch can serve several streams (their own block offsets, output and status arrays; one block size): the
// persistent wavefronts draw GLOBAL block numbers and map them to (stream, block), so a batch has one tail instead of one
// per stream -- the decode-side twin of K1Batch.  Passed by value; the kernel argument segment is read with scalar loads.
struct K2Batch {
    uint32_t count;
    uint32_t first_block[kMaxBatch + 1];   // first_block[count] = total number of blocks
    const uint8_t* stream[kMaxBatch];
    uint64_t stream_len[kMaxBatch];
    const uint64_t* stream_len_dev[kMaxBatch];   // when not null: the length is read from the device (no host round trip)
    const uint64_t* block_offsets[kMaxBatch];
    uint64_t total_len[kMaxBatch];
    uint8_t* out[kMaxBatch];
    uint32_t* status[kMaxBatch];
};

__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) void decompress_blocks_kernel(const K2Batch w, uint32_t block_size,
                                                                                                    uint32_t* next_block)
{
    __shared__ __attribute__((aligned(16))) uint8_t k2_stage_mem[kK2StageBytes];   // one window's output
    lds_bytes_t stage = (lds_bytes_t)k2_stage_mem;
    const uint32_t lane = threadIdx.x;
    const uint32_t num_blocks = w.first_block[w.count];

    // Persistent: wavefronts draw blocks from *next_block (zeroed per launch).
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
        uint8_t* win = dst;    // the block's output: written in place, back-references read from there

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
        WindowLoad next2 = {0, 64}; // batch form: the 64 bytes after those, so that W1 has arrived when a literal runs on into it
        bool have_window = false;
        // batch form: a window wholly inside the stream (all but the last windows of a stream's last blocks) is a plain load at a
        // 32-bit offset from the block's first element; the others take window_issue()'s clamped address and tail shift
        const uint32_t avail32 = avail > 0xffffff00ull ? 0xffffff00u : (uint32_t)avail;
        bool next_tail = true, next2_tail = true;
        auto issue = [&](uint32_t base, bool& tail) -> WindowLoad {
            WindowLoad r;
            if (base + 72u <= avail32) {
                r.raw = ld64(src + (base + lane));
                r.shift = 0;
                tail = false;
            } else {
                r = window_issue(src, (uint64_t)base + lane, avail);
                tail = true;
            }
            return r;
        };
        while (st == kBlockOk && cp < csz) {                             // one iteration per 64-byte window
            // the window registers rotate at the END of a window (see there); only the first window of a block and the one
            // after a long literal load afresh: all three requests go out before the first is awaited
            if (!have_window) {
                g = cp & ~63u;
                bool cur_tail;
                const WindowLoad cur = issue(g, cur_tail);
                next = issue(g + 64u, next_tail);
                next2 = issue(g + 128u, next2_tail);
                w0 = cur_tail ? window_value(cur) : cur.raw;
                have_window = true;
            }
            uint32_t offv = 0;
            const uint32_t wend = (csz < g + 64) ? csz : g + 64;
            {
                // ================= the whole window at once =================
                const uint32_t wlim = wend - g;
                uint32_t e_type, e_hdr, e_len, e_consumed;
                unsigned long long REJ;
                predecode_window(w0, g + lane, csz, e_type, e_hdr, e_len, offv, e_consumed, REJ);
                const uint32_t advv = __builtin_amdgcn_inverse_ballot_w64(REJ) ? 64u : e_consumed;
                uint32_t s = cp - g;
                unsigned long long E = 0;
                // The serial walk visits every 2^L-th element (L = kK2WalkLevels): jump[k] = the compressed bytes of this element
                // and its next 2^k - 1 successors, built by doubling (jump[k] = jump[k-1] + jump[k-1] of the start 2^(k-1)
                // elements ahead: one ds_bpermute per level; nothing is added for a start beyond the window, so the walk still
                // ends on the first start at or beyond wlim).  The starts in between are filled in afterwards, level by level:
                // every lane of E whose 2^k-th successor starts inside the window pushes a 1 to it (ds_permute, the forward
                // form; lanes of E lie on one chain, so their targets are distinct); the other lanes push to lane 0, which
                // cannot be anybody's successor.
                {
                    uint32_t jump[kK2WalkLevels + 1], tgt[kK2WalkLevels + 1];
                    jump[0] = advv;
                    tgt[0] = lane + advv;
#pragma unroll
                    for (uint32_t k = 1; k <= kK2WalkLevels; ++k) {
                        const uint32_t a_n = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tgt[k - 1] << 2), (int)jump[k - 1]);
                        jump[k] = jump[k - 1] + (tgt[k - 1] < wlim ? a_n : 0u);
                        tgt[k] = lane + jump[k];
                    }
                    k2_chain_walk(jump[kK2WalkLevels], wlim, s, E);
#pragma unroll
                    for (uint32_t k = kK2WalkLevels; k-- > 0;) {
                        const bool pusher = __builtin_amdgcn_inverse_ballot_w64(E) && tgt[k] < wlim;
                        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(pusher ? tgt[k] << 2 : 0u), pusher ? 1 : 0);
                        E |= __ballot(got != 0) & ~1ull;
                    }
                }
                if (E & REJ) {                                           // an element predecode rejected
                    st = kBlockInvalid;
                    break;
                }
                const uint32_t mylen = __builtin_amdgcn_inverse_ballot_w64(E) ? e_len : 0u;
                const uint32_t incl = wave_inclusive_scan(mylen, lane);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                const uint32_t rel = incl - mylen;                       // this lane's element inside the window's output
                const uint32_t dstp = op + rel;                          // ... and inside the block's
                const unsigned long long COPY = E & __ballot(e_type != 0);
                if (op + total > out_len || (COPY & (__ballot(offv == 0) | __ballot(offv > dstp)))) {   // strict, cf. :167-173
                    st = kBlockInvalid;
                    break;
                }
                // ================= staged in LDS: the window's output is assembled in k2_stage and flushed once =================
                // Output bytes [op, op + staged) of this window live in stage[0, staged) until the flush; a back-reference into
                // them costs an LDS round trip instead of a trip to L2.  Only the part of a last literal that runs on beyond the
                // 64 window bytes ("spill") bypasses the stage: nothing in this window can refer to it.
                // (the walk ended right behind the window's last element: a literal there ends at pe = s, and only s > 64 can spill)
                uint32_t le = 0, ps = 0;
                const uint32_t pe = s;
                bool spills = false;
                if (s > 64u) {
                    le = 63u - (uint32_t)__builtin_clzll(E);
                    if ((uint32_t)__builtin_amdgcn_readlane((int)e_type, (int)le) == 0) {
                        ps = le + (uint32_t)__builtin_amdgcn_readlane((int)e_hdr, (int)le);
                        spills = true;
                    }
                }
                const uint32_t staged = total - (spills ? pe - (ps > 64u ? ps : 64u) : 0u);   // a tag in the last lanes: payload from ps > 64
                if (staged > kK2StageBytes) {                            // cannot happen: 22 copies of 64 bytes are the most 64 bytes can hold
                    st = kBlockInvalid;
                    break;
                }
                // ---- copies, first part.  "Steppable": does not overlap its own destination and is at least 4 bytes long, so ONE
                //      lane can do it in unaligned dword steps.  Far ones (source wholly before this window's output) load from
                //      global memory, all of them together: the loads of their first 8 bytes (most copies of a text end there)
                //      go out HERE, so that they travel while the literal bytes are placed.  Offsets beyond a copy's length are
                //      clamped to its last dword: those steps reload and rewrite that dword (same bytes, same place), so the
                //      loads and stores need no predicate of their own ----
                const unsigned long long STEP = COPY & __ballot(e_len >= 4u) & __ballot(offv >= e_len);
                const unsigned long long FAR = STEP & __ballot(offv >= rel + e_len);
                const unsigned long long NEAR = STEP & __ballot(offv <= rel);
                const bool is_far = __builtin_amdgcn_inverse_ballot_w64(FAR);
                const uint32_t last = e_len - 4u;
                const uint32_t far_o1 = 4u < last ? 4u : last;
                const uint8_t* sbase = win;                              // uniform base + 32-bit lane offset: no 64-bit address arithmetic
                const uint32_t so = dstp - offv;
                uint32_t far_v0 = 0, far_v1 = 0;
                if (is_far) {
                    far_v0 = ld32(sbase + so);
                    far_v1 = ld32(sbase + (so + far_o1));
                }
                __builtin_amdgcn_sched_barrier(0);                       // keep the two loads up here
                // ---- literals: a payload byte belongs to the last element that starts at or below its lane ----
                {
                    const unsigned long long below = E & (~0ull >> (63u - lane));         // element starts at or below this lane
                    const bool any = below != 0;
                    const uint32_t em = 63u - (uint32_t)__builtin_clzll(below | 1ull);     // branch-free: 0 when there is none
                    const uint32_t packed = rel | (e_hdr << 16) | (e_type << 20);             // offsets inside a window are < 4096
                    const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(em << 2), (int)packed);
                    const uint32_t pstart = em + ((pk >> 16) & 7u);
                    if (any && ((pk >> 20) & 3u) == 0 && lane >= pstart && lane < wlim) stage[(pk & 0xffffu) + lane - pstart] = (uint8_t)w0;
                }
                // the spill: the next 64 bytes come from the prefetch registers, anything beyond straight from memory
                if (spills) {
                    const uint32_t dbase = (uint32_t)__builtin_amdgcn_readlane((int)dstp, (int)le) - ps;   // byte at g + q goes to dbase + q
                    WindowLoad nx = next;
                    SNAPPY_PIN(nx.shift);                                // first use of the prefetch: wait here, not earlier
                    const uint64_t w1 = next_tail ? window_value(nx) : nx.raw;
                    const uint32_t q = 64u + lane;
                    if (q >= ps && q < pe) win[dbase + q] = (uint8_t)w1;
                    if (pe > 128u) {
                        const uint8_t* __restrict__ p = src + g + 128u;
                        uint8_t* d = win + (uint32_t)(dbase + 128u);     // dbase may be "negative" (mod 2^32): add before widening
                        const uint32_t rest = pe - 128u;
                        uint32_t i = 4 * lane;
                        for (; i + 4 <= rest; i += 4 * kWave) st32(d + i, ld32(p + i));
                        for (; i < rest; ++i) d[i] = p[i];
                    }
                }
                // ---- copies, second part.  The far copies land in the stage.  Near ones (source wholly inside the stage) go
                //      LDS -> LDS in rounds: the first copy still to do, with every other near one whose source ends before that
                //      copy's destination -- all output below it is complete.  The rest (overlapping, i.e. :174-181 replicating
                //      the last `off` bytes; shorter than 4; source straddling the start of the stage) go one at a time, a lane
                //      per byte ----
                {
                    const uint32_t src_end = rel - offv + e_len;         // meaningful for near copies only
                    unsigned long long rem = COPY & ~FAR;
                    lds_bytes_t dp = stage + rel;
                    if (is_far) {
                        lds_st32u(dp, far_v0);
                        lds_st32u(dp + far_o1, far_v1);
                        for (uint32_t base = 8u; base < e_len; base += 16u) {
                            const uint32_t o0 = base < last ? base : last, o1 = base + 4u < last ? base + 4u : last,
                                           o2 = base + 8u < last ? base + 8u : last, o3 = base + 12u < last ? base + 12u : last;
                            const uint32_t v0 = ld32(sbase + (so + o0)), v1 = ld32(sbase + (so + o1)), v2 = ld32(sbase + (so + o2)),
                                           v3 = ld32(sbase + (so + o3));
                            lds_st32u(dp + o0, v0);
                            lds_st32u(dp + o1, v1);
                            lds_st32u(dp + o2, v2);
                            lds_st32u(dp + o3, v3);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    while (rem) {
                        const uint32_t f = (uint32_t)__builtin_ctzll(rem);
                        const uint32_t fd = (uint32_t)__builtin_amdgcn_readlane((int)rel, (int)f);
                        if ((NEAR >> f) & 1ull) {
                            const unsigned long long ready = rem & NEAR & __ballot(src_end <= fd);
                            if (__builtin_amdgcn_inverse_ballot_w64(ready)) {
                                lds_bytes_t sp = stage + (rel - offv);
                                {
                                    const uint32_t o1 = 4u < last ? 4u : last;
                                    const uint32_t v0 = lds_ld32u(sp), v1 = lds_ld32u(sp + o1);
                                    lds_st32u(dp, v0);
                                    lds_st32u(dp + o1, v1);
                                }
                                for (uint32_t base = 8u; base < e_len; base += 16u) {
                                    const uint32_t o0 = base < last ? base : last, o1 = base + 4u < last ? base + 4u : last,
                                                   o2 = base + 8u < last ? base + 8u : last, o3 = base + 12u < last ? base + 12u : last;
                                    const uint32_t v0 = lds_ld32u(sp + o0), v1 = lds_ld32u(sp + o1), v2 = lds_ld32u(sp + o2),
                                                   v3 = lds_ld32u(sp + o3);
                                    lds_st32u(dp + o0, v0);
                                    lds_st32u(dp + o1, v1);
                                    lds_st32u(dp + o2, v2);
                                    lds_st32u(dp + o3, v3);
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                            rem &= ~ready;
                            continue;
                        }
                        const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)e_len, (int)f);
                        const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)offv, (int)f);
                        uint32_t src_idx = lane;
                        if (off < len) {                                 // overlap: lane % off (lane < 64, off < 64)
                            const uint32_t q = (lane * kRecip16[off]) >> 16;
                            src_idx = lane - q * off;
                        }
                        if (lane < len) {
                            const uint32_t at_stage = fd + src_idx;      // source byte = stage[at_stage - off] when that is inside the stage
                            const uint8_t v = at_stage >= off ? stage[at_stage - off] : win[op + at_stage - off];
                            stage[fd + lane] = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                        rem &= rem - 1;
                    }
                }
                // ---- the next window.  Normally the following 64 bytes: rotate the window registers HERE, before the flush, and
                //      send the new prefetch into the register that just became free.  (Rotating at the top of the next
                //      iteration makes the compiler copy a register an outstanding load will write; it then waits at the back
                //      edge for EVERY outstanding vector memory operation, vmcnt(0), the flush stores included -- a full write
                //      round trip per window.  Here the two prefetches are a window old, and only stores are in flight at the
                //      back edge.)  After a literal that ran on beyond the next window the block loads afresh ----
                const uint32_t flush_at = op;
                op += total;
                cp = g + s;
                if (cp < g + 128u) {
                    SNAPPY_PIN(next.raw);
                    SNAPPY_PIN(next2.raw);
                    w0 = next_tail ? window_value(next) : next.raw;
                    next = next2;
                    next_tail = next2_tail;
                    g += 64;
                    next2 = issue(g + 128u, next2_tail);
                } else {
                    have_window = false;
                }
                // ---- flush: stage[0, staged) -> the block's output at op, a dword per lane (the last one clamped back) ----
                if (staged >= 4u) {
                    {                                                    // the first 256 bytes: all there is in most windows
                        const uint32_t i = 4u * lane;
                        const uint32_t o = i + 4u <= staged ? i : staged - 4u;
                        if (i < staged) st32(win + flush_at + o, lds_ld32u(stage + o));
                    }
                    for (uint32_t i = 4u * (lane + kWave); i < staged; i += 4u * kWave) {
                        const uint32_t o = i + 4u <= staged ? i : staged - 4u;
                        st32(win + flush_at + o, lds_ld32u(stage + o));
                    }
                } else if (lane < staged) {
                    win[flush_at + lane] = stage[lane];
                }
                __builtin_amdgcn_wave_barrier();
                continue;
            }
        }
        if (st == kBlockOk && (op != out_len || cp != csz)) st = kBlockInvalid;

        if (lane == 0) status[b] = st;
        __syncthreads();
    }
}


}  // namespace snappy_hip

