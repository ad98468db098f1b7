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
#include <stdint.h>

namespace snappy_hip {

constexpr uint32_t kWave = 64;
constexpr uint32_t kMaxTableEntries = 16384;   // snappy_compress.c:16-17
constexpr uint32_t kHashMul = 0x1e35a7bdu;     // snappy_compress.c:163
constexpr uint32_t kInputMargin = 15;          // snappy_compress.c:299

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
// Input-access policies for K1.  The parse reads the block at (a) the cursor, sequentially, and
// (b) hash-table candidates, randomly inside [0, cursor).  Where those bytes live decides the
// latency of every serial step, so the kernel is templated on it.
// ---------------------------------------------------------------------------

// Block read straight from HBM/L2 with vector loads; LDS holds only the hash table (5 blocks/CU).
struct InputGlobalVector {
    const uint8_t* __restrict__ blk;
    __device__ __forceinline__ void stage(const uint8_t* __restrict__ b, uint32_t, uint32_t, uint8_t*) { blk = b; }
    __device__ __forceinline__ uint32_t u32(uint32_t p) const { return uld32(blk + p); }
    __device__ __forceinline__ uint64_t u64(uint32_t p) const { return uld64(blk + p); }
    __device__ __forceinline__ const uint8_t* bytes() const { return blk; }
};

// Same, but the wave-uniform reads go through the scalar cache (s_load_dwordx2/x4 on aligned dwords +
// 64-bit shift), which returns into SGPRs directly.  `base16` is the 16-byte aligned container base.
struct InputGlobalScalar {
    const uint8_t* __restrict__ blk;
    const uint8_t* __restrict__ base16;
    uint64_t start;
    __device__ __forceinline__ void stage(const uint8_t* __restrict__ b, uint32_t, uint32_t, uint8_t*) { blk = b; }
    __device__ __forceinline__ uint32_t u32(uint32_t p) const
    {
        const uint64_t a = start + p;
        const uint32_t* w = static_cast<const uint32_t*>(__builtin_assume_aligned(base16 + (a & ~3ull), 4));
        const uint64_t two = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        return (uint32_t)(two >> (8 * (uint32_t)(a & 3)));
    }
    __device__ __forceinline__ uint64_t u64(uint32_t p) const
    {
        const uint64_t a = start + p;
        const uint32_t* w = static_cast<const uint32_t*>(__builtin_assume_aligned(base16 + (a & ~3ull), 4));
        const uint32_t sh = 8 * (uint32_t)(a & 3);
        const uint64_t lo = ((uint64_t)w[0] | ((uint64_t)w[1] << 32)) >> sh;
        const uint64_t hi = ((uint64_t)w[1] | ((uint64_t)w[2] << 32)) >> sh;
        return (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32);
    }
    __device__ __forceinline__ const uint8_t* bytes() const { return blk; }
};

// Block staged into LDS once (coalesced 16 B/lane), every later read is an LDS read.
// 64 KiB of LDS per 32 KiB block (table + input): 2 blocks/CU, but ~10x lower read latency.
struct InputLds {
    const uint8_t* lds;
    __device__ __forceinline__ void stage(const uint8_t* __restrict__ b, uint32_t n, uint32_t lane, uint8_t* buf)
    {
        uint32_t i = 16 * lane;
        for (; i + 16 <= n; i += 16 * kWave) {
            uint4 v;
            __builtin_memcpy(&v, b + i, 16);
            *reinterpret_cast<uint4*>(buf + i) = v;
        }
        for (; i < n; ++i) buf[i] = b[i];     // one lane, < 16 bytes
        lds = buf;
    }
    __device__ __forceinline__ uint32_t u32(uint32_t p) const { return uld32(lds + p); }
    __device__ __forceinline__ uint64_t u64(uint32_t p) const { return uld64(lds + p); }
    __device__ __forceinline__ const uint8_t* bytes() const { return lds; }
};

// ---------------------------------------------------------------------------
// K1: compress.  One wavefront per block (grid-stride).
// ---------------------------------------------------------------------------
template <class Input>
__device__ __forceinline__ void compress_one_block(Input& in, const uint8_t* __restrict__ blk_global, uint32_t n,
                                                   uint8_t* __restrict__ dst, uint16_t* table, uint8_t* stage_buf,
                                                   uint32_t lane, uint32_t* __restrict__ block_bytes_out)
{
    // get_hash_table, snappy_compress.c:139-146 (+ shift, :288)
    const uint32_t ts = table_entries_for(n);
    const uint32_t shift = (uint32_t)__builtin_clz(ts) + 1;   // 32 - log2(ts)
    {
        uint4* t = reinterpret_cast<uint4*>(table);
        for (uint32_t i = lane; i < ts / 8; i += kWave) t[i] = make_uint4(0, 0, 0, 0);
    }
    in.stage(blk_global, n, lane, stage_buf);
    __syncthreads();
    const uint8_t* blk = in.bytes();   // per-lane (vector) reads: match extension

    uint32_t op = 4;          // :291 room for the u32 size prefix
    uint32_t next_emit = 0;   // :298

    if (n >= kInputMargin) {  // :301
        const uint32_t limit = n - kInputMargin;
        uint32_t ip = 1;      // :305
        uint32_t cur = in.u32(ip);        // bytes at ip
        for (;;) {
            // ---- step 1: scan for a 4-byte match (:333-348) ----
            uint32_t skip = 32;
            uint32_t cand;
            bool out_of_input = false;
            for (;;) {
                const uint32_t h = (cur * kHashMul) >> shift;
                const uint32_t next_ip = ip + (skip++ >> 5);
                if (next_ip > limit) {          // :342-343, before touching the table
                    out_of_input = true;
                    break;
                }
                const uint32_t nxt = in.u32(next_ip);
                cand = uni((uint32_t)table[h]);
                if (lane == 0) table[h] = (uint16_t)ip;
                __builtin_amdgcn_wave_barrier();
                if (cur == in.u32(cand)) break;
                ip = next_ip;
                cur = nxt;
            }
            if (out_of_input) break;

            // ---- step 2: literal run [next_emit, ip) (:355); payload copied from global memory ----
            op = emit_literal(dst, op, blk_global + next_emit, ip - next_emit, lane);

            // ---- step 3: copy chain (:370-398) ----
            bool again;
            bool done = false;
            uint32_t tail = 0;     // le32(ip+1) after the chain, for the next scan
            do {
                const uint32_t base = ip;
                const uint32_t matched = 4 + match_extend(blk, cand + 4, ip + 4, n, lane);
                ip += matched;
                op = emit_copy(dst, op, base - cand, matched, lane);
                next_emit = ip;
                if (ip >= limit) {              // :388-389
                    done = true;
                    break;
                }
                const uint64_t w = in.u64(ip - 1);               // bytes ip-1 .. ip+6
                const uint32_t prev_bytes = (uint32_t)w;
                const uint32_t here = (uint32_t)(w >> 8);
                tail = (uint32_t)(w >> 16);
                const uint32_t hp = (prev_bytes * kHashMul) >> shift;
                const uint32_t hc = (here * kHashMul) >> shift;
                if (lane == 0) table[hp] = (uint16_t)(ip - 1);   // :391-392
                __builtin_amdgcn_wave_barrier();
                cand = uni((uint32_t)table[hc]);                 // :394-395
                if (lane == 0) table[hc] = (uint16_t)ip;         // :397
                __builtin_amdgcn_wave_barrier();
                again = (here == in.u32(cand));                  // :396,:398
            } while (again);
            if (done) break;

            ++ip;                                                // :400-401
            cur = tail;
        }
    }

    // emit_remainder (:405-410) and the size prefix (:412)
    if (next_emit < n) op = emit_literal(dst, op, blk_global + next_emit, n - next_emit, lane);
    if (lane == 0) {
        st32(dst, op - 4);
        *block_bytes_out = op;
    }
    __syncthreads();   // table (and the staged block) are rewritten by the next iteration
}

enum CompressVariant { kVariantGlobalVector = 0, kVariantGlobalScalar = 1, kVariantLdsInput = 2 };

template <int kVariant>
__global__ __launch_bounds__(64) void compress_blocks_kernel(const uint8_t* __restrict__ in, uint64_t in_len,
                                                             uint32_t block_size, uint8_t* __restrict__ slots,
                                                             uint32_t slot_stride, uint32_t* __restrict__ block_bytes,
                                                             uint32_t num_blocks)
{
    __shared__ __attribute__((aligned(16))) uint16_t table[kMaxTableEntries];
    HIP_DYNAMIC_SHARED(uint8_t, stage_buf)   // kVariantLdsInput: block_size rounded up to 16 (+16); else unused
    const uint32_t lane = threadIdx.x;

    for (uint32_t b = blockIdx.x; b < num_blocks; b += gridDim.x) {
        const uint64_t start = (uint64_t)b * block_size;
        const uint64_t left = in_len - start;
        const uint32_t n = (left < block_size) ? (uint32_t)left : block_size;
        const uint8_t* __restrict__ blk = in + start;
        uint8_t* __restrict__ dst = slots + (uint64_t)b * slot_stride;
        if constexpr (kVariant == kVariantGlobalScalar) {
            InputGlobalScalar src{blk, in, start};
            compress_one_block(src, blk, n, dst, table, stage_buf, lane, block_bytes + b);
        } else if constexpr (kVariant == kVariantLdsInput) {
            InputLds src{stage_buf};
            compress_one_block(src, blk, n, dst, table, stage_buf, lane, block_bytes + b);
        } else {
            InputGlobalVector src{blk};
            compress_one_block(src, blk, n, dst, table, stage_buf, lane, block_bytes + b);
        }
    }
}

// ---------------------------------------------------------------------------
// scan + gather: slots -> contiguous framed stream
// ---------------------------------------------------------------------------

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

__global__ __launch_bounds__(64) void index_streams_kernel(const StreamDesc* __restrict__ descs, uint32_t count)
{
    const uint32_t s = blockIdx.x;
    if (s >= count) return;
    const StreamDesc d = descs[s];
    const uint32_t lane = threadIdx.x;
    uint64_t at = d.header_len;
    uint32_t status = kBlockOk;
    uint32_t i = 0;
    for (; i < d.num_blocks; ++i) {
        if (at + 4 > d.stream_len) {
            status = kBlockInvalid;
            break;
        }
        if (lane == 0) d.block_offsets[i] = at;
        at += 4 + (uint64_t)uld32(d.stream + at);
    }
    if (status == kBlockOk && at != d.stream_len) status = kBlockInvalid;
    if (lane == 0) {
        d.result[0] = status;
        d.result[1] = i;
    }
}

// ---------------------------------------------------------------------------
// K2: decompress.  One wavefront per block; the decoded block is staged in LDS so that
// back-references are LDS reads, then written out with coalesced stores.
// ---------------------------------------------------------------------------

// floor(x / d) for x < 64, 1 <= d < 64:  (x * kRecip16[d]) >> 16  with kRecip16[d] = 65536/d + 1
__constant__ uint32_t kRecip16[64] = {
        0, 65537, 32769, 21846, 16385, 13108, 10923,  9363,  8193,  7282,  6554,  5958,  5462,  5042,  4682,  4370,
     4097,  3856,  3641,  3450,  3277,  3121,  2979,  2850,  2731,  2622,  2521,  2428,  2341,  2260,  2185,  2115,
     2049,  1986,  1928,  1873,  1821,  1772,  1725,  1681,  1639,  1599,  1561,  1525,  1490,  1457,  1425,  1395,
     1366,  1338,  1311,  1286,  1261,  1237,  1214,  1192,  1171,  1150,  1130,  1111,  1093,  1075,  1058,  1041};

// safe uniform load of up to 8 bytes at stream[ip..], zero-filled past `end`
__device__ __forceinline__ uint64_t uld64_clamped(const uint8_t* __restrict__ s, uint64_t ip, uint64_t end)
{
    if (ip + 8 <= end) return uld64(s + ip);
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8 && ip + k < end; ++k) v |= (uint64_t)s[ip + k] << (8 * k);
    return (uint64_t)uni((uint32_t)v) | ((uint64_t)uni((uint32_t)(v >> 32)) << 32);
}

__global__ __launch_bounds__(64) void decompress_blocks_kernel(const uint8_t* __restrict__ stream, uint64_t stream_len,
                                                               const uint64_t* __restrict__ block_offsets,
                                                               uint64_t total_len, uint32_t block_size,
                                                               uint8_t* __restrict__ out, uint32_t* __restrict__ status,
                                                               uint32_t num_blocks)
{
    HIP_DYNAMIC_SHARED(uint8_t, win)   // block_size rounded up to 16; dynamic LDS starts 16-byte aligned
    const uint32_t lane = threadIdx.x;

    for (uint32_t b = blockIdx.x; b < num_blocks; b += gridDim.x) {
        const uint64_t ostart = (uint64_t)b * block_size;
        const uint64_t oleft = total_len - ostart;
        const uint32_t out_len = (oleft < block_size) ? (uint32_t)oleft : block_size;
        uint8_t* __restrict__ dst = out + ostart;

        uint32_t st = kBlockOk;
        uint32_t op = 0;
        const uint64_t at = block_offsets[b];
        uint64_t ip = at + 4, end = ip;
        if (at + 4 > stream_len) {
            st = kBlockInvalid;
        } else {
            end = ip + (uint64_t)uld32(stream + at);           // snappy_decompress.c:229-230
            if (end > stream_len) st = kBlockInvalid;
        }

        while (st == kBlockOk && ip < end) {                    // :232
            const uint64_t w = uld64_clamped(stream, ip, end);  // tag + up to 7 following bytes
            const uint32_t tag = (uint32_t)w & 0xff;
            const uint32_t type = tag & 3;
            if (type == 0) {                                    // literal, :244-256
                uint32_t len = (tag >> 2) + 1, hdr = 1;
                if (len > 60) {                                 // :64-74
                    const uint32_t nb = len - 60;
                    hdr = 1 + nb;
                    len = ((uint32_t)(w >> 8) & (0xffffffffu >> (32 - 8 * nb))) + 1;
                }
                if (ip + hdr + len > end || op + len > out_len || len == 0) {
                    st = kBlockInvalid;
                    break;
                }
                if (hdr + len <= 8) {
                    // payload already sits in w
                    if (lane < len) win[op + lane] = (uint8_t)(w >> (8 * (hdr + lane)));
                } else if (len <= kWave) {
                    if (lane < len) win[op + lane] = stream[ip + hdr + lane];
                } else {
                    const uint8_t* __restrict__ src = stream + ip + hdr;
                    uint32_t i = 4 * lane;
                    for (; i + 4 <= len; i += 4 * kWave) st32(win + op + i, ld32(src + i));
                    for (; i < len; ++i) win[op + i] = src[i];
                }
                __builtin_amdgcn_wave_barrier();
                ip += hdr + len;
                op += len;
                continue;
            }
            uint32_t len, off, hdr;
            if (type == 1) {                                    // :264-266, :83-88
                len = ((tag >> 2) & 7) + 4;
                off = ((tag >> 5) << 8) | ((uint32_t)(w >> 8) & 0xff);
                hdr = 2;
            } else if (type == 2) {                             // :271-273, :97-109
                len = (tag >> 2) + 1;
                off = (uint32_t)(w >> 8) & 0xffff;
                hdr = 3;
            } else {                                            // :278-280, :118-133
                len = (tag >> 2) + 1;
                off = (uint32_t)(w >> 8);
                hdr = 5;
            }
            // strict: source must lie inside this block's own output (cf. :167-173)
            if (ip + hdr > end || off == 0 || off > op || op + len > out_len) {
                st = kBlockInvalid;
                break;
            }
            // :174-181 forward byte copy == periodic replication of the last `off` bytes
            {
                uint32_t src_idx = lane;
                if (off < len) {                                // overlap: lane % off
                    const uint32_t q = (lane * kRecip16[off]) >> 16;
                    src_idx = lane - q * off;
                }
                uint8_t v = 0;
                if (lane < len) v = win[op - off + src_idx];
                __builtin_amdgcn_wave_barrier();
                if (lane < len) win[op + lane] = v;
                __builtin_amdgcn_wave_barrier();
            }
            ip += hdr;
            op += len;
        }
        if (st == kBlockOk && (op != out_len || ip != end)) st = kBlockInvalid;

        // write-out: LDS -> global, 16 B per lane when the destination allows it
        __syncthreads();
        if (st == kBlockOk) {
            if ((((uintptr_t)dst) & 15) == 0) {
                const uint32_t body = out_len & ~15u;
                const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(win);
                uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
                for (uint32_t i = lane; i < body / 16; i += kWave) d4[i] = w4[i];
                for (uint32_t i = body + lane; i < out_len; i += kWave) dst[i] = win[i];
            } else {
                for (uint32_t i = lane; i < out_len; i += kWave) dst[i] = win[i];
            }
        }
        if (lane == 0) status[b] = st;
        __syncthreads();
    }
}

}  // namespace snappy_hip
