// Fiber-based lockstep wave emulator + C entry points that drive the real kernel source on the CPU.
// Test infrastructure only (see tests/emu/hip/hip_runtime.h).
#include <hip/hip_runtime.h>   // resolves to tests/emu/hip/hip_runtime.h via -I
#include <sys/mman.h>
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "snappy_kernels.hpp"

#ifndef EMU_K1_AHEAD
#define EMU_K1_AHEAD 64
#endif

namespace emu {

constexpr int kStack = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    std::vector<uint8_t> stack;
    Dim tid;
    bool done = false;
    bool started = false;
};

struct WaveSlot {
    uint64_t val[64];
    uint32_t arg[64];
    uint64_t res[2][64];
    int arrived = 0;
    int active = 0;
    uint32_t gen = 0;
    int site = -1;
    Op op = OP_BARRIER;
};

static std::vector<Fiber> g_fibers;
static std::vector<WaveSlot> g_waves;
static ucontext_t g_sched;
static int g_cur = -1;
static Dim g_bidx, g_gdim, g_bdim;
static std::function<void()> g_body;
static int g_wg_active = 0, g_sync_arrived = 0, g_sync_site = -1;
static uint32_t g_sync_gen = 0;
static uint8_t g_dyn_lds[98304 + 64] __attribute__((aligned(64)));

const Dim& tidx() { return g_fibers[g_cur].tid; }
const Dim& bidx() { return g_bidx; }
const Dim& gdim() { return g_gdim; }
const Dim& bdim() { return g_bdim; }
void* dynamic_lds() { return g_dyn_lds; }

static void yield() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }

static void complete(WaveSlot& w)
{
    uint64_t* r = w.res[w.gen & 1];
    // participating lanes are exactly those that deposited (active ones)
    int first = -1;
    uint64_t ballot = 0;
    for (int l = 0; l < 64; ++l)
        if (w.arg[l] != 0xdeadbeefu) {
            if (first < 0) first = l;
            if (w.op == OP_BALLOT && w.val[l]) ballot |= 1ull << l;
        }
    uint64_t pushed[64] = {0};
    if (w.op == OP_PERMUTE)
        for (int l = 0; l < 64; ++l)
            if (w.arg[l] != 0xdeadbeefu) pushed[w.arg[l] & 63] = w.val[l];
    for (int l = 0; l < 64; ++l) {
        if (w.arg[l] == 0xdeadbeefu) continue;
        switch (w.op) {
        case OP_PERMUTE: r[l] = pushed[l]; break;
        case OP_FIRSTLANE: r[l] = w.val[first]; break;
        case OP_BALLOT: r[l] = ballot; break;
        case OP_READLANE: r[l] = w.val[w.arg[l] & 63]; break;
        case OP_SHFL_UP: r[l] = (l >= (int)w.arg[l]) ? w.val[l - w.arg[l]] : w.val[l]; break;
        case OP_BARRIER: r[l] = 0; break;
        }
    }
    for (int l = 0; l < 64; ++l) w.arg[l] = 0xdeadbeefu;
    w.arrived = 0;
    w.site = -1;
    w.gen++;
}

uint64_t collective(Op op, uint64_t value, uint32_t arg, int site)
{
    const uint32_t t = g_fibers[g_cur].tid.x;
    WaveSlot& w = g_waves[t >> 6];
    const int lane = t & 63;
    if (w.arrived == 0) {
        w.site = site;
        w.op = op;
    } else if (w.site != site || w.op != op) {
        fprintf(stderr, "emu: divergent collective: lane %d at line %d, wave waiting at line %d\n", lane, site, w.site);
        abort();
    }
    w.val[lane] = value;
    w.arg[lane] = arg;
    w.arrived++;
    const uint32_t gen = w.gen;
    if (w.arrived == w.active)
        complete(w);
    else
        while (w.gen == gen) yield();
    return w.res[gen & 1][lane];
}

void syncthreads(int site)
{
    if (g_sync_arrived == 0)
        g_sync_site = site;
    else if (g_sync_site != site) {
        fprintf(stderr, "emu: divergent __syncthreads (line %d vs %d)\n", site, g_sync_site);
        abort();
    }
    const uint32_t gen = g_sync_gen;
    if (++g_sync_arrived == g_wg_active) {
        g_sync_arrived = 0;
        g_sync_gen++;
    } else
        while (g_sync_gen == gen) yield();
}

static void fiber_main()
{
    g_body();
    Fiber& f = g_fibers[g_cur];
    f.done = true;
    WaveSlot& w = g_waves[f.tid.x >> 6];
    w.active--;
    g_wg_active--;
    if (w.arrived > 0 && w.arrived == w.active) complete(w);
    if (g_sync_arrived > 0 && g_sync_arrived == g_wg_active) {
        g_sync_arrived = 0;
        g_sync_gen++;
    }
    swapcontext(&f.ctx, &g_sched);
}

void launch(uint32_t grid, uint32_t block, const std::function<void()>& body)
{
    g_body = body;
    g_gdim = Dim{grid, 1, 1};
    g_bdim = Dim{block, 1, 1};
    if (g_fibers.size() < block) g_fibers.resize(block);
    for (uint32_t b = 0; b < grid; ++b) {
        g_bidx = Dim{b, 0, 0};
        g_waves.assign((block + 63) / 64, WaveSlot());
        for (auto& w : g_waves)
            for (int l = 0; l < 64; ++l) w.arg[l] = 0xdeadbeefu;
        g_wg_active = (int)block;
        g_sync_arrived = 0;
        for (uint32_t t = 0; t < block; ++t) {
            Fiber& f = g_fibers[t];
            f.tid = Dim{t, 0, 0};
            f.done = false;
            if (f.stack.empty()) f.stack.resize(kStack);
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack.data();
            f.ctx.uc_stack.ss_size = f.stack.size();
            f.ctx.uc_link = &g_sched;
            makecontext(&f.ctx, fiber_main, 0);
            g_waves[t >> 6].active++;
        }
        int remaining = (int)block;
        // EMU_SHUFFLE=<seed>: run the fibers of a pass in a random order (and whole wavefronts in bursts), so that the
        // wavefronts of a workgroup interleave differently from run to run -- a stress for code whose wavefronts talk to
        // each other through LDS (the two-wavefront K1); the default order is deterministic round-robin
        static const char* shuffle_env = getenv("EMU_SHUFFLE");
        static uint64_t rng_state = shuffle_env ? (uint64_t)strtoull(shuffle_env, nullptr, 10) * 0x9e3779b97f4a7c15ull + 1 : 0;
        std::vector<uint32_t> order(block);
        for (uint32_t t = 0; t < block; ++t) order[t] = t;
        while (remaining > 0) {
            int progressed = 0;
            bool burst = false;
            if (shuffle_env) {
                auto next = [&]() {
                    rng_state ^= rng_state << 13;
                    rng_state ^= rng_state >> 7;
                    rng_state ^= rng_state << 17;
                    return rng_state;
                };
                for (uint32_t t = 0; t < block; ++t) order[t] = t;
                for (uint32_t i = block; i > 1; --i) std::swap(order[i - 1], order[next() % i]);
                if (block > 64 && (next() & 3) == 0) {           // a burst: one wavefront alone for this pass
                    burst = true;
                    const uint32_t wsel = (uint32_t)(next() % ((block + 63) / 64));
                    for (uint32_t i = 0; i < block; ++i) order[i] = wsel * 64 + (i & 63) < block ? wsel * 64 + (i & 63) : i;
                }
            }
            for (uint32_t ti = 0; ti < block; ++ti) {
                const uint32_t t = order[ti];
                Fiber& f = g_fibers[t];
                if (f.done) continue;
                g_cur = (int)t;
                swapcontext(&g_sched, &f.ctx);
                if (f.done) remaining--;
                progressed++;
            }
            if (!progressed && !burst) break;
        }
    }
}

}  // namespace emu

// A copy of `n` bytes that ends exactly at an inaccessible page: a read of even one byte beyond the stream faults here
// instead of passing unnoticed (the kernels that take streams from outside must stay inside them whatever the bytes say).
struct GuardedCopy {
    uint8_t* base = nullptr;
    size_t mapped = 0;
    uint8_t* p = nullptr;
    GuardedCopy(const uint8_t* src, size_t n)
    {
        const size_t page = 4096;
        mapped = ((n + page - 1) / page + 1) * page;
        base = (uint8_t*)mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base == MAP_FAILED) abort();
        if (mprotect(base + mapped - page, page, PROT_NONE) != 0) abort();
        p = base + mapped - page - n;
        if (n) memcpy(p, src, n);
    }
    ~GuardedCopy() { munmap(base, mapped); }
};

// ---------------------------------------------------------------------------
// C entry points used by tests/test_emulated_kernels.py
// ---------------------------------------------------------------------------
extern "C" {

// Runs K1 + scan + gather on the CPU emulator.  Returns the stream length (0: unknown variant).
// variant = table kind * 10000 + form * 1000 + 500 + kernel:
//   kernel 1 = LDS-table kernel (one workgroup per block), 3 = global-table kernel (persistent, a few workgroups on one counter)
//   form   2 = bulk form, 3 = stream form of the parse
//   table  (global-table kernel) 1 = behind the LDS slot filter, 4 / 5 = behind the write-back slot cache of 512 / 256 slots
// e.g. 43503 = the product's default for blocks of more than 8 KiB, 12503 = for smaller ones, 3501 = the LDS-table kernel.
// (Rounds 1-3's other forms -- windowed / masked parses, look-ahead widths, lane-per-block, group, pair and duo kernels -- were
// removed in round 4 together with csrc/ablation/; profiles/HISTORY.md.)
uint64_t emu_compress_variant(const uint8_t* in, uint64_t n, uint32_t block_size, uint8_t* stream, uint64_t stream_cap, int variant)
{
    const int table_kind = variant / 10000;
    const int form = (variant % 10000) / 1000;
    const int kernel = variant % 100;
    if ((form != 2 && form != 3) || (kernel != 1 && kernel != 3) || (kernel == 3 && table_kind != 1 && table_kind != 4 && table_kind != 5) ||
        (kernel == 1 && table_kind != 0))
        return 0;
    const uint64_t need = 4ull + 32ull + block_size + block_size / 6;
    const uint32_t stride = (uint32_t)((need + 15) & ~15ull);
    const uint32_t nb = (uint32_t)((n + block_size - 1) / block_size);
    if (stream_cap < 10 + (uint64_t)nb * stride) return 0;
    std::vector<uint8_t> slots((size_t)nb * stride + 64, 0xAA);
    std::vector<uint32_t> bytes(nb + 1, 0);
    std::vector<uint64_t> offsets(nb + 1, 0);
    uint64_t stream_len = 0;
    // padded copy so the same (unaligned, slightly over-reading) loads stay inside the allocation
    std::vector<uint8_t> inbuf(n + 64, 0x55);
    if (n) memcpy(inbuf.data(), in, n);
    snappy_hip::K1Batch w{};
    w.count = 1;
    w.first_block[0] = 0;
    w.first_block[1] = nb;
    w.in[0] = inbuf.data();
    w.in_len[0] = n;
    w.slots[0] = slots.data();
    w.block_bytes[0] = bytes.data();
    if (nb && kernel == 3) {
        const uint32_t grid = nb < 3 ? nb : 3;
        std::vector<uint32_t> tables((size_t)grid * 16384, 0xBEEFBEEFu);     // never initialised on the GPU either
        uint32_t counter = 0;
        emu::launch(grid, 64, [&] {
            if (table_kind == 4 && form == 3) snappy_hip::compress_blocks_global_table_kernel<64, 3, 1, 512>(w, block_size, stride, tables.data(), &counter);
            else if (table_kind == 4) snappy_hip::compress_blocks_global_table_kernel<64, 2, 1, 512>(w, block_size, stride, tables.data(), &counter);
            else if (table_kind == 5 && form == 3) snappy_hip::compress_blocks_global_table_kernel<64, 3, 1, 256>(w, block_size, stride, tables.data(), &counter);
            else if (table_kind == 5) snappy_hip::compress_blocks_global_table_kernel<64, 2, 1, 256>(w, block_size, stride, tables.data(), &counter);
            else if (form == 3) snappy_hip::compress_blocks_global_table_kernel<64, 3, 1>(w, block_size, stride, tables.data(), &counter);
            else snappy_hip::compress_blocks_global_table_kernel<64, 2, 1>(w, block_size, stride, tables.data(), &counter);
        });
    } else if (nb) {
        emu::launch(nb, 64, [&] {
            if (form == 3) snappy_hip::compress_blocks_lds_table_kernel<64, 3>(w, block_size, stride, nullptr);
            else snappy_hip::compress_blocks_lds_table_kernel<64, 2>(w, block_size, stride, nullptr);
        });
    }
    emu::launch(1, 1024, [&] {
        snappy_hip::scan_block_bytes_kernel(bytes.data(), nb, (uint32_t)n, block_size, stream, offsets.data(), &stream_len);
    });
    if (nb)
        emu::launch(nb, 256, [&] {
            snappy_hip::gather_slots_kernel(slots.data(), stride, bytes.data(), offsets.data(), stream, nb);
        });
    return stream_len;
}

// The ceiling experiment's kernel (csrc/ablation/k1_oracle_table.hpp): the table answered from `rec` (n + 64 records
// made by tools/gate_b_records.c).  Returns the stream length.
uint64_t emu_compress_oracle(const uint8_t* in, uint64_t n, uint32_t block_size, uint8_t* stream, uint64_t stream_cap, const uint32_t* rec)
{
    const uint64_t need = 4ull + 32ull + block_size + block_size / 6;
    const uint32_t stride = (uint32_t)((need + 15) & ~15ull);
    const uint32_t nb = (uint32_t)((n + block_size - 1) / block_size);
    if (!nb || stream_cap < 10 + (uint64_t)nb * stride) return 0;
    std::vector<uint8_t> slots((size_t)nb * stride + 64, 0xAA);
    std::vector<uint32_t> bytes(nb + 1, 0);
    std::vector<uint64_t> offsets(nb + 1, 0);
    uint64_t stream_len = 0;
    std::vector<uint8_t> inbuf(n + 64, 0x55);
    memcpy(inbuf.data(), in, n);
    snappy_hip::K1Batch w{};
    w.count = 1;
    w.first_block[0] = 0;
    w.first_block[1] = nb;
    w.in[0] = inbuf.data();
    w.in_len[0] = n;
    w.slots[0] = slots.data();
    w.block_bytes[0] = bytes.data();
    uint32_t counter = 0;
    emu::launch(nb < 3 ? nb : 3, 64, [&] { snappy_hip::compress_blocks_oracle_kernel<false>(w, block_size, stride, rec, nullptr, nullptr, &counter); });
    emu::launch(1, 1024, [&] {
        snappy_hip::scan_block_bytes_kernel(bytes.data(), nb, (uint32_t)n, block_size, stream, offsets.data(), &stream_len);
    });
    emu::launch(nb, 256, [&] { snappy_hip::gather_slots_kernel(slots.data(), stride, bytes.data(), offsets.data(), stream, nb); });
    return stream_len;
}

uint64_t emu_compress(const uint8_t* in, uint64_t n, uint32_t block_size, uint8_t* stream, uint64_t stream_cap)
{
    return emu_compress_variant(in, n, block_size, stream, stream_cap, 12503);
}

// Runs index_streams_kernel + decompress_blocks_kernel on the emulator.
// Returns 0 on success, 1 if any block (or the chain) is invalid.
int emu_decompress_variant(const uint8_t* stream_in, uint64_t stream_len, uint32_t total_len, uint32_t block_size,
                           uint32_t header_len, uint8_t* out, int variant)
{
    GuardedCopy guarded(stream_in, stream_len);                  // K2 and the walk must not read one byte beyond the stream
    const uint8_t* stream = guarded.p;
    const uint32_t nb = block_size ? (uint32_t)(((uint64_t)total_len + block_size - 1) / block_size) : 0;
    if (nb == 0) return stream_len == header_len ? 0 : 1;
    std::vector<uint64_t> boff(nb, 0);
    uint32_t result[2] = {7, 7};
    snappy_hip::StreamDesc d{stream, stream_len, boff.data(), result, total_len, block_size, header_len, nb};
    emu::launch(1, 64, [&] { snappy_hip::index_streams_kernel(&d, 1, 1u); });
    if (result[0] != 0 || result[1] != nb) return 1;
    std::vector<uint32_t> status(nb, 9);
    uint32_t k2_counter = 0;
    snappy_hip::K2Batch kb{};
    kb.count = 1;
    kb.first_block[0] = 0;
    kb.first_block[1] = nb;
    kb.stream[0] = stream;
    kb.stream_len[0] = stream_len;
    kb.block_offsets[0] = boff.data();
    kb.total_len[0] = total_len;
    kb.out[0] = out;
    kb.status[0] = status.data();
    (void)variant;                                                       // (one decoder: the per-window batch)
    emu::launch(nb < 3 ? nb : 3, 64, [&] { snappy_hip::decompress_blocks_kernel(kb, block_size, &k2_counter); });
    for (uint32_t i = 0; i < nb; ++i)
        if (status[i] != 0) return 1;
    return 0;
}

// The size chain in parallel segments (chain_anchor / chain_walk / chain_finish kernels) followed by the serial walk for
// what they leave unresolved, as snappy_hip_index_streams enqueues them.  offsets: num_blocks entries; result[0..1] as the
// kernels leave it; returns 1 if the parallel segments resolved the stream, 0 if the serial walk had to.
int emu_index_parallel(const uint8_t* stream_in, uint64_t stream_len, uint64_t* offsets, uint32_t total_len, uint32_t block_size,
                       uint32_t header_len, uint32_t* result)
{
    const uint32_t nb = block_size ? (uint32_t)(((uint64_t)total_len + block_size - 1) / block_size) : 0;
    result[0] = result[1] = 7;
    GuardedCopy guarded(stream_in, stream_len);
    const uint8_t* stream = guarded.p;
    snappy_hip::StreamDesc d{stream, stream_len, offsets, result, total_len, block_size, header_len, nb};
    constexpr size_t K = snappy_hip::kChainSegments, C = snappy_hip::kChainSegCap;
    std::vector<uint32_t> mem(K * (3 + C) + 1, 0xdeadbeefu);
    snappy_hip::ChainWork w;
    w.anchor = mem.data();
    w.seg_hops = w.anchor + K;
    w.seg_ok = w.seg_hops + K;
    w.hops = w.seg_ok + K;
    w.resolved = w.hops + K * C;
    emu::launch((uint32_t)K, 64, [&] { snappy_hip::chain_anchor_kernel(&d, 1, w); });
    emu::launch((uint32_t)K, 64, [&] { snappy_hip::chain_walk_kernel(&d, 1, w); });
    emu::launch(1, 1024, [&] { snappy_hip::chain_finish_kernel(&d, 1, w); });
    const int resolved = (int)w.resolved[0];
    emu::launch(1, 64, [&] { snappy_hip::index_streams_kernel(&d, 1, 1u, w.resolved); });
    return resolved;
}

// Runs verify_index_begin_kernel + verify_index_kernel on a candidate index of num_blocks + 1 offsets.
void emu_verify_index(const uint8_t* stream, uint64_t stream_len, uint64_t* offsets, uint32_t total_len, uint32_t block_size,
                      uint32_t header_len, uint32_t* result)
{
    const uint32_t nb = block_size ? (uint32_t)(((uint64_t)total_len + block_size - 1) / block_size) : 0;
    result[0] = result[1] = 7;
    snappy_hip::StreamDesc d{stream, stream_len, offsets, result, total_len, block_size, header_len, nb};
    emu::launch(1, 64, [&] { snappy_hip::verify_index_begin_kernel(&d, 1); });
    emu::launch(snappy_hip::kVerifyGroup, 256, [&] { snappy_hip::verify_index_kernel(&d, 1); });
}

// statistics of the stream form (snappy_k1_stream.hpp), cleared by the read
void emu_stream_stats(unsigned long long* out)
{
    for (int i = 0; i < 16; ++i) {
        out[i] = snappy_hip::g_stream_stats[i];
        snappy_hip::g_stream_stats[i] = 0;
    }
}

int emu_decompress(const uint8_t* stream, uint64_t stream_len, uint32_t total_len, uint32_t block_size, uint32_t header_len,
                   uint8_t* out)
{
    return emu_decompress_variant(stream, stream_len, total_len, block_size, header_len, out, 1);
}
}
