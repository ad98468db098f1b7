// snappy_hip.hip -- host side of libsnappy_hip.so: the C ABI declared in include/snappy_hip.h.
//
// Replaces the UPMEM offload plumbing of the reference (dpu_alloc / dpu_load / dpu_push_xfer /
// dpu_launch / dpu_free in snappy/snappy_compress.c:487-714 and snappy/snappy_decompress.c:292-493)
// with hipMalloc / hipMemcpy / kernel launches.  No CPU codec lives in this library: if HIP is
// unusable every entry point fails and says so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/snappy_hip.h"
#include "shard_devices.hpp"
#include "launch_shape.hpp"
#include "host_chain.hpp"
#include "snappy_kernels.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& what)
{
    g_last_error = what;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(SNAPPY_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

double now_seconds()
{
    struct timeval tv;
    gettimeofday(&tv, nullptr);   // same clock as the reference's get_runtime (dpu_snappy.c:87-91)
    return (double)tv.tv_sec + (double)tv.tv_usec / 1000000.0;
}

uint32_t put_varint32(uint8_t* dst, uint32_t v)   // snappy_compress.c:69-98
{
    uint32_t k = 0;
    while (v >= 0x80) {
        dst[k++] = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    dst[k++] = (uint8_t)v;
    return k;
}

uint32_t get_varint32(const uint8_t* src, uint64_t avail, uint32_t* out)   // snappy_decompress.c:23-37
{
    uint32_t v = 0;
    for (uint32_t k = 0; k < 5 && k < avail; ++k) {
        const uint8_t c = src[k];
        v |= (uint32_t)(c & 0x7f) << (7 * k);
        if (!(c & 0x80)) {
            *out = v;
            return k + 1;
        }
    }
    return 0;
}

uint32_t le32_host(const uint8_t* p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

bool block_size_ok(uint32_t bs) { return bs >= SNAPPY_HIP_MIN_BLOCK_SIZE && bs <= SNAPPY_HIP_MAX_BLOCK_SIZE; }

// Run fn(g) for g in [0, count) -- one host thread per device when count > 1 (pageable
// hipMemcpy is synchronous per call, so threads are what overlaps the per-GPU transfers).
int for_each_device(int count, const std::function<int(int)>& fn)
{
    if (count == 1) return fn(0);
    std::vector<int> rc(count, 0);
    std::vector<std::string> err(count);
    std::vector<std::thread> th;
    for (int g = 0; g < count; ++g)
        th.emplace_back([&, g] {
            rc[g] = fn(g);
            if (rc[g]) err[g] = g_last_error;
        });
    for (auto& t : th) t.join();
    for (int g = 0; g < count; ++g)
        if (rc[g]) return fail(rc[g], "GPU " + std::to_string(g) + ": " + err[g]);
    return 0;
}

constexpr int kVariantLdsTable = 1, kVariantGlobalTable = 3;
#ifdef SNAPPY_ABLATION
// The ablation build (tools/build_ablation.py) = the product + ONE experiment kernel: K1 with its hash table answered from
// host-made records (csrc/ablation/k1_oracle_table.hpp, gate (b) of round 4's two-pass question).  The kernel forms of rounds
// 1-3 that do not ship (windowed / masked parses, unfiltered and class-filtered tables, lane-per-block, four blocks per
// wavefront, two-wavefront LDS forms, K2's element loop) were removed in round 4; they are in the history (profiles/HISTORY.md).
constexpr int kVariantOracle = 6, kVariantOracleWithCosts = 7;
const uint32_t* g_oracle_records = nullptr;    // device array, one u32 per input position (tools/gate_b_ceiling.py)
const uint16_t* g_oracle_prevw = nullptr;      // device array, one u16 per input position (variant 7)
#endif
constexpr int kDefaultLdsHeadStart = 6; // ~20 us for the LDS-table workgroups to be placed before the global-table kernel starts
constexpr int kDefaultGtCache = 512;    // SNAPPY_HIP_GT_CACHE: slots of the write-back cache in LDS in front of the global table, for blocks with full-size hash tables; 0 = none
constexpr int kDefaultK1Stream = 1;     // SNAPPY_HIP_K1_STREAM: bit 0 = stream form (snappy_k1_stream.hpp) for the LDS-table kernel (default: +2 % in the mix), bit 1 = for the global-table kernel (default with the slot cache: +3 % there, -2 % without)

// Work counters for persistent kernels: a ring in the code object's own global memory (one copy per device), so launches
// need no allocation.  Each launch takes the next slot of its device's ring, zeroes it on its stream and leaves an event
// behind; a slot is handed out again only after the event of its previous launch has completed (normally long ago: the
// ring has 256 slots), so two launches in flight never share a counter however many streams the caller uses.
constexpr uint32_t kWorkCounterSlots = 256;
__device__ uint32_t g_work_counters[kWorkCounterSlots * 16];      // one counter per 64-byte line

struct WorkCounterRing {
    std::mutex m;
    uint32_t turn = 0;
    hipEvent_t busy[kWorkCounterSlots] = {};
    bool taken[kWorkCounterSlots] = {};      // handed out, launch event not recorded yet
};
WorkCounterRing* work_counter_ring(int dev)
{
    static WorkCounterRing* rings[64] = {};
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!rings[dev]) rings[dev] = new WorkCounterRing;             // never destroyed (threads may outlive statics)
    return rings[dev];
}

// a zeroed counter for one launch on `st`; call work_counter_launched() right after the launch
struct WorkCounter {
    uint32_t* ptr = nullptr;
    hipEvent_t done = nullptr;
    WorkCounterRing* ring = nullptr;
    uint32_t slot = 0;
};
int next_work_counter(WorkCounter* out, hipStream_t st)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(SNAPPY_HIP_ERR_ARG, "device index out of range");
    uint32_t* base = nullptr;
    HIP_TRY(hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_work_counters)));
    WorkCounterRing* ring = work_counter_ring(dev);
    uint32_t slot;
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lock(ring->m);
        uint32_t tries = 0;
        do {
            slot = ring->turn++ % kWorkCounterSlots;
        } while (ring->taken[slot] && ++tries < kWorkCounterSlots);
        if (ring->taken[slot]) return fail(SNAPPY_HIP_ERR_RUNTIME, "all work counters are being launched");
        if (!ring->busy[slot]) HIP_TRY(hipEventCreateWithFlags(&ring->busy[slot], hipEventDisableTiming));
        ring->taken[slot] = true;
        ev = ring->busy[slot];
    }
    out->ptr = base + slot * 16;
    out->done = ev;
    out->ring = ring;
    out->slot = slot;
    hipError_t e = hipEventSynchronize(ev);                        // an event never recorded counts as complete
    if (e == hipSuccess) e = hipMemsetAsync(out->ptr, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> lock(ring->m);
        ring->taken[slot] = false;
        return fail(SNAPPY_HIP_ERR_RUNTIME, std::string("work counter: ") + hipGetErrorString(e));
    }
    return 0;
}
// records the launch's completion event on `st` and releases the slot for reuse behind that event
int work_counter_launched(const WorkCounter& c, hipStream_t st)
{
    const hipError_t e = hipEventRecord(c.done, st);
    {
        std::lock_guard<std::mutex> lock(c.ring->m);
        c.ring->taken[c.slot] = false;
    }
    if (e != hipSuccess) return fail(SNAPPY_HIP_ERR_RUNTIME, std::string("hipEventRecord: ") + hipGetErrorString(e));
    return 0;
}

// Helper stream + fork/join events for launches that co-run two kernels, one set per (host thread, device).  The
// events are timing-disabled; the helper stream is non-blocking, so the only ordering is the explicit fork (ev_begin)
// and join (ev_end) around the caller's stream.
struct CoRunResources {
    hipStream_t helper = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
};

// The sets live in a per-device pool; a host thread leases one per device on first use and hands it back when it ends
// (the drop-in pair runs one short-lived thread per shard), so streams are created once per process, not per call.
struct CoRunPool {
    std::mutex m;
    std::vector<CoRunResources*> idle[64];
};
CoRunPool& corun_pool()
{
    static CoRunPool* pool = new CoRunPool;     // never destroyed: threads may outlive static destruction order
    return *pool;
}
struct CoRunLease {
    CoRunResources* held[64] = {};
    ~CoRunLease()
    {
        CoRunPool& pool = corun_pool();
        std::lock_guard<std::mutex> lock(pool.m);
        for (int d = 0; d < 64; ++d)
            if (held[d]) pool.idle[d].push_back(held[d]);
    }
};

int corun_resources(CoRunResources** out)
{
    static thread_local CoRunLease lease;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(SNAPPY_HIP_ERR_ARG, "device index out of range");
    if (!lease.held[dev]) {
        CoRunPool& pool = corun_pool();
        {
            std::lock_guard<std::mutex> lock(pool.m);
            if (!pool.idle[dev].empty()) {
                lease.held[dev] = pool.idle[dev].back();
                pool.idle[dev].pop_back();
            }
        }
        if (!lease.held[dev]) {
            CoRunResources* r = new CoRunResources;
            HIP_TRY(hipStreamCreateWithFlags(&r->helper, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&r->ev_begin, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&r->ev_end, hipEventDisableTiming));
            lease.held[dev] = r;
        }
    }
    *out = lease.held[dev];
    return 0;
}

int env_int(const char* name, int fallback)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}

// The shape of the current device -- compute units, LDS per CU, wavefront slots per CU -- read once per device.  Every launch
// and the hash-table scratch are sized from it (csrc/launch_shape.hpp), so a partition of the chip (CPX / NPS modes: 32 CUs
// per logical device) gets grids of its own size.  Falls back to a whole MI355X if the runtime cannot be asked.
launch_shape::DeviceShape device_shape()
{
    static std::mutex m;
    static launch_shape::DeviceShape shapes[64];
    static bool known[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return launch_shape::DeviceShape();
    std::lock_guard<std::mutex> lock(m);
    if (!known[dev]) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 && p.maxSharedMemoryPerMultiProcessor >= (32u << 10) &&
            p.maxThreadsPerMultiProcessor >= 64) {
            shapes[dev].cus = (uint32_t)p.multiProcessorCount;
            shapes[dev].lds_per_cu = (uint32_t)p.maxSharedMemoryPerMultiProcessor;
            shapes[dev].wave_slots_per_cu = (uint32_t)p.maxThreadsPerMultiProcessor / 64u;
        }
        // SNAPPY_HIP_TEST_DEVICE_CUS (test hook, read once per device): pretend the device has this many compute units -- a
        // partition of the chip as CPX / NPS modes make it -- so that the launch and scratch sizing for a small device can be
        // exercised on a whole one (tests/test_gpu_parity.py)
        const int fake = env_int("SNAPPY_HIP_TEST_DEVICE_CUS", 0);
        if (fake > 0 && (uint32_t)fake <= shapes[dev].cus) shapes[dev].cus = (uint32_t)fake;
        known[dev] = true;
    }
    return shapes[dev];
}

using launch_shape::lds_alloc_bytes;

// K1's defaults depend on the block size: blocks of more than 8 KiB have the full 16384-slot table, whose global-table form
// runs at the HBM's random-access rate; there the slot cache and the stream form pay (profiles/r03_gt_cache_block_size_sweep.txt)
int gt_cache_slots(uint32_t block_size)
{
    return env_int("SNAPPY_HIP_GT_CACHE", block_size > 8192u ? kDefaultGtCache : 0) ? 512 : 0;    // (values other than 0 / 512 are refused: check_knobs)
}
int k1_stream_forms(uint32_t block_size)
{
    return env_int("SNAPPY_HIP_K1_STREAM", kDefaultK1Stream | (gt_cache_slots(block_size) ? 2 : 0));
}

// dynamic LDS of one LDS-table workgroup of the product's launch at this block size
uint32_t lds_table_wave_bytes(uint32_t block_size)
{
    return (k1_stream_forms(block_size) & 1) ? snappy_hip::lds_table_stream_lds_bytes(block_size)
                                             : snappy_hip::lds_table_kernel_lds_bytes(block_size, true);
}

uint32_t default_lds_waves_per_cu(uint32_t block_size)
{
    launch_shape::K1Knobs k;
    k.cached_global_table = gt_cache_slots(block_size) != 0;
    k.lds_wave_bytes = lds_table_wave_bytes(block_size);
    return launch_shape::default_lds_waves_per_cu(device_shape(), k);
}

// Number of shards the drop-in pair splits a file into: SNAPPY_HIP_NUM_GPUS (default: every visible device).
// SNAPPY_HIP_OVERSUBSCRIBE=1 (test hook) allows more shards than devices; shard g then runs on device
// (base + g) % device_count, so the sharding and host-side concat paths can be exercised on a one-GPU box.
// (ShardDevices -- per call: where the shards of THIS call of the drop-in pair run -- lives in shard_devices.hpp)
ShardDevices requested_devices()
{
    ShardDevices d;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return d;
    d.physical = have;
    int cur = 0;
    if (hipGetDevice(&cur) == hipSuccess && cur > 0 && cur < have) d.base = cur;
    const char* env = getenv("SNAPPY_HIP_NUM_GPUS");
    if (env && *env) {
        const int want = atoi(env);
        const bool over = env_int("SNAPPY_HIP_OVERSUBSCRIBE", 0) != 0;
        if (want >= 1 && (want < have || (over && want <= 64))) have = want;
    }
    d.shards = have;
    return d;
}

// The drop-in pair leaves the calling thread's current device as it found it (its shard threads are its own).
struct CallerDevice {
    int dev = -1;
    CallerDevice()
    {
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    }
    ~CallerDevice()
    {
        if (dev >= 0) (void)hipSetDevice(dev);
    }
};

// K1 launchers: the LDS-table kernel and the global-table kernel, each in the bulk or the stream form of the parse, the
// global-table one behind its slot cache for blocks with full-size hash tables.

// Knobs of earlier rounds' kernel forms, and values of the product's own knobs that no build implements any more, are
// refused, not ignored: a sweep must never produce numbers labelled with a configuration that did not run.
int check_knobs()
{
    for (const char* name : {"SNAPPY_HIP_K1_AHEAD", "SNAPPY_HIP_K1_AHEAD_LDS", "SNAPPY_HIP_K1_FORM", "SNAPPY_HIP_K1_FORM_LDS",
                             "SNAPPY_HIP_K1_FILTER", "SNAPPY_HIP_EXTRA_LDS", "SNAPPY_HIP_LANES_PER_BLOCK", "SNAPPY_HIP_GROUP_WAVES",
                             "SNAPPY_HIP_PAIR_PER_CU"})
        if (getenv(name))
            return fail(SNAPPY_HIP_ERR_ARG, std::string(name) + " selected a kernel form of rounds 1-3 that was removed in round 4 "
                                                                "(profiles/HISTORY.md names the commit that has them)");
    if (const char* v = getenv("SNAPPY_HIP_GT_CACHE"))
        if (*v && atoi(v) != 0 && atoi(v) != 512)
            return fail(SNAPPY_HIP_ERR_ARG, "SNAPPY_HIP_GT_CACHE: the slot cache has 512 slots or is off (0)");
    if (const char* v = getenv("SNAPPY_HIP_K1_STREAM"))
        if (*v && (atoi(v) & ~3))
            return fail(SNAPPY_HIP_ERR_ARG, "SNAPPY_HIP_K1_STREAM: bits 0 and 1 select the stream form per kernel (bit 2, round 3's duo form, was removed)");
    return 0;
}

void launch_lds_table_kernel(uint32_t grid, hipStream_t st, const snappy_hip::K1Batch& w, uint32_t block_size, uint32_t slot_stride,
                             uint32_t* counter)
{
    if (k1_stream_forms(block_size) & 1)
        hipLaunchKernelGGL((snappy_hip::compress_blocks_lds_table_kernel<64, 3>), dim3(grid), dim3(64),
                           snappy_hip::lds_table_stream_lds_bytes(block_size), st, w, block_size, slot_stride, counter);
    else
        hipLaunchKernelGGL((snappy_hip::compress_blocks_lds_table_kernel<64, 2>), dim3(grid), dim3(64),
                           snappy_hip::lds_table_kernel_lds_bytes(block_size, true), st, w, block_size, slot_stride, counter);
}

void launch_global_table_kernel(uint32_t grid, hipStream_t st, const snappy_hip::K1Batch& w, uint32_t block_size, uint32_t slot_stride,
                                uint32_t* tables, uint32_t* counter)
{
    const bool cached = gt_cache_slots(block_size) != 0;
    const bool stream_form = (k1_stream_forms(block_size) & 2) != 0;
    if (cached && stream_form)
        hipLaunchKernelGGL((snappy_hip::compress_blocks_global_table_kernel<64, 3, 1, 512>), dim3(grid), dim3(64), 0, st, w, block_size,
                           slot_stride, tables, counter);
    else if (cached)
        hipLaunchKernelGGL((snappy_hip::compress_blocks_global_table_kernel<64, 2, 1, 512>), dim3(grid), dim3(64), 0, st, w, block_size,
                           slot_stride, tables, counter);
    else if (stream_form)
        hipLaunchKernelGGL((snappy_hip::compress_blocks_global_table_kernel<64, 3, 1>), dim3(grid), dim3(64), 0, st, w, block_size,
                           slot_stride, tables, counter);
    else
        hipLaunchKernelGGL((snappy_hip::compress_blocks_global_table_kernel<64, 2, 1>), dim3(grid), dim3(64), 0, st, w, block_size,
                           slot_stride, tables, counter);
}

}  // namespace

namespace {

// Workspace of the parallel-segment chain resolution (csrc/snappy_kernels.hpp: chain_*_kernel), one per device, owned by the
// library: 2.1 MB per stream of a call, allocated on first use and when a call brings more streams than any before it (the
// only moment snappy_hip_index_streams touches the allocator).  Calls that overlap in time on different streams are ordered
// by an event: the later one waits ON THE DEVICE for the earlier one to be done with the workspace.
struct ChainWorkspace {
    std::mutex m;
    uint32_t* mem = nullptr;
    uint32_t streams = 0;
    hipEvent_t last_use = nullptr;
};
ChainWorkspace* chain_workspace(int dev)
{
    static ChainWorkspace* per_device[64] = {};
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!per_device[dev]) per_device[dev] = new ChainWorkspace;    // never destroyed (threads may outlive statics)
    return per_device[dev];
}

}  // namespace

// ===========================================================================
// resident API
// ===========================================================================
extern "C" {

void* snappy_hip_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        g_last_error = "hipHostMalloc failed";
        return nullptr;
    }
    return p;
}

void snappy_hip_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int snappy_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        g_last_error = "hipGetDeviceCount failed: no usable HIP runtime/device";
        return 0;
    }
    return n;
}

int snappy_hip_set_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return SNAPPY_HIP_OK;
}

const char* snappy_hip_last_error(void) { return g_last_error.c_str(); }

const char* snappy_hip_arch(void) { return "gfx950"; }

uint32_t snappy_hip_slot_stride(uint32_t block_size)
{
    const uint64_t need = 4ull + 32ull + block_size + block_size / 6;   // snappy_compress.c:55-60 + prefix
    return (uint32_t)((need + 15) & ~15ull);
}

uint64_t snappy_hip_num_blocks(uint64_t input_len, uint32_t block_size)
{
    return block_size ? (input_len + block_size - 1) / block_size : 0;
}

uint64_t snappy_hip_stream_bound(uint64_t input_len, uint32_t block_size)
{
    return 10 + snappy_hip_num_blocks(input_len, block_size) * (uint64_t)snappy_hip_slot_stride(block_size);
}

uint32_t snappy_hip_write_header(uint8_t* dst, uint32_t total_len, uint32_t block_size)
{
    uint32_t k = put_varint32(dst, total_len);
    k += put_varint32(dst + k, block_size);
    return k;
}

uint32_t snappy_hip_parse_header(const uint8_t* src, uint64_t avail, uint32_t* total_len, uint32_t* block_size)
{
    const uint32_t a = get_varint32(src, avail, total_len);
    if (!a) return 0;
    const uint32_t b = get_varint32(src + a, avail - a, block_size);
    if (!b) return 0;
    return a + b;
}

#ifdef SNAPPY_ABLATION
// ablation build only: the records OracleTable / RecMate read (csrc/ablation/k1_oracle_table.hpp)
void snappy_hip_debug_set_oracle_records(const uint32_t* d_records) { g_oracle_records = d_records; }
void snappy_hip_debug_set_oracle_prevw(const uint16_t* d_prevw) { g_oracle_prevw = d_prevw; }
#endif

#ifdef SNAPPY_PROF
// probe builds only (tools/prof_stream.py): read / reset the lap timers of the stream form
int snappy_hip_debug_prof(unsigned long long* out, int reset)
{
    if (reset) {
        unsigned long long z[32] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(snappy_hip::g_prof), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(snappy_hip::g_prof), 32 * sizeof(unsigned long long));
}
#endif

uint32_t snappy_hip_k1_lds_waves_per_cu(uint32_t block_size)
{
    if (!block_size_ok(block_size)) return 0;
    const launch_shape::DeviceShape shape = device_shape();
    const int forced = env_int("SNAPPY_HIP_LDS_WAVES", -1);
    return forced >= 0 ? ((uint32_t)forced + shape.cus - 1) / shape.cus : default_lds_waves_per_cu(block_size);
}

uint64_t snappy_hip_compress_scratch_bytes(void)
{
    // 256-byte header (work counter) + one 64 KiB hash table per wavefront slot of the CURRENT device (MI355X: 256 CUs x 32)
    return launch_shape::compress_scratch_bytes(device_shape());
}

// K1 over a batch of containers (count >= 1, every container non-empty and validated by the callers below)
static int launch_compress(const snappy_hip::K1Batch& w, uint32_t block_size, uint32_t slot_stride, void* d_scratch,
                           uint64_t scratch_bytes, void* stream)
{
    const uint64_t nb = w.first_block[w.count];
    // SNAPPY_HIP_COMPRESS_VARIANT: 3 = the concurrent launch below (default); 1 = the LDS-table kernel alone (also the path
    // taken when no scratch is given); the ablation build has 6 = the free-table experiment (csrc/ablation/k1_oracle_table.hpp).
    int variant = env_int("SNAPPY_HIP_COMPRESS_VARIANT", kVariantGlobalTable);
    if (variant == kVariantGlobalTable &&
        (!d_scratch || scratch_bytes < snappy_hip_compress_scratch_bytes() || ((uintptr_t)d_scratch & 255)))
        variant = kVariantLdsTable;   // no scratch: LDS-table kernel (still on the GPU)
    if (int rc = check_knobs()) return rc;
    hipStream_t st = (hipStream_t)stream;
    const launch_shape::DeviceShape shape = device_shape();
    const bool scratch_usable = d_scratch && scratch_bytes >= 256 && !((uintptr_t)d_scratch & 255);
    // A small input -- every block can have an LDS-table wavefront at once, at most one per SIMD -- goes to the LDS-table
    // kernel alone: its wavefronts run the stream form with nothing else on their SIMD (dickens_like, 312 blocks: K1 1.27 ->
    // 1.05 ms, profiles/r03_small_inputs.txt); with more blocks than that a second round would follow, and the global-table
    // kernel's 32 wave slots per CU win.
    if (variant == kVariantGlobalTable && !getenv("SNAPPY_HIP_COMPRESS_VARIANT") && !getenv("SNAPPY_HIP_LDS_WAVES") &&
        (k1_stream_forms(block_size) & 1) &&
        launch_shape::small_input_takes_lds_kernel_alone(shape, snappy_hip::lds_table_stream_lds_bytes(block_size), nb))
        variant = kVariantLdsTable;
#ifdef SNAPPY_ABLATION
    if (variant == kVariantOracle || variant == kVariantOracleWithCosts) {     // ceiling experiment (csrc/ablation/k1_oracle_table.hpp)
        const bool costs = variant == kVariantOracleWithCosts;
        if (w.count != 1 || !g_oracle_records || !d_scratch ||
            (costs && (!g_oracle_prevw || scratch_bytes < snappy_hip_compress_scratch_bytes() || block_size > 32768u)))
            return fail(SNAPPY_HIP_ERR_ARG, "variants 6 / 7 take one container, a scratch and snappy_hip_debug_set_oracle_records() (7: + _prevw())");
        uint32_t* counter = static_cast<uint32_t*>(d_scratch);
        uint16_t* memo = reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(d_scratch) + 256);
        HIP_TRY(hipMemsetAsync(counter, 0, 32, st));
        const uint32_t waves = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(nb, shape.wave_slots()), (uint64_t)env_int("SNAPPY_HIP_GT_WAVES", (int)(20 * shape.cus)));
        if (costs)
            hipLaunchKernelGGL(snappy_hip::compress_blocks_oracle_kernel<true>, dim3(waves), dim3(64), 0, st, w, block_size, slot_stride,
                               g_oracle_records, g_oracle_prevw, memo, counter);
        else
            hipLaunchKernelGGL(snappy_hip::compress_blocks_oracle_kernel<false>, dim3(waves), dim3(64), 0, st, w, block_size, slot_stride,
                               g_oracle_records, g_oracle_prevw, memo, counter);
        HIP_TRY(hipGetLastError());
        return SNAPPY_HIP_OK;
    }
#endif
    if (variant != kVariantGlobalTable && variant != kVariantLdsTable)
        return fail(SNAPPY_HIP_ERR_ARG, "SNAPPY_HIP_COMPRESS_VARIANT: 1 (LDS-table kernel alone) or 3 (the concurrent launch, default)");
    if (variant == kVariantLdsTable) {
        launch_lds_table_kernel((uint32_t)nb, st, w, block_size, slot_stride, (uint32_t*)nullptr);
        HIP_TRY(hipGetLastError());
        // the statistics word of the scratch (include/snappy_hip.h): every block of this launch had an LDS-table wavefront
        if (scratch_usable) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(static_cast<uint32_t*>(d_scratch) + 4), (int)nb, 1, st));
        return SNAPPY_HIP_OK;
    }

    // ---- the default: persistent grids, blocks handed out by an atomic counter kept in the first bytes of the scratch ----
    // Wave budget per CU (MI355X: 256 CUs, 32 wave slots, 160 KiB of LDS each; device_shape()): a global-table wavefront holds
    // the duplicate test, the slot filter (2 KiB) and, for blocks with full-size tables, the slot cache (2 KiB) in LDS and its
    // table in the scratch; the wavefronts whose table lives in LDS are sized by the block length (csrc/launch_shape.hpp).
    // SNAPPY_HIP_LDS_WAVES overrides the LDS-table wavefronts, SNAPPY_HIP_GT_WAVES the TOTAL of both kinds.
    uint32_t* counter = static_cast<uint32_t*>(d_scratch);
    uint32_t* tables = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(d_scratch) + 256);
    HIP_TRY(hipMemsetAsync(counter, 0, 32, st));   // [0] next block, [4] blocks compressed by wavefronts with an LDS table
    const int k1_stream = k1_stream_forms(block_size);
    const int gt_cache = gt_cache_slots(block_size);
    launch_shape::K1Knobs knobs;
    knobs.cached_global_table = gt_cache != 0;
    knobs.lds_wave_bytes = lds_table_wave_bytes(block_size);
    knobs.gt_wave_bytes = gt_cache ? ((k1_stream & 2) ? 4u << 10 : 3u << 10) + 4u * (uint32_t)gt_cache
                          : (k1_stream & 2) ? (2u << 10) + snappy_hip::stream_scratch_bytes(snappy_hip::kStreamSlotsGlobal)
                                            : 3u << 10;
    knobs.lds_waves_forced = env_int("SNAPPY_HIP_LDS_WAVES", -1);
    knobs.waves_forced = env_int("SNAPPY_HIP_GT_WAVES", -1);
    knobs.hybrid_min_blocks = (uint64_t)env_int("SNAPPY_HIP_HYBRID_MIN_BLOCKS", 4096);     // small inputs: one kernel is enough
    const launch_shape::K1Launch l = launch_shape::k1_default_launch(shape, knobs, nb);
    if (l.lds_waves) {
        // fork / join around the caller's stream: the LDS-table kernel goes to the helper stream, the global-table kernel stays
        // on `st`; both draw blocks from the same counter, so the split balances itself
        CoRunResources* cr = nullptr;
        if (int rc = corun_resources(&cr)) return rc;
        HIP_TRY(hipEventRecord(cr->ev_begin, st));                     // after the counter memset and all prior work
        HIP_TRY(hipStreamWaitEvent(cr->helper, cr->ev_begin, 0));
        launch_lds_table_kernel(l.lds_waves, cr->helper, w, block_size, slot_stride, counter);
        HIP_TRY(hipEventRecord(cr->ev_end, cr->helper));
        // a head start for the LDS-heavy workgroups: placed first, the small allocations cannot fragment the LDS under them
        if (const int head_start = env_int("SNAPPY_HIP_LDS_HEAD_START", kDefaultLdsHeadStart))   // x 3.4 us
            hipLaunchKernelGGL(snappy_hip::delay_kernel, dim3(1), dim3(64), 0, st, (uint32_t)head_start);
        launch_global_table_kernel(l.gt_waves, st, w, block_size, slot_stride, tables, counter);
        HIP_TRY(hipStreamWaitEvent(st, cr->ev_end, 0));                // the caller's stream resumes when both are done
    } else {
        launch_global_table_kernel(l.gt_waves, st, w, block_size, slot_stride, tables, counter);
    }
    HIP_TRY(hipGetLastError());
    return SNAPPY_HIP_OK;
}

static int check_container(const void* d_in, uint64_t input_len, uint32_t block_size, const void* d_slots, uint32_t slot_stride,
                           const void* d_block_bytes)
{
    if (!block_size_ok(block_size)) return fail(SNAPPY_HIP_ERR_ARG, "block_size must be 1..65535");
    if (input_len > 0xffffffffull) return fail(SNAPPY_HIP_ERR_ARG, "container length must fit uint32 (snappy_compress.c:461)");
    if (slot_stride < snappy_hip_slot_stride(block_size)) return fail(SNAPPY_HIP_ERR_ARG, "slot_stride too small");
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_slots & 15)) return fail(SNAPPY_HIP_ERR_ARG, "d_in and d_slots must be 16-byte aligned");
    if (input_len && (!d_in || !d_slots || !d_block_bytes)) return fail(SNAPPY_HIP_ERR_ARG, "null device pointer");
    return SNAPPY_HIP_OK;
}

int snappy_hip_compress_blocks(const uint8_t* d_in, uint64_t input_len, uint32_t block_size, uint8_t* d_slots,
                               uint32_t slot_stride, uint32_t* d_block_bytes, void* d_scratch, uint64_t scratch_bytes,
                               void* stream)
{
    if (int rc = check_container(d_in, input_len, block_size, d_slots, slot_stride, d_block_bytes)) return rc;
    const uint64_t nb = snappy_hip_num_blocks(input_len, block_size);
    if (nb == 0) return SNAPPY_HIP_OK;
    snappy_hip::K1Batch w{};
    w.count = 1;
    w.first_block[0] = 0;
    w.first_block[1] = (uint32_t)nb;
    w.in[0] = d_in;
    w.in_len[0] = input_len;
    w.slots[0] = d_slots;
    w.block_bytes[0] = d_block_bytes;
    return launch_compress(w, block_size, slot_stride, d_scratch, scratch_bytes, stream);
}

int snappy_hip_compress_blocks_batch(const struct snappy_hip_compress_item* items, uint32_t count, uint32_t block_size,
                                     uint32_t slot_stride, void* d_scratch, uint64_t scratch_bytes, void* stream)
{
    if (!items && count) return fail(SNAPPY_HIP_ERR_ARG, "null item array");
    // empty containers are skipped; a launch takes at most kMaxBatch non-empty ones, so longer lists go out in groups
    uint32_t i = 0;
    while (i < count) {
        snappy_hip::K1Batch w{};
        uint64_t blocks = 0;
        while (i < count && w.count < snappy_hip::kMaxBatch) {
            const snappy_hip_compress_item& it = items[i];
            if (int rc = check_container(it.d_input, it.input_len, block_size, it.d_slots, slot_stride, it.d_block_bytes)) return rc;
            const uint64_t nb = snappy_hip_num_blocks(it.input_len, block_size);
            if (nb) {
                if (blocks + nb > 0xffffffffull) break;
                w.first_block[w.count] = (uint32_t)blocks;
                w.in[w.count] = static_cast<const uint8_t*>(it.d_input);
                w.in_len[w.count] = it.input_len;
                w.slots[w.count] = static_cast<uint8_t*>(it.d_slots);
                w.block_bytes[w.count] = static_cast<uint32_t*>(it.d_block_bytes);
                blocks += nb;
                ++w.count;
            }
            ++i;
        }
        w.first_block[w.count] = (uint32_t)blocks;
        if (w.count)
            if (int rc = launch_compress(w, block_size, slot_stride, d_scratch, scratch_bytes, stream)) return rc;
    }
    return SNAPPY_HIP_OK;
}

int snappy_hip_compact(const uint8_t* d_slots, uint32_t slot_stride, const uint32_t* d_block_bytes, uint64_t input_len,
                       uint32_t block_size, uint8_t* d_stream, uint64_t* d_offsets, uint64_t* d_stream_len, void* stream)
{
    if (!block_size_ok(block_size)) return fail(SNAPPY_HIP_ERR_ARG, "block_size must be 1..65535");
    if (input_len > 0xffffffffull) return fail(SNAPPY_HIP_ERR_ARG, "container length must fit uint32");
    if (!d_stream || !d_offsets) return fail(SNAPPY_HIP_ERR_ARG, "null device pointer");
    const uint32_t nb = (uint32_t)snappy_hip_num_blocks(input_len, block_size);
    hipLaunchKernelGGL(snappy_hip::scan_block_bytes_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_block_bytes, nb,
                       (uint32_t)input_len, block_size, d_stream, d_offsets, d_stream_len);
    HIP_TRY(hipGetLastError());
    if (nb) {
        hipLaunchKernelGGL(snappy_hip::gather_slots_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, d_slots, slot_stride,
                           d_block_bytes, d_offsets, d_stream, nb);
        HIP_TRY(hipGetLastError());
    }
    return SNAPPY_HIP_OK;
}

int snappy_hip_index_streams(const snappy_hip_stream_desc* d_descs, uint32_t count, void* stream)
{
    static_assert(sizeof(snappy_hip_stream_desc) == sizeof(snappy_hip::StreamDesc), "descriptor layout");
    if (count == 0) return SNAPPY_HIP_OK;
    if (!d_descs) return fail(SNAPPY_HIP_ERR_ARG, "null descriptor array");
    hipStream_t st = (hipStream_t)stream;
    const snappy_hip::StreamDesc* dd = reinterpret_cast<const snappy_hip::StreamDesc*>(d_descs);
    // SNAPPY_HIP_INDEX_READERS=0: the walking wavefront alone, without the read-ahead workgroups on its XCD
    const uint32_t group = env_int("SNAPPY_HIP_INDEX_READERS", 1) ? snappy_hip::kIndexGroup : 1u;
    // SNAPPY_HIP_INDEX_PARALLEL=0: the serial walk only (the parallel segments resolve a stream or leave it to that walk)
    if (!env_int("SNAPPY_HIP_INDEX_PARALLEL", 1)) {
        hipLaunchKernelGGL(snappy_hip::index_streams_kernel, dim3(count * group), dim3(64 * snappy_hip::kIndexWgWaves), 0, st, dd, count,
                           group, (const uint32_t*)nullptr);
        HIP_TRY(hipGetLastError());
        return SNAPPY_HIP_OK;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(SNAPPY_HIP_ERR_ARG, "device index out of range");
    ChainWorkspace* cw = chain_workspace(dev);
    std::lock_guard<std::mutex> lock(cw->m);
    constexpr size_t kSegs = snappy_hip::kChainSegments, kCap = snappy_hip::kChainSegCap;
    const size_t words_per_stream = kSegs * (3 + kCap) + 1;
    if (cw->streams < count) {
        if (cw->last_use) HIP_TRY(hipEventSynchronize(cw->last_use));
        if (cw->mem) (void)hipFree(cw->mem);
        cw->mem = nullptr;
        cw->streams = 0;
        HIP_TRY(hipMalloc((void**)&cw->mem, (size_t)count * words_per_stream * sizeof(uint32_t)));
        cw->streams = count;
    }
    if (!cw->last_use) HIP_TRY(hipEventCreateWithFlags(&cw->last_use, hipEventDisableTiming));
    HIP_TRY(hipStreamWaitEvent(st, cw->last_use, 0));             // (an event never recorded counts as complete)
    snappy_hip::ChainWork w;
    w.anchor = cw->mem;
    w.seg_hops = w.anchor + (size_t)count * kSegs;
    w.seg_ok = w.seg_hops + (size_t)count * kSegs;
    w.hops = w.seg_ok + (size_t)count * kSegs;
    w.resolved = w.hops + (size_t)count * kSegs * kCap;
    hipLaunchKernelGGL(snappy_hip::chain_anchor_kernel, dim3(count * (uint32_t)kSegs), dim3(64), 0, st, dd, count, w);
    hipLaunchKernelGGL(snappy_hip::chain_walk_kernel, dim3(count * (uint32_t)kSegs), dim3(64), 0, st, dd, count, w);
    hipLaunchKernelGGL(snappy_hip::chain_finish_kernel, dim3(count), dim3(1024), 0, st, dd, count, w);
    hipLaunchKernelGGL(snappy_hip::index_streams_kernel, dim3(count * group), dim3(64 * snappy_hip::kIndexWgWaves), 0, st, dd, count, group,
                       (const uint32_t*)w.resolved);
    const hipError_t launched = hipGetLastError();
    HIP_TRY(hipEventRecord(cw->last_use, st));
    HIP_TRY(launched);
    return SNAPPY_HIP_OK;
}

int snappy_hip_verify_index(const snappy_hip_stream_desc* d_descs, uint32_t count, void* stream)
{
    if (count == 0) return SNAPPY_HIP_OK;
    if (!d_descs) return fail(SNAPPY_HIP_ERR_ARG, "null descriptor array");
    const snappy_hip::StreamDesc* dd = reinterpret_cast<const snappy_hip::StreamDesc*>(d_descs);
    hipLaunchKernelGGL(snappy_hip::verify_index_begin_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, dd, count);
    hipLaunchKernelGGL(snappy_hip::verify_index_kernel, dim3(count * snappy_hip::kVerifyGroup), dim3(256), 0, (hipStream_t)stream,
                       dd, count);
    HIP_TRY(hipGetLastError());
    return SNAPPY_HIP_OK;
}

// K2 over a batch of streams (count >= 1, every stream non-empty and validated by the callers below)
static int launch_decompress(const snappy_hip::K2Batch& w, uint32_t block_size, void* stream)
{
    const uint64_t nb = w.first_block[w.count];
    hipStream_t st = (hipStream_t)stream;
    // every block's status starts as "not decoded": a block the launch never reaches cannot read back as OK
    for (uint32_t c = 0; c < w.count; ++c)
        HIP_TRY(hipMemsetAsync(w.status[c], 0xff, (size_t)(w.first_block[c + 1] - w.first_block[c]) * sizeof(uint32_t), st));
    WorkCounter wc;
    if (int rc = next_work_counter(&wc, st)) return rc;
    uint32_t* counter = wc.ptr;
    const launch_shape::DeviceShape shape = device_shape();
    const uint32_t k2_cap = (uint32_t)std::max(1, env_int("SNAPPY_HIP_K2_WAVES", (int)shape.wave_slots()));   // fewer wavefronts leave slots for a co-running kernel
    if (getenv("SNAPPY_HIP_DECOMPRESS_VARIANT") || getenv("SNAPPY_HIP_K2_BATCH") || getenv("SNAPPY_HIP_K2_LDS_WAVES")) {
        (void)work_counter_launched(wc, st);
        return fail(SNAPPY_HIP_ERR_ARG, "SNAPPY_HIP_DECOMPRESS_VARIANT / SNAPPY_HIP_K2_BATCH / SNAPPY_HIP_K2_LDS_WAVES selected decoder forms of "
                                        "rounds 1-2 that were removed in round 4 (profiles/HISTORY.md)");
    }
    hipLaunchKernelGGL(snappy_hip::decompress_blocks_kernel, dim3(launch_shape::k2_launch_waves(shape, nb, (int)k2_cap)), dim3(64), 0, st, w,
                       block_size, counter);
    const hipError_t launched = hipGetLastError();
    if (int rc = work_counter_launched(wc, st)) return rc;
    HIP_TRY(launched);
    return SNAPPY_HIP_OK;
}

static int check_stream(const void* d_stream, const void* d_block_offsets, uint64_t total_len, uint32_t block_size, const void* d_out,
                        const void* d_status)
{
    if (!block_size_ok(block_size)) return fail(SNAPPY_HIP_ERR_ARG, "block_size must be 1..65535");
    if (total_len && (!d_stream || !d_block_offsets || !d_out || !d_status)) return fail(SNAPPY_HIP_ERR_ARG, "null device pointer");
    if (snappy_hip_num_blocks(total_len, block_size) > 0x7fffffffull) return fail(SNAPPY_HIP_ERR_ARG, "too many blocks");
    return SNAPPY_HIP_OK;
}

int snappy_hip_decompress_blocks(const uint8_t* d_stream, uint64_t stream_len, const uint64_t* d_block_offsets,
                                 uint64_t total_len, uint32_t block_size, uint8_t* d_out, uint32_t* d_status, void* stream)
{
    if (total_len == 0) return SNAPPY_HIP_OK;
    if (int rc = check_stream(d_stream, d_block_offsets, total_len, block_size, d_out, d_status)) return rc;
    snappy_hip::K2Batch w{};
    w.count = 1;
    w.first_block[0] = 0;
    w.first_block[1] = (uint32_t)snappy_hip_num_blocks(total_len, block_size);
    w.stream[0] = d_stream;
    w.stream_len[0] = stream_len;
    w.block_offsets[0] = d_block_offsets;
    w.total_len[0] = total_len;
    w.out[0] = d_out;
    w.status[0] = d_status;
    return launch_decompress(w, block_size, stream);
}

int snappy_hip_decompress_blocks_batch(const struct snappy_hip_decompress_item* items, uint32_t count, uint32_t block_size, void* stream)
{
    if (!items && count) return fail(SNAPPY_HIP_ERR_ARG, "null item array");
    // empty streams are skipped; a launch takes at most kMaxBatch non-empty ones, so longer lists go out in groups
    uint32_t i = 0;
    while (i < count) {
        snappy_hip::K2Batch w{};
        uint64_t blocks = 0;
        while (i < count && w.count < snappy_hip::kMaxBatch) {
            const snappy_hip_decompress_item& it = items[i];
            if (int rc = check_stream(it.d_stream, it.d_block_offsets, it.total_len, block_size, it.d_out, it.d_status)) return rc;
            const uint64_t nb = snappy_hip_num_blocks(it.total_len, block_size);
            if (nb) {
                if (blocks + nb > 0x7fffffffull) break;
                w.first_block[w.count] = (uint32_t)blocks;
                w.stream[w.count] = static_cast<const uint8_t*>(it.d_stream);
                w.stream_len[w.count] = it.stream_len;
                w.stream_len_dev[w.count] = static_cast<const uint64_t*>(it.d_stream_len);
                w.block_offsets[w.count] = static_cast<const uint64_t*>(it.d_block_offsets);
                w.total_len[w.count] = it.total_len;
                w.out[w.count] = static_cast<uint8_t*>(it.d_out);
                w.status[w.count] = static_cast<uint32_t*>(it.d_status);
                blocks += nb;
                ++w.count;
            }
            ++i;
        }
        w.first_block[w.count] = (uint32_t)blocks;
        if (w.count)
            if (int rc = launch_decompress(w, block_size, stream)) return rc;
    }
    return SNAPPY_HIP_OK;
}

}  // extern "C"

// ===========================================================================
// drop-in pair (reference L2 signatures)
// ===========================================================================

namespace {   // (C++ linkage: an unnamed namespace inside extern "C" would still export unmangled names)

// One pipeline stage unit of the overlapped drop-in pair: a contiguous run of blocks of a shard whose copy-in, kernels
// and copy-out overlap those of its neighbours.
struct CompressChunk {
    uint64_t first_block = 0, num_blocks = 0;   // relative to the shard
    uint64_t in_off = 0, in_len = 0;            // relative to the shard's input slice
    uint8_t* d_stream = nullptr;                // this chunk's own framed stream (local header + blocks)
    uint64_t *d_offsets = nullptr, *d_stream_len = nullptr;
    uint32_t local_hdr = 0;
    uint64_t stream_len = 0, out_off = 0;
    hipEvent_t ev_in = nullptr, ev_k1 = nullptr, ev_run = nullptr;
};

struct DecompressChunk {
    uint64_t first_block = 0, num_blocks = 0;   // relative to the shard
    uint64_t in_off = 0, in_len = 0;            // relative to the shard's slice of the stream
    uint64_t out_off = 0, out_len = 0;          // relative to the shard's slice of the output
    hipEvent_t ev_in = nullptr, ev_run = nullptr;
};

// The compress pipeline keeps six streams busy at once (copy-in, two K1 launches, the LDS-table helper, framing,
// copy-out).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams on
// one queue run in enqueue order: measured here, the copy-in of chunk k+1 then waits for the K1 launch of chunk k and the
// pipeline degenerates to the phased form (38 instead of 52 GB/s on a 3 GiB input).  The library does NOT touch the
// process environment: a host program that wants the overlap exports GPU_MAX_HW_QUEUES=8 before its first HIP call (the
// CLI and the Python binding do); without it only the overlap is lost, never bytes.

// Streams of one shard's pipeline.  Creating a stream costs milliseconds (a hardware queue each), so the sets are made
// once per process and shard index and kept: a long-lived caller pays for them in its first call only.
struct PipelineStreams {
    hipStream_t in = nullptr, run = nullptr, run2 = nullptr, post = nullptr, out = nullptr;
    hipEvent_t start = nullptr;
    uint64_t* h_len = nullptr;          // page-locked scratch: stream length per chunk / block offsets
    size_t h_len_count = 0;
};

// The cached streams and their page-locked scratch are per process, so overlapped calls from several host threads take
// turns (the reference's entry points are single-threaded and synchronous anyway, snappy_compress.c:618).
std::mutex* pipeline_mutex()
{
    static std::mutex* m = new std::mutex;
    return m;
}

// Cached per (device, shard): streams, events and the DMA queues behind them belong to the device they were created on, and
// the shard-to-device mapping follows the caller's current device (ShardDevices), so shard g of one call and shard g of the
// next may run on different devices; two shards on ONE device (SNAPPY_HIP_OVERSUBSCRIBE) run in different host threads at
// the same time and must not share a set either.
int pipeline_streams(int device, int shard, size_t chunks, PipelineStreams** out)
{
    static std::map<int, PipelineStreams*> cache;
    static std::mutex cache_mutex;
    if (shard < 0 || shard >= 64 || device < 0 || device >= 64) return fail(SNAPPY_HIP_ERR_ARG, "shard / device index out of range");
    PipelineStreams* pp = nullptr;
    {
        std::lock_guard<std::mutex> lock(cache_mutex);
        PipelineStreams*& slot = cache[pipeline_stream_key(device, shard)];
        if (!slot) slot = new PipelineStreams;               // never destroyed (threads may outlive statics)
        pp = slot;
    }
    PipelineStreams& p = *pp;                            // one (device, shard) is only ever touched by the host thread driving that shard
    if (!p.in) {
        HIP_TRY(hipStreamCreateWithFlags(&p.in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p.run, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p.run2, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p.post, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p.out, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&p.start));
        // a stream gets its hardware queue at first use: use each one now, in the load phase, not under the first chunk
        for (hipStream_t st : {p.in, p.run, p.run2, p.post, p.out}) {
            WorkCounter c;
            if (int rc = next_work_counter(&c, st)) return rc;
            if (int rc = work_counter_launched(c, st)) return rc;
        }
        // ... and the first asynchronous copy in either direction on a stream starts a DMA queue of its own (~8 ms)
        const size_t n = 256u << 10;
        void *h = nullptr, *d = nullptr;
        HIP_TRY(hipHostMalloc(&h, n, hipHostMallocPortable));
        HIP_TRY(hipMalloc(&d, n));
        memset(h, 0, n);
        HIP_TRY(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, p.in));
        HIP_TRY(hipStreamSynchronize(p.in));
        for (hipStream_t st : {p.out, p.post, p.run, p.run2}) HIP_TRY(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(d);
        (void)hipHostFree(h);
    }
    if (chunks > p.h_len_count) {
        if (p.h_len) (void)hipHostFree(p.h_len);
        p.h_len = nullptr;
        p.h_len_count = 0;
        const size_t want = std::max<size_t>(chunks, 64);
        HIP_TRY(hipHostMalloc((void**)&p.h_len, want * sizeof(uint64_t), hipHostMallocPortable));
        p.h_len_count = want;
    }
    *out = &p;
    return 0;
}

// SNAPPY_HIP_PIPELINE_BLOCKS: blocks per pipeline chunk of the drop-in pair (0 = strictly phased copy-in / run /
// copy-out, the reference's own order).  Default: 4096 blocks -- a K1 launch of one block per resident wavefront, and
// 128 MiB per copy -- shrinking to a quarter of the shard (not below 2048) so that a 256 MiB file still overlaps.
uint64_t pipeline_chunk_blocks(uint64_t shard_blocks, uint32_t block_size)
{
    const char* v = getenv("SNAPPY_HIP_PIPELINE_BLOCKS");
    if (v && *v) return atoi(v) > 0 ? (uint64_t)atoi(v) : 0;
    // the chunk is sized in BYTES (128 MiB, at least 64 MiB): with small blocks a chunk of 4096 blocks would be a 16 MiB
    // copy and a 0.2 ms launch, and the pipeline would be bound by launches
    const uint64_t scale = std::max<uint64_t>(1, 32768 / std::max<uint32_t>(block_size, 1));
    const uint64_t quarter = ((shard_blocks + 3) / 4 + 15) & ~15ull;
    return std::min<uint64_t>(4096 * scale, std::max<uint64_t>(2048 * scale, quarter));
}

// split `nb` blocks into equal chunks of at most `chunk` blocks, each a multiple of 16 blocks (keeps every chunk's
// input slice 16-byte aligned whatever the block size)
struct BlockRange {
    uint64_t first, second;
};
void split_blocks(uint64_t nb, uint64_t chunk, std::vector<BlockRange>& v)
{
    v.clear();
    if (!nb) return;
    const uint64_t parts = (nb + chunk - 1) / chunk;
    uint64_t per = (nb + parts - 1) / parts;
    per = (per + 15) & ~15ull;
    for (uint64_t b = 0; b < nb; b += per) v.push_back(BlockRange{b, std::min(per, nb - b)});
}

struct CompressShard {
    uint64_t first_block = 0, num_blocks = 0;
    uint64_t in_off = 0, in_len = 0;
    uint8_t *d_in = nullptr, *d_slots = nullptr, *d_stream = nullptr;
    uint32_t* d_bytes = nullptr;
    uint64_t *d_offsets = nullptr, *d_stream_len = nullptr;
    void *d_scratch = nullptr, *d_scratch2 = nullptr;
    float kernel_ms = 0.f;
    std::vector<CompressChunk> chunks;      // one chunk = the phased form
    PipelineStreams ps;
    float exposed_in_ms = 0.f;

    bool owns_anything() const { return d_in || d_slots || d_bytes || d_offsets || d_stream_len || d_scratch || d_scratch2 || d_stream; }
    // device memory and per-chunk events (the cached streams stay); safe to call twice
    void release()
    {
        for (auto& c : chunks) {
            if (c.ev_in) (void)hipEventDestroy(c.ev_in);
            if (c.ev_k1) (void)hipEventDestroy(c.ev_k1);
            if (c.ev_run) (void)hipEventDestroy(c.ev_run);
            c.ev_in = c.ev_k1 = c.ev_run = nullptr;
        }
        void** owned[] = {(void**)&d_in, (void**)&d_slots, (void**)&d_bytes, (void**)&d_offsets, (void**)&d_stream_len,
                          &d_scratch, &d_scratch2, (void**)&d_stream};
        for (void** q : owned) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
    }
};

struct DecompressShard {
    uint64_t first_block = 0, num_blocks = 0;
    uint64_t in_off = 0, in_len = 0;      // slice of the compressed stream (relative to input->buffer)
    uint64_t out_off = 0, out_len = 0;
    uint8_t *d_stream = nullptr, *d_out = nullptr;
    uint64_t* d_boff = nullptr;
    uint32_t* d_status = nullptr;
    std::vector<uint64_t> rel_off;
    float kernel_ms = 0.f;
    bool bad = false;
    std::vector<DecompressChunk> chunks;    // one chunk = the phased form
    PipelineStreams ps;
    float exposed_in_ms = 0.f;
    // size chain of the shard (snappy_decompress.c:317-340), walked on demand: blocks [0, walked) have their offsets
    // (relative to in_off) in ps.h_len[], [walked] holds the end of the last one; walk_at = stream position of block `walked`
    uint64_t walked = 0, walk_at = 0;
    bool bad_chain = false;

    bool owns_anything() const { return d_stream || d_boff || d_out || d_status; }
    void release()
    {
        for (auto& c : chunks) {
            if (c.ev_in) (void)hipEventDestroy(c.ev_in);
            if (c.ev_run) (void)hipEventDestroy(c.ev_run);
            c.ev_in = c.ev_run = nullptr;
        }
        void** owned[] = {(void**)&d_stream, (void**)&d_boff, (void**)&d_out, (void**)&d_status};
        for (void** q : owned) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
    }
};

// Error returns leave through this: drain every device a shard used, then give its memory back (the normal path has
// released everything in its timed "free" phase by then, and release() is idempotent).
struct CompressCleanup {
    std::vector<CompressShard>& shards;
    const ShardDevices& devs;
    ~CompressCleanup()
    {
        bool any = false;
        for (auto& s : shards) any = any || s.owns_anything();
        if (!any) return;
        for (size_t g = 0; g < shards.size(); ++g) {
            if (hipSetDevice(devs.device_of((int)g)) == hipSuccess) (void)hipDeviceSynchronize();
            shards[g].release();
        }
    }
};
struct DecompressCleanup {
    std::vector<DecompressShard>& shards;
    const ShardDevices& devs;
    ~DecompressCleanup()
    {
        bool any = false;
        for (auto& s : shards) any = any || s.owns_anything();
        if (!any) return;
        for (size_t g = 0; g < shards.size(); ++g) {
            if (hipSetDevice(devs.device_of((int)g)) == hipSuccess) (void)hipDeviceSynchronize();
            shards[g].release();
        }
    }
};

// extend the walk so that blocks [0, upto) are known; false = the chain leaves the stream
bool walk_chain(DecompressShard& s, uint64_t upto, const uint8_t* buf, uint64_t in_total)
{
    uint64_t* rel = s.ps.h_len;
    while (s.walked < upto) {
        if (s.walk_at + 4 > in_total) return false;
        rel[s.walked] = s.walk_at - s.in_off;
        s.walk_at += 4 + (uint64_t)le32_host(buf + s.walk_at);
        if (s.walk_at > in_total) return false;
        ++s.walked;
    }
    rel[s.walked] = s.walk_at - s.in_off;
    return true;
}

// "load" phase (dpu_load, snappy_compress.c:541): make the device ready so that the copy and run phases measure
// copies and kernels -- code object on the device, copy engines and the co-run helper stream initialised.
int warm_up_device()
{
    hipFuncAttributes fa;
    // (the default K1 launch for blocks of more than 8 KiB: the cached global-table kernel and the LDS-table kernel, stream form)
    HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(snappy_hip::compress_blocks_global_table_kernel<64, 3, 1, 512>)));
    HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(snappy_hip::compress_blocks_lds_table_kernel<64, 3>)));
    HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(snappy_hip::gather_slots_kernel)));
    HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(snappy_hip::decompress_blocks_kernel)));
    CoRunResources* cr = nullptr;
    if (int rc = corun_resources(&cr)) return rc;
    WorkCounter c;
    if (int rc = next_work_counter(&c, nullptr)) return rc;      // also touches the module's globals
    uint32_t probe = 0;
    HIP_TRY(hipMemcpy(&probe, c.ptr, sizeof(probe), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c.ptr, &probe, sizeof(probe), hipMemcpyHostToDevice));
    if (int rc = work_counter_launched(c, nullptr)) return rc;
    // the first copy of more than a few KiB in either direction starts the DMA engines (~8 ms, once per process)
    static std::mutex engines_mutex;
    static bool engines_started[64] = {};                        // per process and device, not per (short-lived) shard thread
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> engines_lock(engines_mutex);
    if (dev >= 0 && dev < 64 && !engines_started[dev]) {
        const size_t n = 1u << 20;
        void *h = nullptr, *d = nullptr;
        HIP_TRY(hipHostMalloc(&h, n, hipHostMallocPortable));
        HIP_TRY(hipMalloc(&d, n));
        memset(h, 0, n);
        HIP_TRY(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
        (void)hipFree(d);
        (void)hipHostFree(h);
        engines_started[dev] = true;
    }
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

// copy_in / run = the slowest shard's exposed copy-in and kernel time (the shards run side by side, so the phase lasts as
// long as its slowest member); copy_out = what is left of the section's wall time, so the three still add up to it.
extern "C++" template <class Shard>
void shard_phase_times(const std::vector<Shard>& sh, struct program_runtime* runtime, double wall)
{
    float in_ms = 0.f, run_ms = 0.f;
    for (const Shard& s : sh) {
        in_ms = std::max(in_ms, s.exposed_in_ms);
        run_ms = std::max(run_ms, s.kernel_ms);
    }
    runtime->copy_in = in_ms / 1000.0;
    runtime->run = run_ms / 1000.0;
    runtime->copy_out = std::max(0.0, wall - runtime->copy_in - runtime->run);
}

snappy_status report(const char* where, int rc)
{
    fprintf(stderr, "snappy_hip: %s failed: %s\n", where, g_last_error.c_str());
    (void)rc;
    return SNAPPY_INVALID_INPUT;   // the reference maps a failed launch to this (snappy_compress.c:618-623)
}


// ---------------------------------------------------------------------------
// The drop-in pair's device side (phases of snappy_compress.c:528-709 / snappy_decompress.c:306-493), overlapped per
// SURVEY section 8f row 3.  Each shard is cut into chunks of SNAPPY_HIP_PIPELINE_BLOCKS blocks; chunk k+1 is copied in
// while chunk k is compressed / decoded and chunk k-1 is framed and copied out, on separate streams.  With one chunk per
// shard this IS the reference's phased order.  The bytes do not depend on the chunking: chunks are whole blocks, blocks
// are independent (snappy_compress.c:473, :286), and the host concatenates chunk streams exactly as it concatenates
// per-device streams.  The enqueue order (kernels of k, then copy-in of k+1, then copy-out of k-1) keeps
// the overlap when the caller's buffers are pageable and hipMemcpyAsync degrades to a blocking staged copy.
// program_runtime then holds the EXPOSED parts: copy_in = until the first chunk is on the device, run = from there to
// the last kernel, copy_out = what is left of the wall time.
// ---------------------------------------------------------------------------
snappy_status compress_pipelined(struct host_buffer_context* input, struct host_buffer_context* output, uint32_t block_size,
                                 struct program_runtime* runtime, std::vector<CompressShard>& sh, int gpus, const ShardDevices& devs,
                                 const uint8_t* hdr,
                                 uint32_t hdr_len, uint32_t stride, uint64_t chunk_blocks)
{
    const uint64_t scratch_bytes = snappy_hip_compress_scratch_bytes();
    auto pad = [](uint64_t v) { return (v + 255) & ~255ull; };

    // alloc (dpu_alloc, snappy_compress.c:535)
    double t0 = now_seconds();
    int rc = for_each_device(gpus, [&](int g) -> int {
        CompressShard& s = sh[g];
        HIP_TRY(hipSetDevice(devs.device_of(g)));
        if (!s.num_blocks) return 0;
        uint64_t stream_pool = 0, offsets_pool = 0;
        std::vector<BlockRange> ranges;
        split_blocks(s.num_blocks, chunk_blocks, ranges);
        for (auto& fb : ranges) {
            CompressChunk c;
            c.first_block = fb.first;
            c.num_blocks = fb.second;
            c.in_off = c.first_block * block_size;
            c.in_len = std::min<uint64_t>(s.in_len - c.in_off, c.num_blocks * (uint64_t)block_size);
            uint8_t tmp[10];
            c.local_hdr = snappy_hip_write_header(tmp, (uint32_t)c.in_len, block_size);
            stream_pool += pad(snappy_hip_stream_bound(c.in_len, block_size));
            offsets_pool += pad((c.num_blocks + 1) * sizeof(uint64_t));
            s.chunks.push_back(c);
        }
        HIP_TRY(hipMalloc((void**)&s.d_in, s.in_len + 16));
        HIP_TRY(hipMalloc((void**)&s.d_slots, s.num_blocks * (uint64_t)stride));
        HIP_TRY(hipMalloc((void**)&s.d_bytes, s.num_blocks * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&s.d_offsets, offsets_pool));
        HIP_TRY(hipMalloc((void**)&s.d_stream_len, pad(s.chunks.size() * sizeof(uint64_t))));
        HIP_TRY(hipMalloc((void**)&s.d_stream, stream_pool));
        HIP_TRY(hipMalloc(&s.d_scratch, scratch_bytes));
        if (s.chunks.size() > 1) HIP_TRY(hipMalloc(&s.d_scratch2, scratch_bytes));
        uint64_t stream_at = 0, offsets_at = 0;
        for (size_t k = 0; k < s.chunks.size(); ++k) {
            CompressChunk& c = s.chunks[k];
            c.d_stream = s.d_stream + stream_at;
            c.d_offsets = (uint64_t*)((uint8_t*)s.d_offsets + offsets_at);
            c.d_stream_len = s.d_stream_len + k;
            stream_at += pad(snappy_hip_stream_bound(c.in_len, block_size));
            offsets_at += pad((c.num_blocks + 1) * sizeof(uint64_t));
            HIP_TRY(hipEventCreate(&c.ev_in));
            HIP_TRY(hipEventCreate(&c.ev_k1));
            HIP_TRY(hipEventCreate(&c.ev_run));
        }
        return 0;
    });
    runtime->d_alloc = now_seconds() - t0;
    if (rc) return report("device allocation", rc);

    // load (dpu_load, :541): code object, copy engines, and this shard's streams with their hardware queues
    t0 = now_seconds();
    rc = for_each_device(gpus, [&](int g) -> int {
        HIP_TRY(hipSetDevice(devs.device_of(g)));
        if (int r = warm_up_device()) return r;
        if (!sh[g].num_blocks) return 0;
        PipelineStreams* ps = nullptr;
        if (int r = pipeline_streams(devs.device_of(g), g, sh[g].chunks.size(), &ps)) return r;
        sh[g].ps = *ps;
        return 0;
    });
    runtime->load = now_seconds() - t0;
    if (rc) return report("code object load", rc);

    // the output buffer: the caller's (finite max) or ours, grown when a chunk does not fit
    const bool caller_owned = output->buffer && output->max != ~0UL;
    uint64_t capacity = caller_owned ? output->max : 0;
    if (!caller_owned) {
        capacity = 32 + input->length + input->length / 6;      // the reference's own bound (snappy_compress.c:446-449)
        uint8_t* nbuf = (uint8_t*)realloc(output->buffer, capacity);
        if (!nbuf) {
            fprintf(stderr, "snappy_hip: cannot allocate %lu bytes for the output\n", (unsigned long)capacity);
            return SNAPPY_BUFFER_TOO_SMALL;
        }
        output->buffer = nbuf;
    }
    if (capacity < hdr_len) {
        fprintf(stderr, "snappy_hip: output buffer of %lu bytes cannot hold the stream header\n", (unsigned long)capacity);
        return SNAPPY_BUFFER_TOO_SMALL;
    }
    memcpy(output->buffer, hdr, hdr_len);
    uint64_t total = hdr_len;
    bool too_small = false;
    // place chunk c of shard s at `total` and start its copy-out
    auto copy_out_chunk = [&](CompressShard& s, CompressChunk& c) -> int {
        const uint64_t body = c.stream_len - c.local_hdr;
        c.out_off = total;
        if (total + body > capacity) {
            if (caller_owned) {
                too_small = true;
                total += body;
                return 0;
            }
            // copies into the old buffer must land before it moves: every shard's copy-out stream, each on its own device
            // (this runs either on shard 0's thread inside the pipeline or on the caller's thread after the join)
            int here = 0;
            HIP_TRY(hipGetDevice(&here));
            for (int g2 = 0; g2 < gpus; ++g2) {
                if (!sh[g2].num_blocks || !sh[g2].ps.out) continue;
                HIP_TRY(hipSetDevice(devs.device_of(g2)));
                HIP_TRY(hipStreamSynchronize(sh[g2].ps.out));
            }
            HIP_TRY(hipSetDevice(here));
            capacity = std::max(total + body, capacity + capacity / 2);
            uint8_t* nbuf = (uint8_t*)realloc(output->buffer, capacity);
            if (!nbuf) return fail(SNAPPY_HIP_ERR_RUNTIME, "cannot grow the output buffer");
            output->buffer = nbuf;
        }
        if (!too_small)
            HIP_TRY(hipMemcpyAsync(output->buffer + c.out_off, c.d_stream + c.local_hdr, body, hipMemcpyDeviceToHost, s.ps.out));
        total += body;
        return 0;
    };

    // the pipeline (:547-704).  Shard 0 knows where its output goes and copies out as it runs; later shards learn their
    // place once every earlier shard has reported its lengths, and copy out after the join.
    t0 = now_seconds();
    rc = for_each_device(gpus, [&](int g) -> int {
        CompressShard& s = sh[g];
        HIP_TRY(hipSetDevice(devs.device_of(g)));
        if (!s.num_blocks) return 0;
        const size_t n = s.chunks.size();
        auto copy_in = [&](size_t k) -> int {
            CompressChunk& c = s.chunks[k];
            HIP_TRY(hipMemcpyAsync(s.d_in + c.in_off, input->buffer + s.in_off + c.in_off, c.in_len, hipMemcpyHostToDevice, s.ps.in));
            HIP_TRY(hipEventRecord(c.ev_in, s.ps.in));
            return 0;
        };
        auto finish = [&](size_t k) -> int {
            CompressChunk& c = s.chunks[k];
            HIP_TRY(hipEventSynchronize(c.ev_run));
            c.stream_len = s.ps.h_len[k];
            return g == 0 ? copy_out_chunk(s, c) : 0;
        };
        HIP_TRY(hipEventRecord(s.ps.start, s.ps.in));
        if (int r = copy_in(0)) return r;
        for (size_t k = 0; k < n; ++k) {
            CompressChunk& c = s.chunks[k];
            // Two launches in flight, on alternating streams with a hash-table scratch each: a launch of one block per
            // wavefront ends in a tail of half-empty CUs, which the next chunk's wavefronts fill.
            hipStream_t run = (k & 1) ? s.ps.run2 : s.ps.run;
            void* scratch = (k & 1) ? s.d_scratch2 : s.d_scratch;
            HIP_TRY(hipStreamWaitEvent(run, c.ev_in, 0));
            int r = snappy_hip_compress_blocks(s.d_in + c.in_off, c.in_len, block_size, s.d_slots + c.first_block * (uint64_t)stride,
                                               stride, s.d_bytes + c.first_block, scratch, scratch_bytes, run);
            if (r) return r;
            // scan + gather on a stream of their own: small kernels that crawl beside the next chunk's K1 must not
            // hold back the K1 launch after that
            HIP_TRY(hipEventRecord(c.ev_k1, run));
            HIP_TRY(hipStreamWaitEvent(s.ps.post, c.ev_k1, 0));
            r = snappy_hip_compact(s.d_slots + c.first_block * (uint64_t)stride, stride, s.d_bytes + c.first_block, c.in_len, block_size,
                                   c.d_stream, c.d_offsets, c.d_stream_len, s.ps.post);
            if (r) return r;
            HIP_TRY(hipMemcpyAsync(&s.ps.h_len[k], c.d_stream_len, sizeof(uint64_t), hipMemcpyDeviceToHost, s.ps.post));
            HIP_TRY(hipEventRecord(c.ev_run, s.ps.post));
            if (k + 1 < n)
                if (int r2 = copy_in(k + 1)) return r2;
            if (k >= 1)
                if (int r2 = finish(k - 1)) return r2;
        }
        if (int r = finish(n - 1)) return r;
        HIP_TRY(hipStreamSynchronize(s.ps.out));
        HIP_TRY(hipEventElapsedTime(&s.exposed_in_ms, s.ps.start, s.chunks[0].ev_in));
        HIP_TRY(hipEventElapsedTime(&s.kernel_ms, s.chunks[0].ev_in, s.chunks[n - 1].ev_run));
        if (env_int("SNAPPY_HIP_PIPELINE_TRACE", 0))
            for (size_t k = 0; k < n; ++k) {
                float a = 0.f, b = 0.f, c = 0.f;
                HIP_TRY(hipEventElapsedTime(&a, s.ps.start, s.chunks[k].ev_in));
                HIP_TRY(hipEventElapsedTime(&b, s.ps.start, s.chunks[k].ev_k1));
                HIP_TRY(hipEventElapsedTime(&c, s.ps.start, s.chunks[k].ev_run));
                fprintf(stderr, "chunk %zu: copied in at %.2f ms, compressed at %.2f ms, framed at %.2f ms\n", k, a, b, c);
            }
        if (n >= 2) {                                   // the second-to-last launch runs on the other stream and may end later
            float other = 0.f;
            HIP_TRY(hipEventElapsedTime(&other, s.chunks[0].ev_in, s.chunks[n - 2].ev_run));
            s.kernel_ms = std::max(s.kernel_ms, other);
        }
        return 0;
    });
    if (rc) return report("compress pipeline", rc);
    if (gpus > 1) {
        for (int g = 1; g < gpus && !rc; ++g) {
            CompressShard& s = sh[g];
            if (!s.num_blocks) continue;
            if ((rc = (int)hipSetDevice(devs.device_of(g)))) break;
            for (auto& c : s.chunks)
                if ((rc = copy_out_chunk(s, c))) break;
        }
        if (!rc)
            rc = for_each_device(gpus, [&](int g) -> int {
                HIP_TRY(hipSetDevice(devs.device_of(g)));
                if (g && sh[g].num_blocks) HIP_TRY(hipStreamSynchronize(sh[g].ps.out));
                return 0;
            });
        if (rc) return report("device-to-host copy", rc);
    }
    const double wall = now_seconds() - t0;
    shard_phase_times(sh, runtime, wall);

    for (int g = 0; g < gpus; ++g)   // analogue of the per-tasklet log lines (dpu-compress/dpu_task.c:88)
        printf("GPU %d: %f s, %lu bytes\n", g, sh[g].kernel_ms / 1000.0, (unsigned long)sh[g].in_len);

    // free (dpu_free, :707)
    t0 = now_seconds();
    rc = for_each_device(gpus, [&](int g) -> int {
        CompressShard& s = sh[g];
        HIP_TRY(hipSetDevice(devs.device_of(g)));
        s.release();
        return 0;
    });
    runtime->d_free = now_seconds() - t0;
    if (rc) return report("free", rc);
    if (too_small) {
        fprintf(stderr, "snappy_hip: output buffer of %lu bytes cannot hold the %lu-byte stream\n", (unsigned long)output->max,
                (unsigned long)total);
        return SNAPPY_BUFFER_TOO_SMALL;
    }
    if (!caller_owned) {
        uint8_t* nbuf = (uint8_t*)realloc(output->buffer, total ? total : 1);
        if (nbuf) output->buffer = nbuf;
    }
    output->length = total;
    output->curr = output->buffer + total;
    return SNAPPY_OK;
}

// Decompress counterpart: sizes are known from the host pre-scan, so the whole pipeline is enqueued without a host
// round trip; chunk k's plaintext goes straight into its range of output->buffer (snappy_decompress.c:463).
snappy_status decompress_pipelined(const uint8_t* buf, uint64_t in_total, struct host_buffer_context* output,
                                   struct program_runtime* runtime, std::vector<DecompressShard>& sh, int gpus, const ShardDevices& devs,
                                   uint32_t bs,
                                   uint64_t total, uint64_t chunk_blocks)
{
    double t0 = now_seconds();
    int rc = for_each_device(gpus, [&](int g) -> int {
        DecompressShard& s = sh[g];
        HIP_TRY(hipSetDevice(devs.device_of(g)));
        if (!s.num_blocks) return 0;
        std::vector<BlockRange> ranges;
        split_blocks(s.num_blocks, chunk_blocks, ranges);
        for (auto& fb : ranges) {
            DecompressChunk c;
            c.first_block = fb.first;
            c.num_blocks = fb.second;
            c.out_off = c.first_block * bs;
            c.out_len = std::min<uint64_t>(s.out_len - c.out_off, c.num_blocks * (uint64_t)bs);
            HIP_TRY(hipEventCreate(&c.ev_in));
            HIP_TRY(hipEventCreate(&c.ev_run));
            s.chunks.push_back(c);
        }
        HIP_TRY(hipMalloc((void**)&s.d_stream, s.in_len + 16));
        HIP_TRY(hipMalloc((void**)&s.d_boff, s.num_blocks * sizeof(uint64_t)));
        HIP_TRY(hipMalloc((void**)&s.d_out, s.out_len + 16));
        HIP_TRY(hipMalloc((void**)&s.d_status, s.num_blocks * sizeof(uint32_t)));
        return 0;
    });
    runtime->d_alloc = now_seconds() - t0;
    if (rc) return report("device allocation", rc);

    t0 = now_seconds();
    rc = for_each_device(gpus, [&](int g) -> int {
        DecompressShard& s = sh[g];
        HIP_TRY(hipSetDevice(devs.device_of(g)));
        if (int r = warm_up_device()) return r;
        if (!s.num_blocks) return 0;
        PipelineStreams* ps = nullptr;
        if (int r = pipeline_streams(devs.device_of(g), g, s.num_blocks + 1, &ps)) return r;   // page-locked home of the block offsets
        s.ps = *ps;
        if (!s.rel_off.empty()) {                                           // chain already walked by the caller
            memcpy(s.ps.h_len, s.rel_off.data(), s.num_blocks * sizeof(uint64_t));
            s.ps.h_len[s.num_blocks] = s.in_len;
            s.walked = s.num_blocks;
            s.walk_at = s.in_off + s.in_len;
        }
        return 0;
    });
    runtime->load = now_seconds() - t0;
    if (rc) return report("code object load", rc);

    t0 = now_seconds();
    rc = for_each_device(gpus, [&](int g) -> int {
        DecompressShard& s = sh[g];
        HIP_TRY(hipSetDevice(devs.device_of(g)));
        if (!s.num_blocks) return 0;
        const size_t n = s.chunks.size();
        const uint64_t* rel = s.ps.h_len;
        auto copy_out = [&](size_t k) -> int {
            DecompressChunk& c = s.chunks[k];
            HIP_TRY(hipStreamWaitEvent(s.ps.out, c.ev_run, 0));
            HIP_TRY(hipMemcpyAsync(output->buffer + s.out_off + c.out_off, s.d_out + c.out_off, c.out_len, hipMemcpyDeviceToHost, s.ps.out));
            return 0;
        };
        HIP_TRY(hipEventRecord(s.ps.start, s.ps.in));
        size_t issued = 0;
        for (size_t k = 0; k < n; ++k) {
            DecompressChunk& c = s.chunks[k];
            // the host walks this chunk's part of the size chain while the previous chunk is still being copied in
            if (!walk_chain(s, c.first_block + c.num_blocks, buf, in_total)) {
                s.bad_chain = true;
                break;
            }
            c.in_off = rel[c.first_block];
            c.in_len = rel[c.first_block + c.num_blocks] - c.in_off;
            HIP_TRY(hipMemcpyAsync(s.d_boff + c.first_block, rel + c.first_block, c.num_blocks * sizeof(uint64_t), hipMemcpyHostToDevice,
                                   s.ps.in));
            HIP_TRY(hipMemcpyAsync(s.d_stream + c.in_off, buf + s.in_off + c.in_off, c.in_len, hipMemcpyHostToDevice, s.ps.in));
            HIP_TRY(hipEventRecord(c.ev_in, s.ps.in));
            HIP_TRY(hipStreamWaitEvent(s.ps.run, c.ev_in, 0));
            // block i of the chunk is read at d_stream + d_boff[first + i] (offsets stay relative to the shard's slice)
            int r = snappy_hip_decompress_blocks(s.d_stream, s.in_len, s.d_boff + c.first_block, c.out_len, bs, s.d_out + c.out_off,
                                                 s.d_status + c.first_block, s.ps.run);
            if (r) return r;
            HIP_TRY(hipEventRecord(c.ev_run, s.ps.run));
            ++issued;
            if (k >= 1)
                if (int r2 = copy_out(k - 1)) return r2;
        }
        if (!s.bad_chain && s.walk_at != s.in_off + s.in_len) s.bad_chain = true;    // the chain must end where the slice ends
        if (issued && !s.bad_chain)
            if (int r = copy_out(issued - 1)) return r;
        HIP_TRY(hipStreamSynchronize(s.ps.in));
        HIP_TRY(hipStreamSynchronize(s.ps.run));
        HIP_TRY(hipStreamSynchronize(s.ps.out));
        if (s.bad_chain || !issued) return 0;
        std::vector<uint32_t> st(s.num_blocks);
        HIP_TRY(hipMemcpy(st.data(), s.d_status, s.num_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < s.num_blocks; ++i)
            if (st[i] != SNAPPY_HIP_BLOCK_OK) s.bad = true;
        HIP_TRY(hipEventElapsedTime(&s.exposed_in_ms, s.ps.start, s.chunks[0].ev_in));
        HIP_TRY(hipEventElapsedTime(&s.kernel_ms, s.chunks[0].ev_in, s.chunks[n - 1].ev_run));
        return 0;
    });
    const double wall = now_seconds() - t0;
    if (rc) return report("decompress pipeline", rc);
    shard_phase_times(sh, runtime, wall);

    for (int g = 0; g < gpus; ++g)
        printf("GPU %d: %f s, %lu bytes\n", g, sh[g].kernel_ms / 1000.0, (unsigned long)sh[g].in_len);

    t0 = now_seconds();
    rc = for_each_device(gpus, [&](int g) -> int {
        DecompressShard& s = sh[g];
        HIP_TRY(hipSetDevice(devs.device_of(g)));
        s.release();
        return 0;
    });
    runtime->d_free = now_seconds() - t0;
    if (rc) return report("free", rc);
    for (auto& s : sh) {
        if (s.bad_chain) {
            fprintf(stderr, "snappy_hip: size chain leaves the stream (block %lu of %lu)\n",
                    (unsigned long)(s.first_block + s.walked), (unsigned long)(s.first_block + s.num_blocks));
            return SNAPPY_INVALID_INPUT;
        }
        if (s.bad) {
            fprintf(stderr, "snappy_hip: malformed block in the stream\n");
            return SNAPPY_INVALID_INPUT;
        }
    }
    output->curr = output->buffer + total;
    return SNAPPY_OK;
}

}  // namespace

extern "C" {

static snappy_status compress_gpu_body(struct host_buffer_context* input, struct host_buffer_context* output, uint32_t block_size,
                                       struct program_runtime* runtime)
{
    double t0 = now_seconds();
    if (!input || !output || !runtime) return SNAPPY_INVALID_INPUT;
    if (input->length && !input->buffer) return SNAPPY_INVALID_INPUT;
    runtime->d_alloc = runtime->load = runtime->copy_in = runtime->run = runtime->copy_out = runtime->d_free = 0.0;
    if (!block_size_ok(block_size)) {
        fprintf(stderr, "snappy_hip: block size %u is outside 1..65535 (16-bit hash table entries)\n", block_size);
        return SNAPPY_INVALID_INPUT;
    }
    if (input->length > 0xffffffffull) {
        fprintf(stderr, "snappy_hip: input of %lu bytes does not fit the format's uint32 length\n", input->length);
        return SNAPPY_BUFFER_TOO_SMALL;
    }
    const uint64_t n = input->length;
    const uint64_t nb = snappy_hip_num_blocks(n, block_size);
    const ShardDevices devs = requested_devices();
    int gpus = devs.shards;
    if (gpus <= 0) {
        fprintf(stderr, "snappy_hip: no HIP device available; the -d path has no CPU fallback\n");
        return SNAPPY_INVALID_INPUT;
    }
    if ((uint64_t)gpus > nb) gpus = nb ? (int)nb : 1;

    // partition: contiguous block ranges per device (snappy_compress.c:494-520)
    const uint64_t per = nb ? (nb + gpus - 1) / gpus : 0;
    std::vector<CompressShard> sh(gpus);
    CompressCleanup cleanup{sh, devs};
    for (int g = 0; g < gpus; ++g) {
        sh[g].first_block = (uint64_t)g * per;
        const uint64_t last = std::min(nb, sh[g].first_block + per);
        sh[g].num_blocks = last > sh[g].first_block ? last - sh[g].first_block : 0;
        sh[g].in_off = sh[g].first_block * block_size;
        sh[g].in_len = sh[g].num_blocks ? std::min<uint64_t>(n - sh[g].in_off, sh[g].num_blocks * (uint64_t)block_size) : 0;
    }
    uint8_t hdr[10];
    const uint32_t hdr_len = snappy_hip_write_header(hdr, (uint32_t)n, block_size);   // :523-525
    const uint32_t stride = snappy_hip_slot_stride(block_size);
    runtime->pre += now_seconds() - t0;
    // One code path: a shard is a list of chunks; SNAPPY_HIP_PIPELINE_BLOCKS=0 (or a small shard) makes it one chunk, which
    // is the strictly phased copy-in / run / copy-out of the reference (snappy_compress.c:547-704).
    uint64_t chunk_blocks = pipeline_chunk_blocks(per, block_size);
    if (!chunk_blocks || per <= chunk_blocks) chunk_blocks = std::max<uint64_t>(per, 1);
    return compress_pipelined(input, output, block_size, runtime, sh, gpus, devs, hdr, hdr_len, stride, chunk_blocks);
}

static snappy_status decompress_gpu_body(struct host_buffer_context* input, struct host_buffer_context* output,
                                         struct program_runtime* runtime)
{
    double t0 = now_seconds();
    if (!input || !output || !runtime) return SNAPPY_INVALID_INPUT;
    if (!input->buffer || !input->curr || input->curr < input->buffer) return SNAPPY_INVALID_INPUT;
    runtime->d_alloc = runtime->load = runtime->copy_in = runtime->run = runtime->copy_out = runtime->d_free = 0.0;

    // block-size varint (snappy_decompress.c:298-303)
    const uint8_t* const buf = input->buffer;
    const uint64_t in_total = input->length;
    uint64_t at = (uint64_t)(input->curr - input->buffer);
    uint32_t bs = 0;
    const uint32_t used = (at <= in_total) ? get_varint32(buf + at, in_total - at, &bs) : 0;
    if (!used) {
        fprintf(stderr, "Failed to read decompressed block size\n");
        return SNAPPY_INVALID_INPUT;
    }
    at += used;
    input->curr += used;
    const uint64_t total = output->length;
    if (total == 0) {
        runtime->pre += now_seconds() - t0;
        return (at == in_total) ? SNAPPY_OK : SNAPPY_INVALID_INPUT;
    }
    if (!block_size_ok(bs)) {
        fprintf(stderr, "snappy_hip: block size %u in the stream is outside 1..65535\n", bs);
        return SNAPPY_INVALID_INPUT;
    }
    if (!output->buffer) {
        fprintf(stderr, "snappy_hip: output->buffer is NULL (setup_decompression allocates it, snappy_decompress.c:207-209)\n");
        return SNAPPY_INVALID_INPUT;
    }
    const uint64_t nb = snappy_hip_num_blocks(total, bs);
    // the header is untrusted: every block needs at least its u32 size prefix, so a stream of in_total - at bytes cannot
    // hold more than (in_total - at) / 4 blocks -- checked before anything is sized by nb
    if (nb > (in_total - at) / 4) {
        fprintf(stderr, "snappy_hip: header promises %lu blocks, the stream has room for %lu\n", (unsigned long)nb,
                (unsigned long)((in_total - at) / 4));
        return SNAPPY_INVALID_INPUT;
    }
    const ShardDevices devs = requested_devices();
    int gpus = devs.shards;
    if (gpus <= 0) {
        fprintf(stderr, "snappy_hip: no HIP device available; the -d path has no CPU fallback\n");
        return SNAPPY_INVALID_INPUT;
    }
    if ((uint64_t)gpus > nb) gpus = (int)nb;
    // A decode launch takes about as long for 2048 blocks as for 8192 (one block per wavefront either way), so the
    // overlapped form pays from three chunks per shard upwards.
    const uint64_t chunk_blocks = pipeline_chunk_blocks((nb + gpus - 1) / gpus, bs);
    const bool overlapped = chunk_blocks && (nb + gpus - 1) / gpus >= 3 * chunk_blocks;
    if (overlapped && gpus == 1) {
        // one shard: the host walks the size chain chunk by chunk inside the pipeline instead of up front
        std::vector<DecompressShard> one(1);
        DecompressCleanup cleanup{one, devs};
        one[0].num_blocks = nb;
        one[0].in_off = one[0].walk_at = at;
        one[0].in_len = in_total - at;
        one[0].out_len = total;
        runtime->pre += now_seconds() - t0;
        return decompress_pipelined(buf, in_total, output, runtime, one, 1, devs, bs, total, chunk_blocks);
    }
    // host pre-scan of the size chain (:306-341): the blocks are split over the devices before anything is copied, so every
    // offset is needed first.  In parallel shares where the stream is long enough (csrc/host_chain.hpp: exact by
    // construction, ~7 ms per GiB of serial pointer chase otherwise -- the whole of `pre`, and it does not shrink with the
    // number of devices); SNAPPY_HIP_HOST_WALK_THREADS=1 is the serial walk alone, which also names a damaged stream's fault.
    std::vector<uint64_t> off;
    const unsigned walk_threads = (unsigned)std::max(1, env_int("SNAPPY_HIP_HOST_WALK_THREADS",
                                                               (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()))));
    if (!host_chain::parallel_walk(buf, in_total, at, nb, bs, walk_threads, off)) {
        off.assign(nb + 1, 0);
        for (uint64_t i = 0; i < nb; ++i) {
            if (at + 4 > in_total) {
                fprintf(stderr, "snappy_hip: truncated stream (block %lu of %lu)\n", (unsigned long)i, (unsigned long)nb);
                return SNAPPY_INVALID_INPUT;
            }
            off[i] = at;
            at += 4 + (uint64_t)le32_host(buf + at);
        }
        off[nb] = at;
        if (at != in_total) {
            fprintf(stderr, "snappy_hip: size chain ends at %lu, stream has %lu bytes\n", (unsigned long)at, (unsigned long)in_total);
            return SNAPPY_INVALID_INPUT;
        }
    }
    const uint64_t per = (nb + gpus - 1) / gpus;
    std::vector<DecompressShard> sh(gpus);
    DecompressCleanup cleanup{sh, devs};
    for (int g = 0; g < gpus; ++g) {
        DecompressShard& s = sh[g];
        s.first_block = (uint64_t)g * per;
        const uint64_t last = std::min(nb, s.first_block + per);
        s.num_blocks = last > s.first_block ? last - s.first_block : 0;
        if (!s.num_blocks) continue;
        s.in_off = off[s.first_block];
        s.in_len = off[last] - s.in_off;
        s.out_off = s.first_block * bs;
        s.out_len = std::min<uint64_t>(total - s.out_off, s.num_blocks * (uint64_t)bs);
        s.rel_off.resize(s.num_blocks);
        for (uint64_t i = 0; i < s.num_blocks; ++i) s.rel_off[i] = off[s.first_block + i] - s.in_off;
    }
    runtime->pre += now_seconds() - t0;
    // one chunk per shard = the strictly phased form (the size chain was walked above, in `pre`)
    return decompress_pipelined(buf, in_total, output, runtime, sh, gpus, devs, bs, total, overlapped ? chunk_blocks : std::max<uint64_t>(per, 1));
}

// The exported pair: one call at a time per process (the cached pipeline streams and their page-locked scratch are per
// process; the reference's entry points are single-threaded and synchronous anyway, snappy_compress.c:618), the caller's
// current HIP device restored on every return path, and no C++ exception crosses the C boundary.
snappy_status snappy_compress_gpu(struct host_buffer_context* input, struct host_buffer_context* output, uint32_t block_size,
                                  struct program_runtime* runtime)
{
    try {
        std::lock_guard<std::mutex> one_at_a_time(*pipeline_mutex());
        CallerDevice keep;
        return compress_gpu_body(input, output, block_size, runtime);
    } catch (const std::bad_alloc&) {
        fprintf(stderr, "snappy_hip: out of host memory\n");
        return SNAPPY_BUFFER_TOO_SMALL;
    } catch (const std::exception& e) {
        fprintf(stderr, "snappy_hip: %s\n", e.what());
        return SNAPPY_INVALID_INPUT;
    } catch (...) {
        return SNAPPY_INVALID_INPUT;
    }
}

snappy_status snappy_decompress_gpu(struct host_buffer_context* input, struct host_buffer_context* output,
                                    struct program_runtime* runtime)
{
    try {
        std::lock_guard<std::mutex> one_at_a_time(*pipeline_mutex());
        CallerDevice keep;
        return decompress_gpu_body(input, output, runtime);
    } catch (const std::bad_alloc&) {
        fprintf(stderr, "snappy_hip: out of host memory\n");
        return SNAPPY_BUFFER_TOO_SMALL;
    } catch (const std::exception& e) {
        fprintf(stderr, "snappy_hip: %s\n", e.what());
        return SNAPPY_INVALID_INPUT;
    } catch (...) {
        return SNAPPY_INVALID_INPUT;
    }
}

#ifdef SNAPPY_PAIR_PROBE
int snappy_hip_debug_pair_prof(unsigned long long* out, int reset)
{
    if (reset) {
        unsigned long long z[16] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(snappy_hip::g_pair_prof), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(snappy_hip::g_pair_prof), 16 * sizeof(unsigned long long));
}
#endif

}  // extern "C"
