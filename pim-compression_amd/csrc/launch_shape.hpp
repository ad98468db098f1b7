// launch_shape.hpp -- how many wavefronts of which kind a K1 / K2 launch gets on a device of a given shape.
// Plain arithmetic, no HIP: tests/test_launch_shape.py compiles it on the CPU.
//
// The shape of the device -- compute units, LDS per CU, wavefront slots per CU -- is read once per device with
// hipGetDeviceProperties (snappy_hip.hip: device_shape()); nothing here assumes a whole MI355X (256 CUs): a partition of
// it (CPX / NPS modes: 32 CUs per logical device) gets a grid and a hash-table scratch of its own size.
#pragma once
#include <stdint.h>
#include <algorithm>

namespace launch_shape {

struct DeviceShape {
    uint32_t cus = 256;                   // hipDeviceProp_t::multiProcessorCount
    uint32_t lds_per_cu = 160u << 10;     // maxSharedMemoryPerMultiProcessor
    uint32_t wave_slots_per_cu = 32;      // maxThreadsPerMultiProcessor / 64
    uint32_t wave_slots() const { return cus * wave_slots_per_cu; }
    uint32_t simds_per_cu() const { return std::max(1u, wave_slots_per_cu / 8u); }   // 8 wavefront slots per SIMD on CDNA
};

constexpr uint32_t kMaxTableEntries = 16384;            // snappy_compress.c:16-17
// LDS is handed out in blocks; budgets round every allocation up to 1 KiB (a multiple of every granule CDNA parts have used)
inline uint32_t lds_alloc_bytes(uint32_t bytes) { return (bytes + 1023u) & ~1023u; }
// u16 table entries an LDS-table wavefront reserves for blocks of up to block_size bytes (get_hash_table, :139-146)
inline uint32_t lds_table_entries(uint32_t block_size)
{
    uint32_t ts = 256;
    while (ts < kMaxTableEntries && ts < block_size) ts <<= 1;
    return ts;
}

// hash-table scratch of the global-table kernels: a 256-byte header (work counter + statistics) + one 64 KiB table per
// wavefront slot of the device
inline uint64_t compress_scratch_bytes(const DeviceShape& d) { return 256 + (uint64_t)d.wave_slots() * kMaxTableEntries * sizeof(uint32_t); }

// what the caller has decided about the kernels (block-size dependent defaults and environment overrides, snappy_hip.hip)
struct K1Knobs {
    uint32_t lds_wave_bytes = 0;          // dynamic LDS of one LDS-table workgroup (table + scratch), not yet rounded
    uint32_t lds_wave_slots = 1;          // wavefront slots one LDS-table workgroup takes
    uint32_t gt_wave_bytes = 0;           // static LDS of one global-table wavefront (filter, slot cache, duplicate test)
    bool cached_global_table = false;     // the slot cache is in front of the global table (blocks of more than 8 KiB)
    int lds_waves_forced = -1;            // SNAPPY_HIP_LDS_WAVES (LDS-table wavefronts of the launch), -1 = default
    int waves_forced = -1;                // SNAPPY_HIP_GT_WAVES (with the default launch: the TOTAL of both kinds), -1 = default
    uint64_t hybrid_min_blocks = 4096;    // SNAPPY_HIP_HYBRID_MIN_BLOCKS: below it one kernel is enough
};

// LDS-table wavefronts per CU of the default launch (occupancy follows the table the block size needs).
// Beside cached global-table wavefronts ONE per CU is the measured optimum (profiles/r03_gt_cache_sweep.txt); tables of more
// than 24 KiB without the cache: three per CU (round 2's mix); small tables: as many as fit beside at least 8 global-table
// wavefronts per CU (4 KiB each at most), at most 3/4 of the wavefront slots.
inline uint32_t default_lds_waves_per_cu(const DeviceShape& d, const K1Knobs& k)
{
    if (k.cached_global_table) return 1;
    const uint32_t per_wave = lds_alloc_bytes(k.lds_wave_bytes);
    if (per_wave > (24u << 10)) return std::min(3u, std::max(1u, d.lds_per_cu / per_wave));
    const uint32_t room = d.lds_per_cu > 8u * (4u << 10) ? d.lds_per_cu - 8u * (4u << 10) : 0u;
    return std::min<uint32_t>(d.wave_slots_per_cu * 3u / 4u, room / std::max(1u, per_wave));
}

// A small input -- every block can have an LDS-table wavefront at once -- goes to the LDS-table kernel alone: its wavefronts
// run the stream form with (almost) nothing else on their SIMD.  Measured on both sides of the cut-over
// (profiles/r04_small_inputs_threshold.txt): 32 KiB blocks (36 KiB of LDS per wavefront: four per CU, one per SIMD) -- 1,024
// blocks 1.09 ms against 1.59 ms for the global-table kernel alone, 1,100 blocks 2.08 against 1.74; 8 KiB blocks (20 KiB: eight
// per CU, two per SIMD) -- 2,048 blocks 0.39 against 0.47 ms, 2,571 blocks 0.65 against 0.54.  So: as many blocks as LDS-table
// wavefronts fit at once, at most two per SIMD.
inline bool small_input_takes_lds_kernel_alone(const DeviceShape& d, uint32_t stream_lds_wave_bytes, uint64_t num_blocks)
{
    const uint32_t per_cu = std::min<uint32_t>(2u * d.simds_per_cu(), d.lds_per_cu / std::max(1u, lds_alloc_bytes(stream_lds_wave_bytes)));
    return num_blocks <= (uint64_t)per_cu * d.cus;
}

struct K1Launch {
    uint32_t lds_waves = 0;               // workgroups of the LDS-table kernel (helper stream)
    uint32_t gt_waves = 0;                // wavefronts of the global-table kernel
};

// The default K1 launch: persistent grids drawing blocks from one counter.  Wave budget per CU: the LDS-table workgroups
// first (LDS capacity), then as many global-table wavefronts as the remaining LDS and wavefront slots hold.
inline K1Launch k1_default_launch(const DeviceShape& d, const K1Knobs& k, uint64_t num_blocks)
{
    uint32_t lds_waves = k.lds_waves_forced >= 0 ? (uint32_t)k.lds_waves_forced : default_lds_waves_per_cu(d, k) * d.cus;
    if (num_blocks < k.hybrid_min_blocks) lds_waves = 0;
    const uint32_t lds_per_cu = (lds_waves + d.cus - 1) / d.cus;
    const uint32_t lds_wave_bytes = lds_alloc_bytes(k.lds_wave_bytes);
    const uint32_t lds_slots = lds_per_cu * k.lds_wave_slots;
    uint32_t g_per_cu = d.wave_slots_per_cu > lds_slots ? d.wave_slots_per_cu - lds_slots : 0u;
    if (k.gt_wave_bytes && lds_per_cu * lds_wave_bytes < d.lds_per_cu)
        g_per_cu = std::min(g_per_cu, (d.lds_per_cu - lds_per_cu * lds_wave_bytes) / lds_alloc_bytes(k.gt_wave_bytes));
    uint32_t waves = k.waves_forced >= 0 ? (uint32_t)k.waves_forced : lds_waves + g_per_cu * d.cus;
    waves = std::min(waves, d.wave_slots());
    if (lds_waves >= waves) lds_waves = waves / 2;
    K1Launch l;
    l.lds_waves = lds_waves;
    l.gt_waves = (uint32_t)std::min<uint64_t>(num_blocks, waves - lds_waves);
    return l;
}

// K2: one wavefront per slot of the device at most (SNAPPY_HIP_K2_WAVES caps it below that)
inline uint32_t k2_launch_waves(const DeviceShape& d, uint64_t num_blocks, int cap_forced)
{
    const uint32_t resident = d.wave_slots();
    const uint32_t cap = cap_forced > 0 ? (uint32_t)cap_forced : resident;
    return (uint32_t)std::min<uint64_t>(std::min<uint64_t>(num_blocks, cap), resident);
}

}  // namespace launch_shape
