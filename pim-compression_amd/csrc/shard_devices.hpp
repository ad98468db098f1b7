// shard_devices.hpp -- which device a shard of the drop-in pair runs on, and the key its cached streams are kept under.
// Plain arithmetic, no HIP: tests/test_sharding_gloo.py compiles it on the CPU.
//
// The drop-in pair (snappy_compress_gpu / snappy_decompress_gpu) splits a file's blocks into contiguous ranges, one per
// shard (snappy_compress.c:494-520).  Shard 0 runs on the caller's current device, shard g on (base + g) % physical, so a
// caller that has selected device 3 of 8 keeps its data path on device 3 for a one-shard call.  This is per-call state:
// two calls with different current devices map their shards differently, and everything that belongs to a device
// (streams, events, DMA queues, work counters) must be looked up by the DEVICE, not by the shard number alone.
#pragma once

namespace {

struct ShardDevices {
    int physical = 1;       // devices visible to the process
    int base = 0;           // the caller's current device when the pair was entered
    int shards = 0;         // shards requested (0: no usable device)
    int device_of(int shard) const { return (base + shard) % physical; }
};

// cache key of a shard's stream set: per (device, shard) -- two shards on ONE device (SNAPPY_HIP_OVERSUBSCRIBE) run in
// different host threads at the same time and must not share a set either
inline int pipeline_stream_key(int device, int shard) { return device * 64 + shard; }

}  // namespace
