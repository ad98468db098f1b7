// Micro-benchmark (gfx950): random load+store pairs (one hash-table probe) per second as a function of the number of
// wavefronts and of the slice each of them owns -- i.e. of the total table footprint against L2 (8 x 4 MiB) and the
// 256 MiB memory-side cache.  Build: hipcc --offload-arch=gfx950 -O3 -o footprint_probe footprint_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename Entry>
__global__ __launch_bounds__(64) void probe(Entry* __restrict__ buf, uint32_t iters, uint32_t slice_entries, uint32_t* __restrict__ out)
{
    Entry* t = buf + (size_t)blockIdx.x * slice_entries;
    uint32_t s = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = (s >> 10) & (slice_entries - 1);
        acc += t[idx];
        t[idx] = (Entry)s;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main()
{
    void* d_buf;
    uint32_t* d_out;
    CHECK(hipMalloc(&d_buf, (size_t)8192 * 65536));
    CHECK(hipMalloc(&d_out, 8192 * 64 * 4));
    CHECK(hipMemset(d_buf, 0, (size_t)8192 * 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int bytes : {4, 2})
        for (uint32_t waves : {512u, 1024u, 2048u, 3072u, 4096u, 5120u, 8192u}) {
            const uint32_t iters = 2000;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (bytes == 4) hipLaunchKernelGGL(probe<uint32_t>, dim3(waves), dim3(64), 0, 0, (uint32_t*)d_buf, iters, 16384u, d_out);
                else hipLaunchKernelGGL(probe<uint16_t>, dim3(waves), dim3(64), 0, 0, (uint16_t*)d_buf, iters, 16384u, d_out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("entry %d B  waves %5u  footprint %4u MiB: %8.2f ms  %7.2f G probes/s\n", bytes, waves,
                   (unsigned)((size_t)waves * 16384 * bytes >> 20), best, (double)waves * iters * 64 / best / 1e6);
            fflush(stdout);
        }
    return 0;
}
