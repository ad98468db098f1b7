// Micro-benchmark (gfx950): throughput of random partial-line accesses over a 512 MiB footprint, as K1's per-wavefront
// hash tables generate them: 4-byte stores to random 64-byte lines, 4-byte loads, and load+store of the same word.
// Build: hipcc --offload-arch=gfx950 -O3 -o scatter_probe scatter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// each wave owns a 64 KiB slice (like one hash table); every iteration `width` lanes touch random words of it
template <int kMode>
__global__ __launch_bounds__(64) void scatter(uint32_t* __restrict__ buf, uint32_t iters, uint32_t width, uint32_t* __restrict__ out)
{
    uint32_t* t = buf + (size_t)blockIdx.x * 16384;
    uint32_t s = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = (s >> 10) & 16383u;
        if (threadIdx.x < width) {
            if (kMode == 0) t[idx] = s;                       // store only
            if (kMode == 1) acc += t[idx];                    // load only
            if (kMode == 2) { acc += t[idx]; t[idx] = s; }    // exchange
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main()
{
    uint32_t *d_buf, *d_out;
    const uint32_t waves = 8192;
    CHECK(hipMalloc(&d_buf, (size_t)waves * 65536));
    CHECK(hipMalloc(&d_out, waves * 64 * 4));
    CHECK(hipMemset(d_buf, 0, (size_t)waves * 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode)
        for (uint32_t width : {1u, 4u, 16u, 64u}) {
            const uint32_t iters = width >= 16 ? 2000 : 8000;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(scatter<0>, dim3(waves), dim3(64), 0, 0, d_buf, iters, width, d_out);
                if (mode == 1) hipLaunchKernelGGL(scatter<1>, dim3(waves), dim3(64), 0, 0, d_buf, iters, width, d_out);
                if (mode == 2) hipLaunchKernelGGL(scatter<2>, dim3(waves), dim3(64), 0, 0, d_buf, iters, width, d_out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double acc = (double)waves * iters * width;
            printf("%-9s width %2u: %8.2f ms  %7.2f G accesses/s\n", mode == 0 ? "store" : mode == 1 ? "load" : "exchange", width, best,
                   acc / best / 1e6);
            fflush(stdout);
        }
    return 0;
}
