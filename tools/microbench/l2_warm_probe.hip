// Micro-benchmark (gfx950): does a line read by OTHER waves of the same workgroup serve a later dependent load of wave 0 from L2?
// Phase 1: all waves of a 1024-thread workgroup read `warm` bytes (one dword per 64-byte line).  Phase 2: wave 0 hops through the
// same region with a 16 KiB stride using dependent loads (scalar or vector path) and reports cycles per hop.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_warm_probe l2_warm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int kVector>
__global__ __launch_bounds__(1024) void probe(const uint32_t* __restrict__ buf, uint64_t base_words, uint32_t warm_bytes,
                                              uint32_t hops, uint64_t* __restrict__ out, uint32_t* __restrict__ sink_out)
{
    const uint32_t* __restrict__ p = buf + base_words;
    uint32_t sink = 0;
    for (uint64_t off = (uint64_t)threadIdx.x * 64; off < warm_bytes; off += 1024 * 64) sink ^= p[off / 4];
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t at = 0;                                 // word index; buf[at] holds 0, so the chain is at -> at + 4096 words
        const uint64_t t0 = __builtin_readcyclecounter();
        for (uint32_t i = 0; i < hops; ++i) {
            uint32_t v;
            if (kVector) v = __builtin_amdgcn_readfirstlane(p[at + (threadIdx.x & 0)]);
            else v = p[at];
            at += 4096 + __builtin_amdgcn_readfirstlane(v);
        }
        const uint64_t t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = at; }
    }
    if (sink == 0x1234567 && threadIdx.x == 1023) sink_out[0] = sink;
}

int main()
{
    const size_t bytes = 1024ull << 20;
    uint32_t *d_buf, *d_sink;
    uint64_t* d_out;
    CHECK(hipMalloc(&d_buf, bytes));
    CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMemset(d_buf, 0, bytes));
    const uint32_t hops = 160;                           // 160 x 16 KiB = 2.5 MiB
    int region = 0;
    for (int vec = 0; vec < 2; ++vec)
        for (uint32_t warm : {0u, 2560u << 10}) {
            for (int rep = 0; rep < 3; ++rep) {
                const uint64_t base_words = (uint64_t)(region++) * (8u << 20) / 4;   // a fresh 8 MiB region every time: nothing cached
                if (vec) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(1024), 0, 0, d_buf, base_words, warm, hops, d_out, d_sink);
                else hipLaunchKernelGGL(probe<0>, dim3(1), dim3(1024), 0, 0, d_buf, base_words, warm, hops, d_out, d_sink);
                CHECK(hipDeviceSynchronize());
                uint64_t h[2];
                CHECK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
                printf("%s walker, warm %4u KiB: %7.1f cycles/hop\n", vec ? "vector" : "scalar", warm >> 10, (double)h[0] / hops);
            }
        }
    return 0;
}
