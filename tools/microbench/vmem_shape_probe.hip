// Micro-benchmark (gfx950): what one vector-memory instruction of a given SHAPE costs a CU when all 32 wavefront slots issue
// it back to back -- the shapes of K2's per-window accesses.  Each wavefront works inside its own 32 KiB region (hot in L2
// after the first pass), so this measures the address / L1 path, not HBM.  Build: hipcc --offload-arch=gfx950 -O3 -o
// vmem_shape_probe vmem_shape_probe.hip
//   0  dword load, lane * 4 (aligned, contiguous)                 baseline
//   1  8-byte load, lane * 1 (K2's window load: 64 overlapping unaligned 8-byte reads over 71 bytes)
//   2  dword load at a random unaligned offset per lane (far copies' sources)
//   3  dword store, lane * 4 + 1 (contiguous, unaligned: the flush of the LDS stage)
//   4  dword store, lane * 4 (contiguous, aligned)
//   5  16-byte load by lanes 0..4 only (80 bytes: an alternative window load)
//   6  byte store, lane * 1 (round 1's literal store)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int kShape>
__global__ __launch_bounds__(64) void probe(uint8_t* __restrict__ buf, uint32_t iters, uint32_t* __restrict__ out)
{
    uint8_t* r = buf + (size_t)blockIdx.x * 32768;
    const uint32_t lane = threadIdx.x;
    uint32_t s = (blockIdx.x * 64 + lane) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint32_t base = (i * 64u) & 16383u;
        if (kShape == 0) {
            uint32_t v; __builtin_memcpy(&v, r + base + lane * 4, 4); acc += v;
        } else if (kShape == 1) {
            uint64_t v; __builtin_memcpy(&v, r + base + lane, 8); acc += (uint32_t)v + (uint32_t)(v >> 32);
        } else if (kShape == 2) {
            s = s * 1664525u + 1013904223u;
            uint32_t v; __builtin_memcpy(&v, r + ((s >> 10) & 32767u & ~0u) % 32764u, 4); acc += v;
        } else if (kShape == 3) {
            const uint32_t v = acc + i; __builtin_memcpy(r + base + lane * 4 + 1, &v, 4); acc += lane;
        } else if (kShape == 4) {
            const uint32_t v = acc + i; __builtin_memcpy(r + base + lane * 4, &v, 4); acc += lane;
        } else if (kShape == 5) {
            if (lane < 5) { uint4 v; __builtin_memcpy(&v, r + base + lane * 16, 16); acc += v.x + v.y + v.z + v.w; }
        } else {
            r[base + lane] = (uint8_t)(acc + i); acc += lane;
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int kShape>
void run(uint8_t* d_buf, uint32_t* d_out, const char* name, uint32_t waves = 8192)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t iters = 4000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<kShape>, dim3(waves), dim3(64), 0, 0, d_buf, iters, d_out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double per_cu = (double)waves / 256.0 * iters;                 // instructions of this shape per CU
    printf("shape %d %-58s waves %5u  %7.3f ms  %6.1f ns per instruction per CU (~%5.1f cycles at 2.1 GHz)  %6.1f G lanes/s\n", kShape,
           name, waves, best, best * 1e6 / per_cu, best * 1e6 / per_cu * 2.1, (double)waves * iters * (kShape == 5 ? 5 : 64) / best / 1e6);
    fflush(stdout);
}

int main()
{
    uint8_t* d_buf;
    uint32_t* d_out;
    CHECK(hipMalloc(&d_buf, (size_t)8192 * 32768 + 64));
    CHECK(hipMalloc(&d_out, 8192 * 64 * 4));
    CHECK(hipMemset(d_buf, 1, (size_t)8192 * 32768 + 64));
    run<0>(d_buf, d_out, "dword load, lane*4 (aligned, contiguous)");
    run<1>(d_buf, d_out, "8-byte load, lane*1 (K2 window load)");
    run<2>(d_buf, d_out, "dword load, random unaligned offset per lane");
    for (uint32_t w : {4096u, 2048u, 1024u, 512u}) run<2>(d_buf, d_out, "dword load, random unaligned offset per lane", w);
    run<3>(d_buf, d_out, "dword store, lane*4+1 (contiguous, unaligned: flush)");
    run<4>(d_buf, d_out, "dword store, lane*4 (contiguous, aligned)");
    run<5>(d_buf, d_out, "16-byte load by lanes 0..4 only");
    run<6>(d_buf, d_out, "byte store, lane*1");
    return 0;
}
