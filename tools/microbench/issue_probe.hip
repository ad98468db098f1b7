// Micro-benchmark (gfx950): cost of a wave-uniform scalar/readlane/branch chain like K1's probe step, without memory,
// at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mode 0: chain of dependent s_add only (scalar issue cadence)
// mode 1: readlane -> scalar use -> readlane (VALU->SGPR->VALU round trips)
// mode 2: like 1 plus data-dependent taken branches
// mode 3: like 2 plus one global store per iteration
__global__ __launch_bounds__(64) void chain(uint32_t steps, int mode, uint32_t* __restrict__ sink, uint64_t* __restrict__ cycles,
                                            uint32_t* __restrict__ out)
{
    uint32_t v = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t s = blockIdx.x & 63;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < steps; ++i) {
        if (mode == 0) {
            s = s * 5 + 1; s ^= s >> 3; s = s * 5 + 1; s ^= s >> 3; s = s * 5 + 1; s ^= s >> 3; s = s * 5 + 1; s ^= s >> 3;
        } else {
            // 4 readlanes whose lane select depends on the previous result
            uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(s & 63));
            s = (s + a) & 63;
            uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)(v ^ 0x55), (int)s);
            s = (s ^ b) & 63;
            if (mode >= 2) {
                if (b & 1) { acc += 3; s = (s + 7) & 63; } else { acc ^= b; }
                if (a & 2) { acc += a; } else { s = (s + 1) & 63; acc ^= 0x1234; }
                if ((a ^ b) & 4) { acc = acc * 3 + 1; }
            }
            uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)(v + acc), (int)s);
            s = (s + c) & 63;
            uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)(v * 3), (int)s);
            s = (s ^ d) & 63;
            v += (threadIdx.x == s) ? 1u : 0u;          // one VALU update dependent on the scalar state
            if (mode >= 3) sink[(size_t)blockIdx.x * 16384 + ((d * 16) & 16383)] = acc;   // uniform-address store
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = v + s + acc;
}

int main()
{
    uint32_t *d_sink, *d_out;
    uint64_t* d_cycles;
    CHECK(hipMalloc(&d_sink, (size_t)8192 * 16384 * 4));
    CHECK(hipMalloc(&d_out, 8192 * 64 * 4));
    CHECK(hipMalloc(&d_cycles, 8192 * 8));
    const uint32_t steps = 20000;
    for (int mode = 0; mode < 4; ++mode)
        for (uint32_t waves : {256u, 1024u, 2048u, 4096u, 8192u}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(chain, dim3(waves), dim3(64), 0, 0, steps, mode, d_sink, d_cycles, d_out);
                CHECK(hipDeviceSynchronize());
            }
            std::vector<uint64_t> c(waves);
            CHECK(hipMemcpy(c.data(), d_cycles, waves * 8, hipMemcpyDeviceToHost));
            double cyc = 0;
            for (uint32_t w = 0; w < waves; ++w) cyc += (double)c[w];
            printf("mode %d waves %5u (%4.1f per SIMD): %8.1f cycles/iteration\n", mode, waves, waves / 1024.0, cyc / waves / steps);
            fflush(stdout);
        }
    return 0;
}
