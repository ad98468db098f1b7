// Micro-benchmark (gfx950): dependent-load latency as seen by ONE wavefront's serial chain, idle and under the load of
// many other chains, with and without an interleaved store.  K1's parse is such a chain; this prices its round trips.
// Build: hipcc --offload-arch=gfx950 -O3 -o latency_probe latency_probe.hip     Run: ./latency_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// chain[i] = index of the next 64-byte line; every wave walks its own cycle segment
template <int kStore>
__global__ __launch_bounds__(64) void chase(const uint32_t* __restrict__ chain, uint32_t lines, uint32_t steps,
                                            uint32_t* __restrict__ sink, uint32_t sink_lines, uint64_t* __restrict__ cycles,
                                            uint32_t* __restrict__ out)
{
    uint32_t cur = (uint32_t)(((uint64_t)blockIdx.x * 2654435761u) % lines);
    uint32_t s = blockIdx.x * 7919u;
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t r0 = wall_clock64();
    for (uint32_t i = 0; i < steps; ++i) {
        const uint32_t nxt = chain[(size_t)cur * 16];                 // same address in all lanes
        if (kStore == 1) {                                            // independent store to a random line, all lanes same address
            s = s * 1664525u + 1013904223u;
            sink[(size_t)(s % sink_lines) * 16] = i;
        }
        if (kStore == 2) {                                            // store to the line just read (table-like)
            const_cast<uint32_t*>(chain)[(size_t)cur * 16 + 1] = i;
        }
        cur = __builtin_amdgcn_readfirstlane(nxt);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t r1 = wall_clock64();
    if (threadIdx.x == 0) {
        cycles[blockIdx.x * 2] = t1 - t0;
        cycles[blockIdx.x * 2 + 1] = r1 - r0;
        out[blockIdx.x] = cur;
    }
}

// 8-lane gather chase: lanes 0..7 each follow their own chain, one wait per step for all of them
__global__ __launch_bounds__(64) void chase_gather(const uint32_t* __restrict__ chain, uint32_t lines, uint32_t steps,
                                                   uint32_t width, uint64_t* __restrict__ cycles, uint32_t* __restrict__ out)
{
    uint32_t cur = (uint32_t)(((uint64_t)(blockIdx.x * 64 + threadIdx.x) * 2654435761u) % lines);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < steps; ++i) {
        uint32_t nxt = cur;
        if (threadIdx.x < width) nxt = chain[(size_t)cur * 16];
        // wave-uniform dependency on all lanes' results, as in K1's gather
        const unsigned long long b = __ballot(nxt == 0xffffffffu);
        cur = nxt + (uint32_t)b;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = cur;
}

int main()
{
    const size_t max_lines = (size_t)(1536ull << 20) / 64;
    std::vector<uint32_t> h(max_lines * 16, 0);
    uint32_t *d_chain, *d_sink, *d_out;
    uint64_t* d_cycles;
    const uint32_t sink_lines = (256u << 20) / 64;
    CHECK(hipMalloc(&d_chain, max_lines * 64));
    CHECK(hipMalloc(&d_sink, (size_t)sink_lines * 64));
    CHECK(hipMalloc(&d_out, 8192 * 64 * 4));
    CHECK(hipMalloc(&d_cycles, 8192 * 16));
    std::mt19937 rng(1);
    const size_t foot_mib[] = {1, 16, 128, 1536};   // per-XCD L2 (4 MiB) / all L2s / Infinity Cache (256 MiB) / HBM
    for (size_t fm : foot_mib) {
        const uint32_t lines = (uint32_t)((fm << 20) / 64);
        std::vector<uint32_t> perm(lines);
        for (uint32_t i = 0; i < lines; ++i) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        for (uint32_t i = 0; i < lines; ++i) h[(size_t)perm[i] * 16] = perm[(i + 1) % lines];   // one big cycle
        CHECK(hipMemcpy(d_chain, h.data(), (size_t)lines * 64, hipMemcpyHostToDevice));
        for (uint32_t waves : {1u, 1024u, 8192u}) {
            const uint32_t steps = waves == 1 ? 20000 : 4000;
            for (int mode = 0; mode < 3; ++mode) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(waves), dim3(64), 0, 0, d_chain, lines, steps, d_sink, sink_lines, d_cycles, d_out);
                    if (mode == 1) hipLaunchKernelGGL(chase<1>, dim3(waves), dim3(64), 0, 0, d_chain, lines, steps, d_sink, sink_lines, d_cycles, d_out);
                    if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(waves), dim3(64), 0, 0, d_chain, lines, steps, d_sink, sink_lines, d_cycles, d_out);
                    CHECK(hipDeviceSynchronize());
                }
                std::vector<uint64_t> c(waves * 2);
                CHECK(hipMemcpy(c.data(), d_cycles, waves * 16, hipMemcpyDeviceToHost));
                double cyc = 0, rt = 0;
                for (uint32_t w = 0; w < waves; ++w) { cyc += (double)c[w * 2]; rt += (double)c[w * 2 + 1]; }
                cyc /= waves; rt /= waves;
                printf("footprint %5zu MiB waves %5u mode %s: %8.1f cycles/step  %7.1f ns/step  (shader clock %.0f MHz)\n", fm, waves,
                       mode == 0 ? "load      " : mode == 1 ? "load+store" : "load+wr-same", cyc / steps, rt * 10.0 / steps, cyc / (rt * 10.0) * 1000.0);
                fflush(stdout);
            }
            for (uint32_t width : {1u, 8u, 16u, 64u}) {
                for (int rep = 0; rep < 2; ++rep) {
                    hipLaunchKernelGGL(chase_gather, dim3(waves), dim3(64), 0, 0, d_chain, lines, steps, width, d_cycles, d_out);
                    CHECK(hipDeviceSynchronize());
                }
                std::vector<uint64_t> c(waves * 2);
                CHECK(hipMemcpy(c.data(), d_cycles, waves * 16, hipMemcpyDeviceToHost));
                double cyc = 0;
                for (uint32_t w = 0; w < waves; ++w) cyc += (double)c[w * 2];
                printf("footprint %5zu MiB waves %5u gather width %2u: %8.1f cycles/step\n", fm, waves, width, cyc / waves / steps);
                fflush(stdout);
            }
        }
    }
    return 0;
}
