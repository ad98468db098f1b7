// Does ds_wrxchg_rtn_b32 serialise the lanes of ONE instruction that share an address?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint32_t* out, uint32_t groups)
{
    __shared__ uint32_t cell[64];
    const uint32_t lane = threadIdx.x;
    cell[lane] = 1000 + lane;
    __builtin_amdgcn_wave_barrier();
    // lanes share a cell in groups of `groups` lanes: lane -> cell lane / groups
    const uint32_t c = lane / groups;
    const uint32_t old = __hip_atomic_exchange((__attribute__((address_space(3))) uint32_t*)&cell[c], lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __builtin_amdgcn_wave_barrier();
    out[blockIdx.x * 128 + lane] = old;
    out[blockIdx.x * 128 + 64 + lane] = cell[lane];
}
int main()
{
    uint32_t* d; hipMalloc(&d, 128 * 4 * 8);
    for (uint32_t groups : {1u, 2u, 4u, 64u}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, groups);
        uint32_t h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        // check: within each group, the returned values must be {initial of the cell} + all but one of the group's (lane+1) values, each once; the cell holds the one not returned
        int bad = 0;
        for (uint32_t c = 0; c < 64 / groups; ++c) {
            uint32_t seen_init = 0; uint64_t seen = 0;
            for (uint32_t l = c * groups; l < (c + 1) * groups; ++l) {
                const uint32_t o = h[l];
                if (o == 1000 + c) seen_init++;
                else if (o >= c * groups + 1 && o <= (c + 1) * groups) { if (seen >> (o - 1) & 1) bad++; seen |= 1ull << (o - 1); }
                else bad++;
            }
            const uint32_t fin = h[64 + c];
            if (seen_init != 1) bad++;
            if (!(fin >= c * groups + 1 && fin <= (c + 1) * groups) || (seen >> (fin - 1) & 1)) bad++;
        }
        printf("groups of %2u lanes per cell: %s (bad %d); lane order of the returns in cell 0:", groups, bad ? "NOT a chain" : "chain", bad);
        for (uint32_t l = 0; l < (groups < 8 ? groups : 8); ++l) printf(" %u", h[l]);
        printf("\n");
    }
    return 0;
}
