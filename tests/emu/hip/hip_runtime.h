// Test-only stand-in for <hip/hip_runtime.h>: lets g++ compile the UNMODIFIED kernel source
// (pim-compression_amd/csrc/snappy_kernels.hpp) for a lockstep CPU wave emulator, so the kernel
// logic can be fuzzed against the oracle in the GPU-less container.  Not part of the product and
// never on any GPU path.  Every lane of a workgroup is a ucontext fiber; wave collectives
// (ballot / readlane / readfirstlane / shfl / wave_barrier) rendezvous the 64 lanes of a wave,
// __syncthreads() the whole workgroup.  The emulator aborts if a collective is reached from
// divergent control flow (lanes of one wave waiting at different call sites).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emu::dynamic_lds());

struct uint4 {
    uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

namespace emu {
struct Dim {
    uint32_t x, y, z;
};
const Dim& tidx();
const Dim& bidx();
const Dim& gdim();
const Dim& bdim();
void* dynamic_lds();

enum Op { OP_FIRSTLANE = 1, OP_BALLOT, OP_READLANE, OP_SHFL_UP, OP_BARRIER, OP_PERMUTE };
uint64_t collective(Op op, uint64_t value, uint32_t arg, int site);
void syncthreads(int site);
}  // namespace emu

#define threadIdx (emu::tidx())
#define blockIdx (emu::bidx())
#define gridDim (emu::gdim())
#define blockDim (emu::bdim())

#define __builtin_amdgcn_readfirstlane(v) ((uint32_t)emu::collective(emu::OP_FIRSTLANE, (uint64_t)(uint32_t)(v), 0, __LINE__))
#define __builtin_amdgcn_readlane(v, l) ((int)emu::collective(emu::OP_READLANE, (uint64_t)(uint32_t)(v), (uint32_t)(l), __LINE__))
#define __builtin_amdgcn_wave_barrier() ((void)emu::collective(emu::OP_BARRIER, 0, 0, __LINE__))
#define __ballot(p) ((unsigned long long)emu::collective(emu::OP_BALLOT, (uint64_t)((p) ? 1 : 0), 0, __LINE__))
#define __shfl(v, l) ((int)emu::collective(emu::OP_READLANE, (uint64_t)(uint32_t)(v), (uint32_t)(l), __LINE__))
// ds_bpermute_b32 (pull: lane l reads the value of lane addr[l] / 4) and ds_permute_b32 (push: lane l sends its value to
// lane addr[l] / 4; the highest sender wins, lanes nobody sends to read 0); both take the lane number from address bits 7:2
#define __builtin_amdgcn_ds_bpermute(a, v) ((int)emu::collective(emu::OP_READLANE, (uint64_t)(uint32_t)(v), ((uint32_t)(a) >> 2) & 63u, __LINE__))
#define __builtin_amdgcn_ds_permute(a, v) ((int)emu::collective(emu::OP_PERMUTE, (uint64_t)(uint32_t)(v), ((uint32_t)(a) >> 2) & 63u, __LINE__))
#define __shfl_up(v, d) ((int)emu::collective(emu::OP_SHFL_UP, (uint64_t)(uint32_t)(v), (uint32_t)(d), __LINE__))
#define __syncthreads() emu::syncthreads(__LINE__)
#define SNAPPY_EMU 1
// popcount of the mask bits below this lane (v_mbcnt_lo_u32_b32 / v_mbcnt_hi_u32_b32)
static inline uint32_t emu_mbcnt_lo(uint32_t mask, uint32_t base, uint32_t lane)
{
    const uint32_t below = lane >= 32 ? 0xffffffffu : ((1u << lane) - 1u);
    return base + (uint32_t)__builtin_popcount(mask & below);
}
static inline uint32_t emu_mbcnt_hi(uint32_t mask, uint32_t base, uint32_t lane)
{
    const uint32_t below = lane <= 32 ? 0u : ((1u << (lane - 32)) - 1u);
    return base + (uint32_t)__builtin_popcount(mask & below);
}
#define __builtin_amdgcn_mbcnt_lo(m, b) emu_mbcnt_lo((uint32_t)(m), (uint32_t)(b), emu::tidx().x & 63u)
#define __builtin_amdgcn_mbcnt_hi(m, b) emu_mbcnt_hi((uint32_t)(m), (uint32_t)(b), emu::tidx().x & 63u)
// exec = mask: true in the lanes whose bit is set (no cross-lane traffic)
#define __builtin_amdgcn_inverse_ballot_w64(m) ((((unsigned long long)(m)) >> (emu::tidx().x & 63u)) & 1ull)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __threadfence() ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)

// single-threaded fibers: a plain read-modify-write is atomic
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v)
{
    const uint32_t old = *p;
    *p = old + v;
    return old;
}
static inline uint32_t atomicOr(uint32_t* p, uint32_t v)
{
    const uint32_t old = *p;
    *p = old | v;
    return old;
}
