// ek_reduce.h -- arg-max helpers shared by the k-centers kernels:
// larger value wins, lower index wins ties (np.argmax semantics,
// reference enspara/cluster/kcenters.py:282).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ bool ek_better(float v, uint32_t i, float bv,
                                          uint32_t bi)
{
    return (v > bv) || (v == bv && i < bi);
}

// Wave-wide arg-max, the winner in every lane.  The order (value descending,
// index ascending) is total, so any reduction tree gives the same pair: four
// DPP steps make every row of 16 lanes agree (lane ^ 1, lane ^ 2 inside the
// quads, then the two mirrors: once the quads agree, mirroring pairs them up
// as well as an xor would), and the four row results are combined as scalars.
// Call it with the whole wave active.
// (A butterfly of ds_bpermute shuffles is six dependent trips through the LDS
// crossbar per operand: 0.6 us a call, and the round's small kernels make
// eight or more in a row.)
template <int CTRL>
__device__ __forceinline__ void ek_argmax_dpp_step(float &v, uint32_t &i)
{
    // (a lane whose partner is switched off sees its own pair: neutral)
    const int vb = __builtin_bit_cast(int, v);
    const float ov = __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(vb, vb, CTRL, 0xf, 0xf, false));
    const uint32_t oi =
        (uint32_t)__builtin_amdgcn_update_dpp((int)i, (int)i, CTRL, 0xf, 0xf, false);
    if (ek_better(ov, oi, v, i)) {
        v = ov;
        i = oi;
    }
}

// the arg-max of every ROW of 16 lanes, in all lanes of the row (the first half of
// ek_wave_argmax: what a finer-grained maximum costs is nothing)
__device__ __forceinline__ void ek_row_argmax(float &v, uint32_t &i)
{
    ek_argmax_dpp_step<0xB1>(v, i);     // quad_perm [1,0,3,2]
    ek_argmax_dpp_step<0x4E>(v, i);     // quad_perm [2,3,0,1]
    ek_argmax_dpp_step<0x141>(v, i);    // row_half_mirror
    ek_argmax_dpp_step<0x140>(v, i);    // row_mirror
}

// the rows' results -> the wave's, in every lane
__device__ __forceinline__ void ek_rows_to_wave_argmax(float &v, uint32_t &i)
{
    float bv = __builtin_bit_cast(
        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    uint32_t bi = (uint32_t)__builtin_amdgcn_readlane((int)i, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        const float rv = __builtin_bit_cast(
            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * row));
        const uint32_t ri = (uint32_t)__builtin_amdgcn_readlane((int)i, 16 * row);
        if (ek_better(rv, ri, bv, bi)) {
            bv = rv;
            bi = ri;
        }
    }
    v = bv;
    i = bi;
}

__device__ __forceinline__ void ek_wave_argmax(float &v, uint32_t &i)
{
    ek_argmax_dpp_step<0xB1>(v, i);     // quad_perm [1,0,3,2]
    ek_argmax_dpp_step<0x4E>(v, i);     // quad_perm [2,3,0,1]
    ek_argmax_dpp_step<0x141>(v, i);    // row_half_mirror
    ek_argmax_dpp_step<0x140>(v, i);    // row_mirror
    float bv = __builtin_bit_cast(
        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    uint32_t bi = (uint32_t)__builtin_amdgcn_readlane((int)i, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        const float rv = __builtin_bit_cast(
            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * row));
        const uint32_t ri = (uint32_t)__builtin_amdgcn_readlane((int)i, 16 * row);
        if (ek_better(rv, ri, bv, bi)) {
            bv = rv;
            bi = ri;
        }
    }
    v = bv;
    i = bi;
}

// ---- handing small results from all workgroups of a launch to its last one ------
// (ek_round.hip).  An agent-scope fence costs a write-back / invalidate of the
// whole L2 on gfx950 (eight XCDs, one L2 each) -- per workgroup that doubled the
// pass kernel's time -- so nothing here fences: the few values that cross
// workgroups are written and read with agent-scope relaxed atomics (write-through
// / L2-coherent accesses, `sc1`), every thread waits for its own stores to be
// acknowledged before the workgroup takes its arrival ticket, and the workgroup
// that draws the last ticket therefore finds all of them in place.
//
// This is NOT the HIP/LLVM memory model's release/acquire pairing (there the
// relaxed ticket orders nothing): it leans on three properties of the gfx9
// memory pipeline -- vmcnt counts stores as well as loads (no separate store
// counter as on gfx10+), an `sc1` store is written through to the point of
// agent coherence before it is acknowledged, and a wave issues no speculative
// loads ahead of the ticket it branches on -- and on every value that crosses
// workgroups inside a launch going through the ek_coh_* helpers below.  Hence
// the guard: any other target must get real fences here.  The regression check
// is the fused-vs-unfused comparison of tests/test_gpu_kcenters.py and
// tools/stress_rounds.py.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "ek_arrive_last: fence-free hand-over is written for gfx942/gfx950 (see above)"
#endif
__device__ __forceinline__ void ek_coh_store(float *p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ek_coh_store(int32_t *p, int32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ek_coh_load(const float *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int32_t ek_coh_load(const int32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (value, index) pairs travel as one 64-bit word
template <bool COH, typename BM>
__device__ __forceinline__ BM ek_ld_bm(const BM *p)
{
    static_assert(sizeof(BM) == 8, "EkBlockMax is 8 bytes");
    if (COH) {
        const unsigned long long w = __hip_atomic_load(
            (const unsigned long long *)p, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_bit_cast(BM, w);
    }
    return *p;
}
template <typename BM>
__device__ __forceinline__ void ek_coh_store_bm(BM *p, float v, uint32_t i)
{
    static_assert(sizeof(BM) == 8, "EkBlockMax is 8 bytes");
    BM m;
    m.val = v;
    m.idx = i;
    __hip_atomic_store((unsigned long long *)p,
                       __builtin_bit_cast(unsigned long long, m),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// arrival of this workgroup at `tick`; true in all threads of the last one
// (`expected`: the workgroups that arrive, if not the whole grid)
__device__ __forceinline__ bool ek_arrive_last(unsigned int *tick,
                                               unsigned int expected = 0)
{
    __shared__ bool ek_last_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's stores
    __syncthreads();                                    // ... and everybody's
    if (threadIdx.x == 0)
        ek_last_flag = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT) ==
                       (expected ? expected : gridDim.x) - 1;
    __syncthreads();
    return ek_last_flag;
}

// The same for launches whose workgroups all finish within a few microseconds:
// a thousand returning atomics on ONE address serialise (measured: +15 us on a
// 7 us kernel), so the tickets are drawn on EK_ARRIVE_G addresses -- workgroup b
// on leaves[b % G] -- and only the workgroup that completes its leaf goes on to
// the top counter.  The last workgroup resets both levels.
#define EK_ARRIVE_G 32
__device__ __forceinline__ bool ek_arrive_last_tree(unsigned int *top,
                                                    unsigned int *leaves)
{
    __shared__ bool ek_last_flag2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int G = EK_ARRIVE_G;
        const unsigned int g = blockIdx.x % G;
        const unsigned int size_g = gridDim.x / G + (g < gridDim.x % G ? 1u : 0u);
        const unsigned int n_groups = gridDim.x < G ? gridDim.x : G;
        bool last = false;
        if (__hip_atomic_fetch_add(&leaves[g], 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) == size_g - 1)
            last = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) == n_groups - 1;
        if (last) {
            for (unsigned int i = 0; i < G; ++i)
                leaves[i] = 0;
            *top = 0;
        }
        ek_last_flag2 = last;
    }
    __syncthreads();
    return ek_last_flag2;
}
