// ek_lanes.h -- in-order pairwise sums of doubles over the lanes of a wave,
// without the LDS crossbar.
//
// numpy's pairwise sum over a perfect tree -- (l0 + l1) + (l2 + l3), .. -- is a
// butterfly: floating-point addition commutes, so after level k every lane of
// a group of 2^k holds the group's in-order sum.  __shfl_xor does each level
// with two ds_bpermute_b32 per double (~100 cycles of latency each, one after
// the other); the data-parallel-primitive moves and gfx950's lane swaps below
// cost a few cycles.  After the first two levels all four lanes of a quad hold
// the same value, so a level only needs "some lane of the partner group":
// row_half_mirror (lane i <- 7 - i) and row_mirror (i <- 15 - i) reach it.
// (tools/probes/lane_sum.hip checks the results against __shfl_xor bit for bit.)
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL> __device__ __forceinline__ double ek_dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// groups of eight consecutive lanes
__device__ __forceinline__ double ek_tree_sum8(double r)
{
    r = r + ek_dpp_f64<0xB1>(r);        // quad_perm [1,0,3,2]
    r = r + ek_dpp_f64<0x4E>(r);        // quad_perm [2,3,0,1]
    r = r + ek_dpp_f64<0x141>(r);       // row_half_mirror
    return r;
}

// each half of the wave (lanes 0..31, 32..63)
__device__ __forceinline__ double ek_tree_sum32(double r)
{
    r = ek_tree_sum8(r);
    r = r + ek_dpp_f64<0x140>(r);       // row_mirror
    {   // rows 0|1 and 2|3: v_permlane16_swap exchanges the odd rows of its first
        // operand with the even rows of the second
        const unsigned lo = (unsigned)__double2loint(r), hi = (unsigned)__double2hiint(r);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        r = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return r;
}

// the whole wave
__device__ __forceinline__ double ek_tree_sum64(double r)
{
    r = ek_tree_sum32(r);
    {   // the two halves
        const unsigned lo = (unsigned)__double2loint(r), hi = (unsigned)__double2hiint(r);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        r = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return r;
}

// A workgroup barrier for steps that hand over LDS contents only.  __syncthreads()
// also waits for every global load still in flight (its release fence is
// s_waitcnt vmcnt(0)), which would make each barrier wait for the loads asked for
// ahead of time -- the next slot's frames, the tables -- at memory latency.
__device__ __forceinline__ void ek_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

