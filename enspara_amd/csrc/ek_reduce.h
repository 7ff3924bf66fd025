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

__device__ __forceinline__ void ek_wave_argmax(float &v, uint32_t &i)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const uint32_t oi = __shfl_xor(i, off, 64);
        if (ek_better(ov, oi, v, i)) {
            v = ov;
            i = oi;
        }
    }
}
