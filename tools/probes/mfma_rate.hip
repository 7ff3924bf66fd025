// Probe (measurement only): issue rate of v_mfma_f32_16x16x1_4b_f32 with nine
// independent accumulators, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256, 2) rate(float *out, int iters, float a0, float b0)
{
    v16f acc[NACC];
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r)
            acc[q][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r)
            s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *d;
    hipMalloc(&d, 4 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024, 4096}) {
        const int iters = 300 * 4;      // x 9 MFMAs = 10800 per wave
        rate<9><<<wgs, 256>>>(d, iters, 1.f, 2.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        rate<9><<<wgs, 256>>>(d, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mf = (double)wgs * 4 * iters * 9;
        printf("9 acc, %5d workgroups x 4 waves: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD-slot\n",
               wgs, ms, mf * 2048 / ms / 1e9, ms * 1e6 / (mf / 1024));
        rate<3><<<wgs, 256>>>(d, iters * 3, 1.f, 2.f);
        hipEventRecord(e0);
        rate<3><<<wgs, 256>>>(d, iters * 3, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("3 acc, %5d workgroups x 4 waves: %.3f ms, %.1f TFLOP/s\n", wgs, ms, mf * 2048 / ms / 1e9);
    }
    return 0;
}
