// Probe (measurement only): do one wave's vector instructions (f32 / f64) run
// beside another wave's v_mfma_f32_16x16x1_4b_f32 on the same SIMD?
// Each wave: PH phases of [NM matrix instructions][NV vector FMAs]; two waves
// per SIMD (256-thread workgroups, 2 per CU), the second optionally started late.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int F64>
__global__ void __launch_bounds__(256, 2)
phases(float *out, int nm, int nv, int ph, int stagger, float a0, unsigned *ids)
{
    const unsigned hwid = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    if (stagger > 0) {
        if (hwid & 1u)
            for (int q = 0; q < stagger; ++q)
                __builtin_amdgcn_s_sleep(127);
    } else if (stagger < 0) {
        if (blockIdx.x >= 256)
            for (int q = 0; q < -stagger; ++q)
                __builtin_amdgcn_s_sleep(127);
    }
    if (ids && (threadIdx.x & 63) == 0)
        ids[blockIdx.x * 4 + threadIdx.x / 64] = hwid;
    v16f acc[9];
    for (int q = 0; q < 9; ++q)
        for (int r = 0; r < 16; ++r)
            acc[q][r] = 0.f;
    float a = a0 + threadIdx.x, b = 2.f;
    double d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3;
    float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3;
    for (int p = 0; p < ph; ++p) {
        for (int it = 0; it < nm / 9; ++it) {
#pragma unroll
            for (int q = 0; q < 9; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[q], 0, 0, 0);
        }
        for (int it = 0; it < nv / 4; ++it) {
            if (F64) {
                d0 = __builtin_fma(d0, 1.0000001, 0.5); d1 = __builtin_fma(d1, 1.0000001, 0.5);
                d2 = __builtin_fma(d2, 1.0000001, 0.5); d3 = __builtin_fma(d3, 1.0000001, 0.5);
            } else {
                f0 = __builtin_fmaf(f0, 1.0000001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0000001f, 0.5f);
                f2 = __builtin_fmaf(f2, 1.0000001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0000001f, 0.5f);
            }
        }
    }
    float s = (float)(d0 + d1 + d2 + d3) + f0 + f1 + f2 + f3;
    for (int q = 0; q < 9; ++q)
        for (int r = 0; r < 16; ++r)
            s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int F64>
static void run(float *d, int wgs, int nm, int nv, int ph, int stagger)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    phases<F64><<<wgs, 256>>>(d, nm, nv, ph, stagger, 1.f, nullptr);
    hipEventRecord(e0);
    phases<F64><<<wgs, 256>>>(d, nm, nv, ph, stagger, 1.f, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s wgs %4d  mfma %5d  valu %5d  phases %d  stagger %2d : %.3f ms\n",
           F64 ? "f64" : "f32", wgs, nm, nv, ph, stagger, ms);
}

int main()
{
    float *d;
    hipMalloc(&d, 16 << 20);
    {
        unsigned *ids, h[2048];
        hipMalloc(&ids, sizeof(h));
        phases<1><<<512, 256>>>(d, 2700, 6000, 1, 0, 1.f, ids);
        hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost);
        int cnt[16] = {0};
        for (int i = 0; i < 2048; ++i) cnt[h[i] & 15]++;
        printf("wave slot histogram (512 wgs):");
        for (int i = 0; i < 16; ++i) printf(" %d", cnt[i]);
        printf("\nwg 0: %04x %04x %04x %04x  wg 1: %04x  wg 256: %04x %04x %04x %04x wg 257: %04x\n", h[0], h[1], h[2], h[3], h[4], h[1024], h[1025], h[1026], h[1027], h[1028]);
    }
    for (int wgs : {512}) {
        run<1>(d, wgs, 2700, 0, 4, 0);
        run<1>(d, wgs, 0, 6000, 4, 0);
        run<1>(d, wgs, 2700, 6000, 4, 0);
        run<1>(d, wgs, 2700, 6000, 4, 4);
        run<1>(d, wgs, 2700, 6000, 4, 8);
        run<1>(d, wgs, 2700, 6000, 4, -4);
        run<1>(d, wgs, 2700, 6000, 4, -8);
        run<1>(d, wgs, 2700, 6000, 4, -12);
        run<0>(d, wgs, 0, 6000, 4, 0);
        run<0>(d, wgs, 2700, 6000, 4, 0);
        run<0>(d, wgs, 2700, 6000, 4, 4);
    }
    return 0;
}
