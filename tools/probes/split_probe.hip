// Probe (measurement only, round 6): producer / consumer waves on ONE SIMD.
// A 512-thread workgroup per CU: waves 0-3 issue nothing but v_mfma_f32_16x16x4_f32
// (36 independent accumulators of 4 registers, the pass's trip), waves 4-7 nothing
// but vector work -- f32 FMA chains, f64 FMA chains, LDS reads -- in amounts like the
// pass's epilogue per wave tile (~3000 vector instructions, ~150 LDS reads).  What
// does the matrix wave's tile cost beside it, and the vector wave's work beside the
// matrix wave?  (Which waves share a SIMD: HW_ID is reported.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int SELF>
__global__ void __launch_bounds__(512, 1)
split(float *out, unsigned long long *cyc, unsigned *ids, int tiles, int nm, int nv32,
      int nv64, int nlds, int who, int prio = 0)
{
    __shared__ float lds[8][64 * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned hwid = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    if (ids && lane == 0)
        ids[blockIdx.x * 8 + wave] = hwid;
    for (int k = 0; k < 16; ++k)
        lds[wave][lane * 16 + k] = (float)(lane + k);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float res = 0.f;
    if (wave < 4) {
        if (who & 1) {
            v4f acc[36];
            for (int q = 0; q < 36; ++q)
                acc[q] = v4f{0.f, 0.f, 0.f, 0.f};
            float a = 1.f + lane, b = 2.f;
            float g[8];
            for (int k = 0; k < 8; ++k) g[k] = 1.f + lane + k;
            for (int t = 0; t < tiles; ++t)
                for (int it = 0; it < nm / 36; ++it) {
#pragma unroll
                    for (int q = 0; q < 36; ++q) {
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
                        // (SELF: the SAME wave issues that many independent f32 FMAs
                        // behind every matrix instruction -- compile time, so that the
                        // loop stays straight-line code)
#pragma unroll
                        for (int u = 0; u < SELF; ++u)
                            g[(q + u) & 7] = __builtin_fmaf(g[(q + u) & 7], 1.0000001f, 0.5f);
                    }
                }
            for (int q = 0; q < 36; ++q)
                res += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
            for (int k = 0; k < 8; ++k) res += g[k];
        }
    } else if (who & 2) {
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        if (prio == 2) __builtin_amdgcn_s_setprio(2);
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        float f[8];
        double d[4];
        for (int k = 0; k < 8; ++k) f[k] = 1.f + lane + k;
        for (int k = 0; k < 4; ++k) d[k] = 1.0 + lane + k;
        for (int t = 0; t < tiles; ++t) {
            for (int it = 0; it < nv32 / 8; ++it) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    f[k] = __builtin_fmaf(f[k], 1.0000001f, 0.5f);
            }
            for (int it = 0; it < nv64 / 4; ++it) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    d[k] = __builtin_fma(d[k], 1.0000001, 0.5);
            }
            for (int it = 0; it < nlds; ++it)
                f[it & 7] += lds[wave][((lane + it) & 63) * 16 + (it & 15)];
        }
        for (int k = 0; k < 8; ++k) res += f[k];
        for (int k = 0; k < 4; ++k) res += (float)d[k];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0)
        cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

int main()
{
    const int wgs = 256, tiles = 8;
    float *d;
    unsigned long long *c, h[wgs * 8];
    unsigned *ids, hid[wgs * 8];
    hipMalloc(&d, wgs * 512 * 4);
    hipMalloc(&c, sizeof(h));
    hipMalloc(&ids, sizeof(hid));
    split<0><<<wgs, 512>>>(d, c, ids, 1, 360, 80, 40, 8, 3);
    hipMemcpy(hid, ids, sizeof(hid), hipMemcpyDeviceToHost);
    printf("workgroup 0, HW_ID of waves 0..7 (SIMD = bits 5:4, slot = bits 3:0):");
    for (int w = 0; w < 8; ++w)
        printf(" %x/%x", (hid[w] >> 4) & 3, hid[w] & 15);
    int paired = 0;
    for (int b = 0; b < wgs; ++b)
        for (int w = 0; w < 4; ++w)
            paired += ((hid[b * 8 + w] >> 4) & 3) == ((hid[b * 8 + w + 4] >> 4) & 3);
    printf("\nwaves w and w + 4 on the same SIMD: %d of %d\n", paired, wgs * 4);
    struct { int nm, nv32, nv64, nlds, who; const char *what; int prio; int self; } cfg[] = {
        {2700, 0, 0, 0, 1, "matrix wave alone"},
        {2700, 2400, 600, 150, 2, "vector wave alone (2400 f32 + 600 f64 + 150 LDS per tile)"},
        {2700, 2400, 600, 150, 3, "both"},
        {2700, 3000, 0, 150, 3, "both, vector work all f32"},
        {2700, 1200, 300, 150, 3, "both, half the vector work"},
        {2700, 4800, 1200, 300, 3, "both, twice the vector work"},
        {2700, 2400, 600, 150, 3, "both, vector wave at s_setprio 1", 1},
        {2700, 2400, 600, 150, 3, "both, vector wave at s_setprio 3", 3},
        {2700, 3000, 0, 0, 3, "both, 3000 f32 only, s_setprio 3", 3},
        {2700, 0, 750, 0, 3, "both, 750 f64 only, s_setprio 3", 3},
        {2700, 0, 0, 300, 3, "both, 300 LDS reads only, s_setprio 3", 3},
        {2700, 4800, 1200, 300, 3, "both, twice the vector work, s_setprio 3", 3},
        {2700, 0, 0, 0, 1, "matrix wave alone, 1 f32 FMA of its own behind every MFMA", 0, 1},
        {2700, 0, 0, 0, 1, "matrix wave alone, 2 f32 FMAs of its own behind every MFMA", 0, 2},
        {2700, 0, 0, 0, 1, "matrix wave alone, 4 f32 FMAs of its own behind every MFMA", 0, 4},
        {2700, 2400, 600, 150, 3, "both, matrix wave with 1 FMA of its own per MFMA", 3, 1},
    };
    for (auto &g : cfg) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
#define GO(S) split<S><<<wgs, 512>>>(d, c, nullptr, tiles, g.nm, g.nv32, g.nv64, g.nlds, g.who, g.prio)
#define GOS() do { if (g.self == 1) GO(1); else if (g.self == 2) GO(2); else if (g.self == 4) GO(4); else GO(0); } while (0)
        GOS();
        hipEventRecord(e0);
        GOS();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < wgs; ++b)
            for (int w = 0; w < 8; ++w)
                (w < 4 ? m : v) += (double)h[b * 8 + w];
        printf("%-62s: %.3f ms; per tile: matrix wave %.0f cycles (%.1f per MFMA), vector wave %.0f cycles\n",
               g.what, ms, m / (wgs * 4) / tiles, m / (wgs * 4) / tiles / g.nm, v / (wgs * 4) / tiles);
    }
    return 0;
}
