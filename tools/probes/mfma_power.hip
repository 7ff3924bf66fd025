// Probe (measurement only): does the rate of v_mfma_f32_16x16x1_4b_f32 depend on
// the DATA?  The chip is power-limited: the same instruction stream with
// operands that are zero / constant / different for every instruction.
//   mode 0: all operands zero       mode 1: one (a, b) per lane, never changing
//   mode 2: 12 + 12 random values per lane, a new pair for every instruction
//   mode 3: as 2, magnitudes like centred coordinates (|x| ~ 1), random signs
// Prints TFLOP/s and the ratio of shader clocks (s_memtime) to the 100 MHz
// constant clock (s_memrealtime) seen by one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256, 2)
rate(const float *__restrict__ in, float *out, int iters, unsigned long long *clk)
{
    v16f acc[9];
    for (int q = 0; q < 9; ++q)
        for (int r = 0; r < 16; ++r)
            acc[q][r] = 0.f;
    float A[12], B[12];
    const float *p = in + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 24;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        A[k] = p[k];
        B[k] = p[12 + k];
    }
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                acc[0 + j] = __builtin_amdgcn_mfma_f32_16x16x1f32(A[3 * e + 0], B[3 * e + j], acc[0 + j], 0, 0, 0);
                acc[3 + j] = __builtin_amdgcn_mfma_f32_16x16x1f32(A[3 * e + 1], B[3 * e + j], acc[3 + j], 0, 0, 0);
                acc[6 + j] = __builtin_amdgcn_mfma_f32_16x16x1f32(A[3 * e + 2], B[3 * e + j], acc[6 + j], 0, 0, 0);
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0.f;
    for (int q = 0; q < 9; ++q)
        for (int r = 0; r < 16; ++r)
            s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

int main()
{
    const int wgs = 4096, iters = 75 * 8;       // 36 MFMAs per iteration
    const size_t nin = (size_t)wgs * 256 * 24;
    float *din, *dout;
    unsigned long long *dclk, hclk[2];
    hipMalloc(&din, nin * 4);
    hipMalloc(&dout, (size_t)wgs * 256 * 4);
    hipMalloc(&dclk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> h(nin);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            srand(1);
            for (size_t i = 0; i < nin; ++i) {
                const float u = (float)rand() / RAND_MAX;
                if (mode == 0)
                    h[i] = 0.f;
                else if (mode == 1)
                    h[i] = 1.25f;
                else if (mode == 2)
                    h[i] = u;
                else
                    h[i] = (u - 0.5f) * 3.f;
            }
            if (mode == 1)      // one pair per lane: A[k] all equal, B[k] all equal
                for (size_t i = 0; i < nin; i += 24)
                    for (int k = 0; k < 24; ++k)
                        h[i + k] = k < 12 ? 1.0f + (i % 97) * 0.01f : 2.0f;
            hipMemcpy(din, h.data(), nin * 4, hipMemcpyHostToDevice);
            for (int w = 0; w < 3; ++w)     // warm: clocks settle under the load
                rate<<<wgs, 256>>>(din, dout, iters, dclk);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w)
                rate<<<wgs, 256>>>(din, dout, iters, dclk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 10;
            hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost);
            const double mf = (double)wgs * 4 * iters * 36;
            printf("mode %d: %.3f ms per launch, %.1f TFLOP/s; one wave: %llu s_memtime ticks / %llu x 10 ns "
                   "= %.3f per ns\n", mode, ms, mf * 2048 / ms / 1e9, hclk[0], hclk[1],
                   (double)hclk[0] / (hclk[1] * 10.0));
        }
    return 0;
}
