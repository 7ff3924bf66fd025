// Probe (measurement only): register layout of v_mfma_f32_16x16x4_f32 and whether
// a chain of them accumulates exactly like fmaf in k order (the four k of an
// instruction in ascending order: lane group 0 first).
//   hipcc --offload-arch=gfx950 -O2 -o mfma16x4_probe mfma16x4_probe.hip && ./mfma16x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));

// A[k][i]: lane 16 k + i; B[k][j]: lane 16 k + j; steps of 4 k
__global__ void chain(const float *A, const float *B, int steps, float *out)
{
    const int l = threadIdx.x;
    v4f c = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s)
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s * 64 + l], B[s * 64 + l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r)
        out[l * 4 + r] = c[r];
}

int main()
{
    const int steps = 75, K = 4 * steps;
    float *hA = (float *)malloc(K * 16 * 4), *hB = (float *)malloc(K * 16 * 4), h[256];
    srand(7);
    for (int i = 0; i < K * 16; ++i) {
        hA[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
        hB[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
    }
    float *dA, *dB, *d;
    hipMalloc(&dA, K * 16 * 4); hipMalloc(&dB, K * 16 * 4); hipMalloc(&d, sizeof(h));
    hipMemcpy(dA, hA, K * 16 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, K * 16 * 4, hipMemcpyHostToDevice);
    chain<<<1, 64>>>(dA, dB, steps, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // hypothesis: D[lane l][reg r] = row i = 4 (l / 16) + r, column j = l % 16,
    // sum over k ascending of fmaf(A[k][i], B[k][j], acc); memory index of
    // (k, i): (k / 4) * 64 + (k % 4) * 16 + i
    int diff = 0, diff_rev = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * (l / 16) + r, j = l % 16;
            float acc = 0.f, rev = 0.f;
            for (int k = 0; k < K; ++k)
                acc = fmaf(hA[(k / 4) * 64 + (k % 4) * 16 + i], hB[(k / 4) * 64 + (k % 4) * 16 + j], acc);
            for (int s = 0; s < steps; ++s)
                for (int kk = 3; kk >= 0; --kk)
                    rev = fmaf(hA[s * 64 + kk * 16 + i], hB[s * 64 + kk * 16 + j], rev);
            diff += acc != h[l * 4 + r];
            diff_rev += rev != h[l * 4 + r];
        }
    printf("16x16x4 chain of %d atoms vs fmaf in ascending k: %d of 256 differ "
           "(descending k inside an instruction: %d differ)\n", K, diff, diff_rev);
    return diff != 0;
}
