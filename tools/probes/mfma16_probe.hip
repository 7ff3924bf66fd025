// Probe (measurement only): register layout of v_mfma_f32_16x16x1_f32 (4 blocks)
// and whether a chain of them accumulates exactly like fmaf in k order.
//   hipcc --offload-arch=gfx950 -O2 -o mfma16_probe mfma16_probe.hip && ./mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void layout(float *out)
{
    const int l = threadIdx.x;
    v16f c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // A[block b][row i] = 100 * (16 b + i) ; B[block b][col j] = 1 + (16 b + j) / 1024.
    const float a = (float)(l + 1);
    const float b = (float)(l + 1) * 1000.f;
    c = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r)
        out[l * 16 + r] = c[r];
}

// K-long chain: A[k][lane], B[k][lane]
__global__ void chain(const float *A, const float *B, int K, float *out)
{
    const int l = threadIdx.x;
    v16f c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < K; ++k)
        c = __builtin_amdgcn_mfma_f32_16x16x1f32(A[k * 64 + l], B[k * 64 + l], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r)
        out[l * 16 + r] = c[r];
}

int main()
{
    float *d, h[1024];
    hipMalloc(&d, sizeof(h));
    layout<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // decode: value = a * b = (la + 1) * (lb + 1) * 1000 -> find (la, lb)
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const double v = h[l * 16 + r] / 1000.0;
            int fa = -1, fb = -1;
            for (int la = 0; la < 64 && fa < 0; ++la)
                for (int lb = 0; lb < 64; ++lb)
                    if (fabs((la + 1.0) * (lb + 1.0) - v) < 1e-6 && la / 16 == lb / 16) {
                        fa = la; fb = lb; break;
                    }
            // hypothesis: block = r / 4, row i = 4 * (l / 16) + r % 4, col j = l % 16
            const int b = r / 4, i = 4 * (l / 16) + r % 4, j = l % 16;
            if (fa != 16 * b + i || fb != 16 * b + j) {
                if (bad < 10)
                    printf("lane %d reg %d: A lane %d B lane %d (expected %d %d)\n", l, r,
                           fa, fb, 16 * b + i, 16 * b + j);
                ++bad;
            }
        }
    printf("layout: %s (D[lane l][reg r] = A[16 (r/4) + 4 (l/16) + r%%4] * B[16 (r/4) + l%%16])\n",
           bad ? "DIFFERENT" : "as expected");
    // chain exactness
    const int K = 301;
    float *hA = (float *)malloc(K * 64 * 4), *hB = (float *)malloc(K * 64 * 4);
    srand(5);
    for (int i = 0; i < K * 64; ++i) {
        hA[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
        hB[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
    }
    float *dA, *dB;
    hipMalloc(&dA, K * 64 * 4); hipMalloc(&dB, K * 64 * 4);
    hipMemcpy(dA, hA, K * 64 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, K * 64 * 4, hipMemcpyHostToDevice);
    chain<<<1, 64>>>(dA, dB, K, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int diff = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int b = r / 4, i = 4 * (l / 16) + r % 4, j = l % 16;
            float acc = 0.f;
            for (int k = 0; k < K; ++k)
                acc = fmaf(hA[k * 64 + 16 * b + i], hB[k * 64 + 16 * b + j], acc);
            if (memcmp(&acc, &h[l * 16 + r], 4))
                ++diff;
        }
    printf("chain of %d: %d of 1024 results differ from the fmaf chain in k order\n", K, diff);
    return 0;
}
