// lane_sum.hip -- ek_lanes.h against __shfl_xor, bit for bit (measurement / check only)
//   hipcc --offload-arch=gfx950 -O3 -I enspara_amd/csrc tools/probes/lane_sum.hip -o tools/probes/lane_sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "ek_lanes.h"

__global__ void k(const double *x, double *o8, double *o64, double *r8, double *r64,
                  double *o32, double *r32)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double a = x[i], b = a;
    for (int o = 1; o < 8; o <<= 1)
        a = a + __shfl_xor(a, o, 8);
    for (int o = 1; o < 64; o <<= 1)
        b = b + __shfl_xor(b, o, 64);
    double c = x[i];
    for (int o = 1; o < 32; o <<= 1)
        c = c + __shfl_xor(c, o, 32);
    r32[i] = c;
    o32[i] = ek_tree_sum32(x[i]);
    r8[i] = a;
    r64[i] = b;
    o8[i] = ek_tree_sum8(x[i]);
    o64[i] = ek_tree_sum64(x[i]);
}

int main()
{
    const int n = 1 << 16;
    double *h = (double *)malloc(n * 8), *d[7], *g[6];
    srand(1);
    for (int i = 0; i < n; ++i)
        h[i] = (rand() / (double)RAND_MAX) * exp2((double)(rand() % 40 - 20));
    for (int j = 0; j < 7; ++j)
        hipMalloc((void **)&d[j], n * 8);
    hipMemcpy(d[0], h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
    for (int j = 0; j < 6; ++j) {
        g[j] = (double *)malloc(n * 8);
        hipMemcpy(g[j], d[j + 1], n * 8, hipMemcpyDeviceToHost);
    }
    int bad8 = 0, bad64 = 0, bad32 = 0;
    for (int i = 0; i < n; ++i) {
        bad8 += memcmp(&g[0][i], &g[2][i], 8) != 0;
        bad64 += memcmp(&g[1][i], &g[3][i], 8) != 0;
        bad32 += memcmp(&g[4][i], &g[5][i], 8) != 0;
    }
    printf("lane_sum: %d values, groups of 8: %d differ, half waves: %d, waves: %d differ\n", n,
           bad8, bad32, bad64);
    return bad8 || bad64 || bad32;
}
