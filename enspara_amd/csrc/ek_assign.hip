// ek_assign.hip -- every frame against K centers: nearest-center assignment.
//
// Replaces assign_to_nearest_center for metric 'rmsd'
// (reference enspara/cluster/util.py:159-205; the center-major loop :199-203:
// one distance pass per center, `dist < distances` strict, so the lowest
// center index wins ties; assignments start at 0, distances at +inf).
// Callers in the reference: predict (util.py:74-77), k-centers warm start
// (kcenters.py:203), the PAM "ambiguous" subset (kmedoids.py:666), batch
// reassignment (util.py:627-629).
//
// One lane owns one frame and walks its atoms in order; CT centers are staged
// in LDS per pass and each frame row loaded from HBM feeds 9*CT FMAs, so the
// frame stream is re-read K/CT times instead of K times.  Accumulation order
// per (frame, center) pair is the same sequential-over-atoms FMA chain as the
// one-center kernel (ek_kcenters.hip): identical bits.
#include "ek_common.h"
#include "ek_qcp.h"

#define CT 8   // centers per LDS tile

__global__ void __launch_bounds__(EK_BLOCK)
ek_assign_kernel(const float *__restrict__ tiles, const double *__restrict__ G,
                 int64_t n, int A, const float *__restrict__ centers,
                 const double *__restrict__ Gc, int K,
                 float *__restrict__ dist, int32_t *__restrict__ assign)
{
    // ctile[a][c][k]: the CT centers' coordinates of atom a are contiguous
    extern __shared__ __attribute__((aligned(16))) float ctile[];
    __shared__ double gtile[CT];
    const int tid = threadIdx.x;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + tid;
    const int64_t tile = f / EK_TILE;
    const float *p = tiles + (size_t)tile * 3 * (size_t)A * EK_TILE + (f % EK_TILE);
    const bool live = f < n;
    const double Gf = live ? G[f] : 0.0;

    float best = __builtin_inff();
    int32_t besti = 0;                       // util.py:186 zeros

    for (int k0 = 0; k0 < K; k0 += CT) {
        const int kc = (K - k0 < CT) ? (K - k0) : CT;
        __syncthreads();
        for (int j = tid; j < 3 * A * CT; j += EK_BLOCK) {
            const int a = j / (3 * CT), rem = j % (3 * CT);
            const int c = rem / 3, k = rem % 3;
            ctile[j] = (c < kc) ? centers[(size_t)(k0 + c) * 3 * A + 3 * a + k]
                                : 0.f;
        }
        if (tid < CT)
            gtile[tid] = (tid < kc) ? Gc[k0 + tid] : 0.0;
        __syncthreads();

        float s[CT][9];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int j = 0; j < 9; ++j)
                s[c][j] = 0.f;

        const float4 *ct4 = (const float4 *)ctile;
#pragma unroll 2
        for (int a = 0; a < A; ++a) {
            const float x = p[(size_t)(3 * a + 0) * EK_TILE];
            const float y = p[(size_t)(3 * a + 1) * EK_TILE];
            const float z = p[(size_t)(3 * a + 2) * EK_TILE];
            float cc[3 * CT];
#pragma unroll
            for (int q = 0; q < 3 * CT / 4; ++q) {
                const float4 v = ct4[a * (3 * CT / 4) + q];
                cc[4 * q + 0] = v.x;
                cc[4 * q + 1] = v.y;
                cc[4 * q + 2] = v.z;
                cc[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float cx = cc[3 * c + 0], cy = cc[3 * c + 1],
                            cz = cc[3 * c + 2];
                s[c][0] = __builtin_fmaf(x, cx, s[c][0]);
                s[c][1] = __builtin_fmaf(x, cy, s[c][1]);
                s[c][2] = __builtin_fmaf(x, cz, s[c][2]);
                s[c][3] = __builtin_fmaf(y, cx, s[c][3]);
                s[c][4] = __builtin_fmaf(y, cy, s[c][4]);
                s[c][5] = __builtin_fmaf(y, cz, s[c][5]);
                s[c][6] = __builtin_fmaf(z, cx, s[c][6]);
                s[c][7] = __builtin_fmaf(z, cy, s[c][7]);
                s[c][8] = __builtin_fmaf(z, cz, s[c][8]);
            }
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (c < kc) {
                    const float d = ek_rmsd_from_S(s[c], Gf, gtile[c], A);
                    if (d < best) {          // strict <: util.py:201
                        best = d;
                        besti = k0 + c;
                    }
                }
            }
        }
    }
    if (live) {
        dist[f] = best;
        assign[f] = besti;
    }
}

void ek_launch_assign(const float *tiles, const double *G, int64_t n, int A,
                      const float *centers_aos, const double *Gc, int32_t K,
                      float *dist, int32_t *assign, hipStream_t s)
{
    if (n <= 0)
        return;
    const int64_t blocks = (n + EK_BLOCK - 1) / EK_BLOCK;
    const size_t lds = (size_t)3 * A * CT * sizeof(float);
    if (lds > 48 * 1024)   // beyond the default dynamic-LDS limit (<= 160 KiB/CU)
        (void)hipFuncSetAttribute((const void *)ek_assign_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL(ek_assign_kernel, dim3((unsigned)blocks), dim3(EK_BLOCK),
                       lds, s, tiles, G, n, A, centers_aos, Gc, K, dist,
                       assign);
}
