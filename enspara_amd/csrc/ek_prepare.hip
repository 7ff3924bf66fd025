// ek_prepare.hip -- centring, traces and the AoS -> frame-minor transposition.
//
// Replaces mdtraj's per-call centring inside rmsd(precentered=False) and the
// one-off md.Trajectory.center_coordinates() of enspara/cluster/util.py:624-629.
// Input is the md.Trajectory.xyz layout, float32 [frame][atom][3]
// (enspara/util/load.py:211-216).
//
// One workgroup of 256 threads owns one tile of 256 frames; thread t owns
// frame t.  The AoS rows are staged through LDS in chunks of CH atoms so that
// global reads are contiguous 12*CH-byte runs and every thread then walks its
// own frame sequentially in atom order (the summation order is part of the
// numerical contract, ek_qcp.h).  Two passes over the input: means, then
// centred coordinates + traces + transposed store (1 KiB per row, coalesced).
#include "ek_common.h"

#define CH 16                    // atoms per staged chunk
#define ROWF (3 * CH + 1)        // LDS row stride in floats (odd: no conflicts)

template <bool TILED>
__global__ void __launch_bounds__(EK_BLOCK)
ek_prepare_kernel(const float *__restrict__ src, int64_t count, int A,
                  float *__restrict__ out, double *__restrict__ G,
                  int64_t first_frame, int64_t n_total,
                  float *__restrict__ aos_copy)
{
    __shared__ float stage[EK_BLOCK * ROWF];
    const int t = threadIdx.x;
    const int64_t f0 = (int64_t)blockIdx.x * EK_BLOCK;   // relative to src
    const int64_t f = f0 + t;
    const bool live = f < count;
    const int64_t rows_here = (count - f0 < EK_BLOCK) ? (count - f0) : EK_BLOCK;

    // ---- pass 1: per-frame sums in float64, sequential in atom order -------
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int a0 = 0; a0 < A; a0 += CH) {
        const int w = 3 * ((A - a0 < CH) ? (A - a0) : CH);
        const int64_t total = rows_here * w;
        for (int64_t i = t; i < total; i += EK_BLOCK) {
            const int r = (int)(i / w), j = (int)(i % w);
            stage[r * ROWF + j] = src[(f0 + r) * 3 * (int64_t)A + 3 * a0 + j];
        }
        __syncthreads();
        if (live) {
            const float *row = stage + t * ROWF;
            for (int j = 0; j < w; j += 3) {
                sx = sx + (double)row[j + 0];
                sy = sy + (double)row[j + 1];
                sz = sz + (double)row[j + 2];
            }
        }
        __syncthreads();
    }
    const double mx = sx / (double)A, my = sy / (double)A, mz = sz / (double)A;

    // ---- pass 2: centre, trace, store --------------------------------------
    float gx = 0.f, gy = 0.f, gz = 0.f;
    const int64_t gf = first_frame + f;              // frame index in the shard
    float *obase;
    if (TILED)
        obase = out + (size_t)(gf / EK_TILE) * 3 * (size_t)A * EK_TILE +
                (gf % EK_TILE);
    else
        obase = out + (size_t)gf * 3 * (size_t)A;
    for (int a0 = 0; a0 < A; a0 += CH) {
        const int w = 3 * ((A - a0 < CH) ? (A - a0) : CH);
        const int64_t total = rows_here * w;
        for (int64_t i = t; i < total; i += EK_BLOCK) {
            const int r = (int)(i / w), j = (int)(i % w);
            stage[r * ROWF + j] = src[(f0 + r) * 3 * (int64_t)A + 3 * a0 + j];
        }
        __syncthreads();
        if (live) {
            const float *row = stage + t * ROWF;
            for (int j = 0; j < w; j += 3) {
                const float cx = (float)((double)row[j + 0] - mx);
                const float cy = (float)((double)row[j + 1] - my);
                const float cz = (float)((double)row[j + 2] - mz);
                gx = __builtin_fmaf(cx, cx, gx);
                gy = __builtin_fmaf(cy, cy, gy);
                gz = __builtin_fmaf(cz, cz, gz);
                const int r = 3 * a0 + j;
                if (TILED) {
                    obase[(size_t)(r + 0) * EK_TILE] = cx;
                    obase[(size_t)(r + 1) * EK_TILE] = cy;
                    obase[(size_t)(r + 2) * EK_TILE] = cz;
                    if (aos_copy) {         // back into the stage, centred
                        stage[t * ROWF + j + 0] = cx;
                        stage[t * ROWF + j + 1] = cy;
                        stage[t * ROWF + j + 2] = cz;
                    }
                } else {
                    obase[r + 0] = cx;
                    obase[r + 1] = cy;
                    obase[r + 2] = cz;
                }
            }
        }
        __syncthreads();
        if (TILED && aos_copy) {
            // the same centred coordinates frame-major (one frame = 12 A
            // contiguous bytes): what picks a single frame out of the shard --
            // candidate records, the pairwise distances of the candidate pick --
            // reads instead of 3 A cache lines of the tiles
            for (int64_t i = t; i < total; i += EK_BLOCK) {
                const int r = (int)(i / w), j = (int)(i % w);
                aos_copy[(size_t)(first_frame + f0 + r) * 3 * (size_t)A + 3 * a0 +
                         j] = stage[r * ROWF + j];
            }
            __syncthreads();
        }
    }
    if (live)
        G[gf] = ((double)gx + (double)gy) + (double)gz;
}

void ek_launch_prepare_tiles(const float *src_aos, int64_t count, int A,
                             float *tiles, double *G, int64_t first_frame,
                             int64_t n_total, float *aos_copy, hipStream_t s)
{
    if (count <= 0)
        return;
    // padding frames of the last tile are zeroed once at context creation
    const int64_t blocks = (count + EK_BLOCK - 1) / EK_BLOCK;
    hipLaunchKernelGGL(ek_prepare_kernel<true>, dim3((unsigned)blocks),
                       dim3(EK_BLOCK), 0, s, src_aos, count, A, tiles, G,
                       first_frame, n_total, aos_copy);
}

void ek_launch_prepare_centers(const float *src_aos, int32_t count, int A,
                               float *out_aos, double *Gc, hipStream_t s)
{
    if (count <= 0)
        return;
    const int blocks = (count + EK_BLOCK - 1) / EK_BLOCK;
    hipLaunchKernelGGL(ek_prepare_kernel<false>, dim3(blocks), dim3(EK_BLOCK),
                       0, s, src_aos, (int64_t)count, A, out_aos, Gc,
                       (int64_t)0, (int64_t)count, (float *)nullptr);
}
