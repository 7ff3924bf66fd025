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
    // ctile[a][pair][k][2]: the CT centers' coordinates of atom a are contiguous,
    // centers in pairs so that one packed FMA (v_pk_fma_f32) serves two
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
            ctile[a * (3 * CT) + (c / 2) * 6 + k * 2 + (c & 1)] =
                (c < kc) ? centers[(size_t)(k0 + c) * 3 * A + 3 * a + k] : 0.f;
        }
        if (tid < CT)
            gtile[tid] = (tid < kc) ? Gc[k0 + tid] : 0.0;
        __syncthreads();

        ek_v2f s2[CT / 2][9];   // (center 2p, center 2p+1): independent FMA chains
#pragma unroll
        for (int c = 0; c < CT / 2; ++c)
#pragma unroll
            for (int j = 0; j < 9; ++j)
                s2[c][j] = (ek_v2f){0.f, 0.f};

        const float4 *ct4 = (const float4 *)ctile;
#pragma unroll 2
        for (int a = 0; a < A; ++a) {
            const float xs = p[(size_t)(3 * a + 0) * EK_TILE];
            const float ys = p[(size_t)(3 * a + 1) * EK_TILE];
            const float zs = p[(size_t)(3 * a + 2) * EK_TILE];
            float cc[3 * CT];
#pragma unroll
            for (int q = 0; q < 3 * CT / 4; ++q) {
                const float4 v = ct4[a * (3 * CT / 4) + q];
                cc[4 * q + 0] = v.x;
                cc[4 * q + 1] = v.y;
                cc[4 * q + 2] = v.z;
                cc[4 * q + 3] = v.w;
            }
            const ek_v2f x = (ek_v2f){xs, xs}, y = (ek_v2f){ys, ys},
                         z = (ek_v2f){zs, zs};
#pragma unroll
            for (int c = 0; c < CT / 2; ++c) {
                const ek_v2f cx = (ek_v2f){cc[6 * c + 0], cc[6 * c + 1]},
                             cy = (ek_v2f){cc[6 * c + 2], cc[6 * c + 3]},
                             cz = (ek_v2f){cc[6 * c + 4], cc[6 * c + 5]};
                s2[c][0] = __builtin_elementwise_fma(x, cx, s2[c][0]);
                s2[c][1] = __builtin_elementwise_fma(x, cy, s2[c][1]);
                s2[c][2] = __builtin_elementwise_fma(x, cz, s2[c][2]);
                s2[c][3] = __builtin_elementwise_fma(y, cx, s2[c][3]);
                s2[c][4] = __builtin_elementwise_fma(y, cy, s2[c][4]);
                s2[c][5] = __builtin_elementwise_fma(y, cz, s2[c][5]);
                s2[c][6] = __builtin_elementwise_fma(z, cx, s2[c][6]);
                s2[c][7] = __builtin_elementwise_fma(z, cy, s2[c][7]);
                s2[c][8] = __builtin_elementwise_fma(z, cz, s2[c][8]);
            }
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (c < kc) {
                    float S[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j)
                        S[j] = s2[c / 2][j][c & 1];
                    // +inf once d >= best is certain (ek_qcp.h)
                    const float d = ek_rmsd_from_S_below(S, Gf, gtile[c], A, best);
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

// ===========================================================================
// MFMA variant.  Frames x centers IS a dense contraction: with rows (coord i,
// frame f) and columns (coord j, center c),  S'[(i,f),(j,c)] = sum_a
// x[f,a,i] * y[c,a,j]  is a GEMM with K = atoms.  v_mfma_f32_32x32x2_f32
// accumulates exactly the k-ordered f32 FMA chain of the vector kernels
// (one rounding per product, ascending atom order), so the 3x3 matrices --
// and therefore distances and labels -- are bit-identical.
//
// One wave owns a 32-frame x 32-center tile: 3 x 3 MFMA tiles (coordinate
// blocks), 144 accumulator registers.  Both operands come straight from the
// frame-minor tile layout (the centers are laid out the same way): lane l
// supplies frame/center (l & 31) of atom a0 + (l >> 5).  In every one of the
// nine result tiles the pair (frame, center) sits in the same lane and the same
// register index, so the per-pair 3x3 matrix needs no cross-lane traffic:
// lane l, register r holds pair (f = (r&3) + 8(r>>2) + 4(l>>5), c = l&31).
// A workgroup is 2 x 2 waves = 64 frames x 64 centers per pass over the atoms;
// the running nearest center of each frame is a 64-bit LDS atomic min on
// (distance bits, center index): lowest index wins ties, as util.py:199-203.
// ===========================================================================
typedef float ek_f16v __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(EK_BLOCK, 2)
ek_assign_mfma_kernel(const float *__restrict__ tiles,
                      const double *__restrict__ G, int64_t n, int A,
                      const float *__restrict__ ctiles,
                      const double *__restrict__ Gc, int K,
                      float *__restrict__ dist, int32_t *__restrict__ assign)
{
    // timing-only ablations (bit 0: no quartic solve, bit 1: no memory
    // operands) exist in measurement builds alone: tools/ builds a variant
    // with -DEK_ASSIGN_ABLATE=n; the shipped kernel has none of it
#ifndef EK_ASSIGN_ABLATE
#define EK_ASSIGN_ABLATE 0
#endif
    constexpr int ablate = EK_ASSIGN_ABLATE;
    __shared__ unsigned long long best[64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wf = wave & 1, wc = wave >> 1;
    const int fl = lane & 31, kh = lane >> 5;
    // (+inf, no center): every real key is smaller, and the first look at a
    // frame's best-so-far reads +inf -- which ek_rmsd_from_S_below never
    // abandons against -- not a NaN bit pattern
    if (tid < 64)
        best[tid] = 0x7f800000ffffffffull;
    __syncthreads();

    const int64_t f0 = (int64_t)blockIdx.x * 64 + wf * 32;   // wave's frames
    const float *pa = tiles + (size_t)(f0 / EK_TILE) * 3 * (size_t)A * EK_TILE +
                      (f0 % EK_TILE) + fl;
    const size_t kstride = (size_t)3 * EK_TILE;              // one atom
    const int n_cg = (K + 63) / 64;

    for (int cg = 0; cg < n_cg; ++cg) {
        const int c0 = cg * 64 + wc * 32;                    // wave's centers
        if (c0 < K) {
            const float *pb = ctiles +
                              (size_t)(c0 / EK_TILE) * 3 * (size_t)A * EK_TILE +
                              (c0 % EK_TILE) + fl;
            ek_f16v acc[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][j][r] = 0.f;

            // software pipeline: the raw operands of step t+1 are requested
            // before the nine MFMAs of step t issue and are only touched (the
            // zeroing select for a missing last atom) at the top of step t+1,
            // so the wait for them sits after a full MFMA block.  Loads are
            // unconditional: the address is clamped to the last atom.
            float rx[3], ry[3];
            bool okc = kh < A;
            {
                const int a = okc ? kh : A - 1;
                const float *qa = pa + (size_t)a * kstride;
                const float *qb = pb + (size_t)a * kstride;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    rx[i] = qa[(size_t)i * EK_TILE];
                    ry[i] = qb[(size_t)i * EK_TILE];
                }
            }
            for (int a0 = 0; a0 < A; a0 += 2) {
                float xa[3], yb[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    xa[i] = okc ? rx[i] : 0.f;
                    yb[i] = okc ? ry[i] : 0.f;
                }
                if (ablate & 2) {      // timing only: no memory operands
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        xa[i] = 1.0f + 0.001f * (float)i;
                        yb[i] = 1.0f - 0.001f * (float)i;
                    }
                }
                const int an = a0 + 2 + kh;
                okc = an < A;
                {
                    const int ac = okc ? an : A - 1;
                    const float *qa = pa + (size_t)ac * kstride;
                    const float *qb = pb + (size_t)ac * kstride;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        rx[i] = qa[(size_t)i * EK_TILE];
                        ry[i] = qb[(size_t)i * EK_TILE];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            xa[i], yb[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }

            const int c = c0 + fl;
            const double gc = (c < K) ? Gc[c] : 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f_l = (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int64_t f = f0 + f_l;
                if (f < n && c < K) {
                    const float S[9] = {acc[0][0][r], acc[0][1][r], acc[0][2][r],
                                        acc[1][0][r], acc[1][1][r], acc[1][2][r],
                                        acc[2][0][r], acc[2][1][r], acc[2][2][r]};
                    float d;
                    if (ablate & 1)    // timing only: no quartic solve
                        d = S[0] + S[1] + S[2] + S[3] + S[4] + S[5] + S[6] +
                            S[7] + S[8];
                    else {
                        // the frame's best so far only shrinks, so a stale
                        // read is a valid bound; +inf once d >= it is certain
                        // (ek_qcp.h; equal distances are never abandoned, the
                        // lower center index must still win them)
                        const float seen = __uint_as_float(
                            ((volatile unsigned int *)&best[wf * 32 + f_l])[1]);
                        d = ek_rmsd_from_S_below(S, G[f], gc, A, seen);
                    }
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(d) << 32) |
                        (unsigned int)c;
                    atomicMin(&best[wf * 32 + f_l], key);
                }
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int64_t f = (int64_t)blockIdx.x * 64 + tid;
        if (f < n) {
            const unsigned long long key = best[tid];
            if (K > 0) {
                dist[f] = __uint_as_float((unsigned int)(key >> 32));
                assign[f] = (int32_t)(key & 0xffffffffu);
            } else {
                dist[f] = __builtin_inff();
                assign[f] = 0;
            }
        }
    }
}

void ek_launch_assign_mfma(const float *tiles, const double *G, int64_t n, int A,
                           const float *ctiles, const double *Gc, int32_t K,
                           float *dist, int32_t *assign, hipStream_t s)
{
    if (n <= 0)
        return;
    const int64_t blocks = (n + 63) / 64;
    hipLaunchKernelGGL(ek_assign_mfma_kernel, dim3((unsigned)blocks),
                       dim3(EK_BLOCK), 0, s, tiles, G, n, A, ctiles, Gc, K, dist,
                       assign);
}
