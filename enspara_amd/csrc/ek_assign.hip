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

// ===========================================================================
// The 16x16x4 form (round 4).  The 16-candidate pass of the k-centers rounds
// (ek_pass16.hip) turned out to be power-limited, and faster by 8 % on
// v_mfma_f32_16x16x4_f32 -- four atoms per matrix instruction, a quarter of the
// accumulator traffic -- with its operands as 16-byte loads.  This is that loop
// with K centers for candidates: a wave takes 64 frames x 16 centers per walk
// over the atoms, its A operands from the frames' QUAD copy, its B operands from
// the centers laid out in blocks of 16 like a pass's candidate tile
// (ek_ctile_index); the four waves of a workgroup take the same 64 frames and
// four neighbouring blocks of centers, so the rows come out of the L1 three
// times in four.  Same k-ordered FMA chains => the same bits as the other two
// kernels.  The epilogue is the pass's: float32 certificate for the far pairs
// (against the frame's best so far), the rest queued in LDS and solved densely.
// ===========================================================================
typedef float ek_a4 __attribute__((ext_vector_type(4)));
#define EK_A16_QCAP 256
#define EK_A16_QSTRIDE 12

// centers (centred, [K][3A]) -> blocks of 16 in the candidate-tile layout
__global__ void __launch_bounds__(EK_BLOCK)
ek_cblocks16_kernel(const float *__restrict__ cen_aos, int K, int A,
                    float *__restrict__ cblocks)
{
    const size_t per = (size_t)ek_ctile_atoms(A) * 3 * 16;      // floats per block
    const size_t e = (size_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    const size_t nblk = (size_t)(K + 15) / 16;
    if (e >= nblk * per)
        return;
    // (destination order: one coalesced store per thread; the source is a gather)
    const size_t b = e / per, w = e % per;
    const int q = (int)(w & 3), c = (int)((w >> 2) & 15), kk = (int)((w >> 6) & 3);
    const int k = (int)((w >> 8) % 3), S = (int)((w >> 8) / 3);
    const int a = 16 * S + 4 * q + kk;
    const int64_t cen = (int64_t)b * 16 + c;
    cblocks[e] = (a < A && cen < K) ? cen_aos[(size_t)cen * 3 * A + 3 * a + k] : 0.f;
}

size_t ek_cblocks16_bytes(int32_t K, int A)
{
    return (size_t)((K + 15) / 16) * ek_ctile_atoms(A) * 3 * 16 * sizeof(float);
}

void ek_launch_cblocks16(const float *cen_aos, int32_t K, int A, float *cblocks,
                         hipStream_t s)
{
    const size_t total = ek_cblocks16_bytes(K, A) / sizeof(float);
    if (total == 0)
        return;
    hipLaunchKernelGGL(ek_cblocks16_kernel, dim3((unsigned)((total + EK_BLOCK - 1) / EK_BLOCK)),
                       dim3(EK_BLOCK), 0, s, cen_aos, K, A, cblocks);
}

__global__ void __launch_bounds__(EK_BLOCK, 2)
ek_assign16_kernel(const float *__restrict__ qtiles, const double *__restrict__ G,
                   int64_t n, int A, const float *__restrict__ cblocks,
                   const double *__restrict__ Gc, int K, float *__restrict__ dist,
                   int32_t *__restrict__ assign)
{
    __shared__ unsigned long long best[64];
    __shared__ double s_G[64];
    __shared__ __attribute__((aligned(16)))
    uint32_t s_Q[EK_BLOCK / EK_WAVE][EK_A16_QCAP * EK_A16_QSTRIDE];
    const int tid = threadIdx.x;
    const int lane0 = tid & (EK_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / EK_WAVE);
    const int64_t unit = blockIdx.x;                    // 64 frames of a tile of 256
    const int64_t f0 = unit * EK_WAVE;
    if (tid < 64) {
        // (+inf, no center): every real key is smaller, and +inf is what
        // ek_rmsd_from_S_below never abandons against
        best[tid] = 0x7f800000ffffffffull;
        s_G[tid] = f0 + tid < n ? G[f0 + tid] : 0.0;
    }
    __syncthreads();
    const int NQ = (A + 3) / 4;
    const float *tb = qtiles + (size_t)(unit >> 2) * (size_t)NQ * (3 * EK_TILE * 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)tb, 0, NQ * (3 * EK_TILE * 16), 0x00020000);
    const size_t cb_floats = (size_t)ek_ctile_atoms(A) * 3 * 16;
    const int n_here = (int)(n - f0 < EK_WAVE ? n - f0 : EK_WAVE);
    const int n_cb = (K + 63) / 64;
    // (the rows: default cache policy -- the workgroup's other waves and its next
    // walks read them again; loads past the end of a buffer return zeros)
#define EK_LDR(TR, K_)                                                         \
    __builtin_bit_cast(ek_a4, __builtin_amdgcn_raw_buffer_load_b128(           \
                                  rs, vo, ((TR) * 3 + (K_)) * (EK_TILE * 16), 0))
#define EK_LDC(SS, I)                                                          \
    __builtin_bit_cast(ek_a4, __builtin_amdgcn_raw_buffer_load_b128(           \
                                  cs, co, ((SS) * 3 + (I)) * (EK_WAVE * 16), 0))
    for (int cb = 0; cb < n_cb; ++cb) {
        const int c0 = cb * 64 + 16 * wave;             // this wave's 16 centers
        if (c0 >= K)
            continue;                                   // (no barrier inside this loop)
        // (what depends on the lane is worked out again in every walk: hoisted out
        // of this loop it is two hundred registers of addresses and queue words)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int vo = ((int)(unit & 3) * EK_WAVE + lane) * 16, co = lane * 16;
        const __amdgpu_buffer_rsrc_t cs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(cblocks + (size_t)(c0 / 16) * cb_floats), 0,
            (int)(cb_floats * sizeof(float)), 0x00020000);
        constexpr int DR = 3;
        ek_a4 R[DR + 1][3], Cq[2][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            Cq[0][i] = EK_LDC(0, i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < DR; ++k)
#pragma unroll
            for (int x = 0; x < 3; ++x)
                R[k][x] = EK_LDR(k, x);
        __builtin_amdgcn_sched_barrier(0);
        ek_a4 acc[9][4];        // [3 i + j][group of 16 frames], as in ek_pass16.hip
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                acc[q][g] = ek_a4{0.f, 0.f, 0.f, 0.f};
        // (the trip of ek_pass16.hip: 36 matrix instructions, the next 16 atoms'
        // centers asked for in the first trip of four, the rows three trips ahead)
#define EK_TRIP16(K_, TT)                                                      \
    {                                                                          \
        constexpr int SB = ((K_) / 4) % 2;                                     \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                        \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) {                    \
                _Pragma("unroll") for (int i = 0; i < 3; ++i)                  \
                    acc[3 * i + j][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(  \
                        R[(K_) % 4][i][g], Cq[SB][j][(K_) % 4], acc[3 * i + j][g], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                             \
                if ((K_) % 4 == 0 && g == 0) {                                 \
                    Cq[1 - SB][j] = EK_LDC((TT) / 4 + 1, j);                   \
                    __builtin_amdgcn_sched_barrier(0);                         \
                }                                                              \
                if (g == 1) {                                                  \
                    R[((K_) + DR) % 4][j] = EK_LDR((TT) + DR, j);              \
                    __builtin_amdgcn_sched_barrier(0);                         \
                }                                                              \
            }                                                                  \
        }                                                                      \
    }
        int t0 = 0;
        for (; t0 + 8 <= NQ; t0 += 8) {
            EK_TRIP16(0, t0 + 0)
            EK_TRIP16(1, t0 + 1)
            EK_TRIP16(2, t0 + 2)
            EK_TRIP16(3, t0 + 3)
            EK_TRIP16(4, t0 + 4)
            EK_TRIP16(5, t0 + 5)
            EK_TRIP16(6, t0 + 6)
            EK_TRIP16(7, t0 + 7)
        }
#define EK_REST16(K_)                                                          \
    if (t0 + (K_) < NQ)                                                        \
        EK_TRIP16(K_, t0 + (K_))
        EK_REST16(0)
        EK_REST16(1)
        EK_REST16(2)
        EK_REST16(3)
        EK_REST16(4)
        EK_REST16(5)
        EK_REST16(6)
#undef EK_REST16
#undef EK_TRIP16
        // ---- the lane's 16 pairs: center c0 + lane % 16, frames 16 g + 4 (lane / 16) + r ----
        const int cand = lane & 15;
        const int c = c0 + cand;
        const double gcv = c < K ? Gc[c] : 0.0;
        bool plain = true;
        if (cb > 0) {           // (in the first walk every frame's best is +inf: nothing is far)
            uint32_t *Q = s_Q[wave];
            const float tc = ek_far_t_center((float)gcv);
            const int fr_lim = c < K ? n_here : 0;
            int qn = 0;                         // wave-uniform
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 2) {
                float S[2][9], t[2];
                int fr[2];
                bool far[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = r0 + u;
                    fr[u] = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
#pragma unroll
                    for (int q = 0; q < 9; ++q)
                        S[u][q] = acc[q][r >> 2][r & 3];
                    // (the frame's best so far only shrinks: a stale read is a valid bound)
                    const float seen =
                        __uint_as_float(((volatile unsigned int *)&best[fr[u]])[1]);
                    t[u] = ek_far_t_frame((float)((volatile double *)s_G)[fr[u]], A, seen) + tc;
                }
                ek_far_certified_f32_w<2>(S, t, far);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool need = fr[u] < fr_lim && !far[u];
                    const unsigned long long m = __ballot(need);
                    if (m) {                    // wave-uniform
                        const int pos = qn + __popcll(m & ((1ull << lane) - 1ull));
                        if (need && pos < EK_A16_QCAP) {
                            uint4 *e = (uint4 *)(Q + pos * EK_A16_QSTRIDE);
                            e[0] = make_uint4(__float_as_uint(S[u][0]), __float_as_uint(S[u][1]),
                                              __float_as_uint(S[u][2]), __float_as_uint(S[u][3]));
                            e[1] = make_uint4(__float_as_uint(S[u][4]), __float_as_uint(S[u][5]),
                                              __float_as_uint(S[u][6]), __float_as_uint(S[u][7]));
                            e[2] = make_uint4(__float_as_uint(S[u][8]),
                                              (uint32_t)(cand << 8 | fr[u]), 0u, 0u);
                        }
                        qn += __popcll(m);
                    }
                }
            }
            if (qn <= EK_A16_QCAP) {
                plain = false;
                // (one wave: its LDS accesses are in order, no barrier)
                for (int base = 0; base < qn; base += EK_WAVE) {
                    const int e = base + lane;
                    if (e < qn) {
                        const uint4 *ent = (const uint4 *)(Q + e * EK_A16_QSTRIDE);
                        const uint4 e0 = ent[0], e1 = ent[1], e2 = ent[2];
                        const float S[9] = {
                            __uint_as_float(e0.x), __uint_as_float(e0.y), __uint_as_float(e0.z),
                            __uint_as_float(e0.w), __uint_as_float(e1.x), __uint_as_float(e1.y),
                            __uint_as_float(e1.z), __uint_as_float(e1.w), __uint_as_float(e2.x)};
                        const int cl = (int)(e2.y >> 8), fr = (int)(e2.y & 255u);
                        const float seen =
                            __uint_as_float(((volatile unsigned int *)&best[fr])[1]);
                        const float d = ek_rmsd_from_S_below(S, s_G[fr], Gc[c0 + cl], A, seen);
                        const unsigned long long key =
                            ((unsigned long long)__float_as_uint(d) << 32) |
                            (unsigned int)(c0 + cl);
                        atomicMin(&best[fr], key);
                    }
                }
            }
        }
        if (plain) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                __builtin_amdgcn_sched_barrier(0);
                const int fr = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
                if (fr < n_here && c < K) {
                    float S[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q)
                        S[q] = acc[q][r >> 2][r & 3];
                    // +inf once d >= the frame's best is certain (ek_qcp.h; equal
                    // distances are never abandoned, the lower index must win them)
                    const float seen =
                        __uint_as_float(((volatile unsigned int *)&best[fr])[1]);
                    const float d = ek_rmsd_from_S_below(S, s_G[fr], gcv, A, seen);
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)c;
                    atomicMin(&best[fr], key);
                }
            }
        }
    }
#undef EK_LDC
#undef EK_LDR
    __syncthreads();
    if (tid < 64) {
        const int64_t f = f0 + tid;
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

void ek_launch_assign16(const float *qtiles, const double *G, int64_t n, int A,
                        const float *cblocks, const double *Gc, int32_t K, float *dist,
                        int32_t *assign, hipStream_t s)
{
    if (n <= 0)
        return;
    hipLaunchKernelGGL(ek_assign16_kernel, dim3((unsigned)((n + EK_WAVE - 1) / EK_WAVE)),
                       dim3(EK_BLOCK), 0, s, qtiles, G, n, A, cblocks, Gc, K, dist, assign);
}
