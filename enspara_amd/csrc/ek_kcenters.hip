// ek_kcenters.hip -- the one-center-vs-all-frames distance pass of k-centers,
// fused with the nearest-center update and the farthest-point reduction.
//
// Replaces, for metric 'rmsd' (reference paths relative to /root/reference):
//   dist = distance_method(traj, new_center)        enspara/cluster/kcenters.py:298 / :366
//   inds = dist < distances; distances[inds] = ...  kcenters.py:304-306 / :370-373
//   np.argmax(distances), distances.max()           kcenters.py:282, :226, :332-338
//
// Mapping.  HBM-bound, 1.5 flop/byte: no MFMA.  One lane
// owns FPL consecutive frames; a wave streams its 64*FPL-frame slice of a
// 256-frame tile row by row, so every load instruction of a wave covers one
// contiguous 256*FPL-byte run and a workgroup walks one contiguous tile.
// The 3x3 inner-product matrix of a frame is accumulated by its own lane
// sequentially over atoms (9 FMAs per atom, no cross-lane traffic), which
// fixes the summation order the parity tests rely on.  The center is staged
// once per workgroup in LDS and read back as wave-wide broadcasts.
#include "ek_common.h"
#include "ek_qcp.h"
#include "ek_reduce.h"

#ifndef EK_TRIP
#define EK_TRIP 4      // atoms per loop trip (multiple of 4)
#endif

typedef float ek_f2 __attribute__((ext_vector_type(2)));
typedef float ek_f4 __attribute__((ext_vector_type(4)));
template <int N> struct EkVec;
template <> struct EkVec<1> { typedef float type; };
template <> struct EkVec<2> { typedef ek_f2 type; };
template <> struct EkVec<4> { typedef ek_f4 type; };

// one row load of FPL consecutive frames; NT = non-temporal hint (the frame
// stream is read once per launch and is far larger than L2 + Infinity Cache)
template <int N, bool NT>
__device__ __forceinline__ typename EkVec<N>::type ek_ld(const float *p)
{
    typedef typename EkVec<N>::type vec_t;
    if (NT)
        return __builtin_nontemporal_load((const vec_t *)p);
    return *(const vec_t *)p;
}

template <int N>
__device__ __forceinline__ void ek_unpack(const typename EkVec<N>::type &v,
                                          float (&o)[N])
{
    if constexpr (N == 1) {
        o[0] = v;
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i)
            o[i] = v[i];
    }
}

// ---------------------------------------------------------------------------
// the distance pass
//   FPL   frames per lane (1, 2, 4): load width 4/8/16 bytes per lane
//   MODE  0 = fused k-centers step, 1 = distances only
// A wave owns frames [gw*64*FPL, (gw+1)*64*FPL); a workgroup has 4 waves.
// ---------------------------------------------------------------------------
template <int FPL, int MODE, bool NT>
__global__ void __launch_bounds__(EK_BLOCK)
ek_step_kernel(const float *__restrict__ tiles, const double *__restrict__ G,
               float *__restrict__ dist, int32_t *__restrict__ assign,
               float *__restrict__ out_dist,
               const unsigned char *__restrict__ recs, int n_recs, int64_t n,
               int A, int label, double cutoff,
               EkBlockMax *__restrict__ blockmax, EkHist *__restrict__ hist,
               EkCtl *__restrict__ ctl, const uint8_t *__restrict__ tile_skip)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ctr = lds;                       // 3A floats, padded to 4 atoms
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];

    typedef typename EkVec<FPL>::type vec_t;
    const int tid = threadIdx.x;
    const int lane = tid & (EK_WAVE - 1);
    const int wave = tid / EK_WAVE;

    // ---- winner among the candidate records (kcenters.py:337) --------------
    const size_t rstride = ek_rec_bytes(A);
    int win = 0;
    float wmax = ((const EkRecHdr *)recs)->valid
                     ? ((const EkRecHdr *)recs)->maxdist
                     : -__builtin_inff();
    for (int r = 1; r < n_recs; ++r) {
        const EkRecHdr *h = (const EkRecHdr *)(recs + (size_t)r * rstride);
        const float m = h->valid ? h->maxdist : -__builtin_inff();
        if (m > wmax) {
            wmax = m;
            win = r;
        }
    }
    win = __builtin_amdgcn_readfirstlane(win);
    const EkRecHdr *wh = (const EkRecHdr *)(recs + (size_t)win * rstride);
    if (MODE == 0) {
        // stop rule of kcenters.py:217: continue only while maxdist > cutoff
        if (!((double)wmax > cutoff)) {
            if (blockIdx.x == 0 && tid == 0)
                ctl->stopped = 1;
            return;
        }
    }
    const double Gc = wh->trace;
    {
        const float *wc = (const float *)(wh + 1);
        const int A4 = (A + EK_TRIP - 1) / EK_TRIP * EK_TRIP;
        for (int j = tid; j < 3 * A4; j += EK_BLOCK)
            ctr[j] = (j < 3 * A) ? wc[j] : 0.f;
    }
    if (MODE == 0 && blockIdx.x == 0 && tid == 0) {
        hist[label].gidx = wh->gidx;
        hist[label].dist = wmax;
        hist[label].set = 1;
        ctl->n_done = label + 1;
    }
    __syncthreads();

    // ---- this wave's slice ---------------------------------------------------
    const int64_t gw = (int64_t)blockIdx.x * (EK_BLOCK / EK_WAVE) + wave;
    const int64_t fw = gw * (EK_WAVE * FPL);          // first frame of the wave
    const int64_t f0 = fw + (int64_t)lane * FPL;      // first frame of the lane
    float bestv = -__builtin_inff();
    uint32_t besti = 0xffffffffu;

    // triangle inequality (kcenters.py:287-296, ek_ti_* below): no frame of this
    // tile can come closer to the new center than it is to its own, so its
    // coordinates are not read; the farthest point of the slice is still needed
    if (MODE == 0 && tile_skip && fw < n && tile_skip[fw / EK_TILE]) {
#pragma unroll
        for (int q = 0; q < FPL; ++q) {
            const int64_t f = f0 + q;
            if (f < n) {
                const float cur = dist[f];
                if (ek_better(cur, (uint32_t)f, bestv, besti)) {
                    bestv = cur;
                    besti = (uint32_t)f;
                }
            }
        }
    } else if (fw < n) {
        const int64_t tile = fw / EK_TILE;
        const int in_tile = (int)(fw % EK_TILE) + lane * FPL;
        const float *p = tiles + (size_t)tile * 3 * (size_t)A * EK_TILE + in_tile;

        float s[FPL][9];
#pragma unroll
        for (int q = 0; q < FPL; ++q)
#pragma unroll
            for (int j = 0; j < 9; ++j)
                s[q][j] = 0.f;

        // EK_TRIP atoms per trip: 3*EK_TRIP row loads in flight, broadcast
        // LDS reads of the center (ds_read_b128)
        const float4 *ctr4 = (const float4 *)ctr;
        const int A4 = A - A % EK_TRIP;
        int a = 0;
        for (; a < A4; a += EK_TRIP) {
            vec_t v[3 * EK_TRIP];
#pragma unroll
            for (int r = 0; r < 3 * EK_TRIP; ++r)
                v[r] = ek_ld<FPL, NT>(p + (size_t)(3 * a + r) * EK_TILE);
#ifdef EK_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
            float c[3 * EK_TRIP];
#pragma unroll
            for (int q = 0; q < 3 * EK_TRIP / 4; ++q) {
                const float4 cq = ctr4[(3 * a) / 4 + q];
                c[4 * q + 0] = cq.x;
                c[4 * q + 1] = cq.y;
                c[4 * q + 2] = cq.z;
                c[4 * q + 3] = cq.w;
            }
#pragma unroll
            for (int u = 0; u < EK_TRIP; ++u) {
                float x[FPL], y[FPL], z[FPL];
                ek_unpack<FPL>(v[3 * u + 0], x);
                ek_unpack<FPL>(v[3 * u + 1], y);
                ek_unpack<FPL>(v[3 * u + 2], z);
                const float cx = c[3 * u + 0], cy = c[3 * u + 1],
                            cz = c[3 * u + 2];
#pragma unroll
                for (int q = 0; q < FPL; ++q) {
                    s[q][0] = __builtin_fmaf(x[q], cx, s[q][0]);
                    s[q][1] = __builtin_fmaf(x[q], cy, s[q][1]);
                    s[q][2] = __builtin_fmaf(x[q], cz, s[q][2]);
                    s[q][3] = __builtin_fmaf(y[q], cx, s[q][3]);
                    s[q][4] = __builtin_fmaf(y[q], cy, s[q][4]);
                    s[q][5] = __builtin_fmaf(y[q], cz, s[q][5]);
                    s[q][6] = __builtin_fmaf(z[q], cx, s[q][6]);
                    s[q][7] = __builtin_fmaf(z[q], cy, s[q][7]);
                    s[q][8] = __builtin_fmaf(z[q], cz, s[q][8]);
                }
            }
        }
        for (; a < A; ++a) {
            float x[FPL], y[FPL], z[FPL];
            ek_unpack<FPL>(ek_ld<FPL, NT>(p + (size_t)(3 * a + 0) * EK_TILE), x);
            ek_unpack<FPL>(ek_ld<FPL, NT>(p + (size_t)(3 * a + 1) * EK_TILE), y);
            ek_unpack<FPL>(ek_ld<FPL, NT>(p + (size_t)(3 * a + 2) * EK_TILE), z);
            const float cx = ctr[3 * a + 0], cy = ctr[3 * a + 1],
                        cz = ctr[3 * a + 2];
#pragma unroll
            for (int q = 0; q < FPL; ++q) {
                s[q][0] = __builtin_fmaf(x[q], cx, s[q][0]);
                s[q][1] = __builtin_fmaf(x[q], cy, s[q][1]);
                s[q][2] = __builtin_fmaf(x[q], cz, s[q][2]);
                s[q][3] = __builtin_fmaf(y[q], cx, s[q][3]);
                s[q][4] = __builtin_fmaf(y[q], cy, s[q][4]);
                s[q][5] = __builtin_fmaf(y[q], cz, s[q][5]);
                s[q][6] = __builtin_fmaf(z[q], cx, s[q][6]);
                s[q][7] = __builtin_fmaf(z[q], cy, s[q][7]);
                s[q][8] = __builtin_fmaf(z[q], cz, s[q][8]);
            }
        }

        // ---- per-frame epilogue: quartic, update, running arg-max -----------
#pragma unroll
        for (int q = 0; q < FPL; ++q) {
            const int64_t f = f0 + q;
            if (f < n) {
                if (MODE == 1) {
                    out_dist[f] = ek_rmsd_from_S(s[q], G[f], Gc, A);
                } else {
                    float cur = dist[f];
                    // +inf once d >= cur is certain (ek_qcp.h)
                    const float d = ek_rmsd_from_S_below(s[q], G[f], Gc, A, cur);
                    if (d < cur) {              // strict <: kcenters.py:304
                        cur = d;
                        dist[f] = d;
                        assign[f] = label;
                    }
                    if (ek_better(cur, (uint32_t)f, bestv, besti)) {
                        bestv = cur;
                        besti = (uint32_t)f;
                    }
                }
            }
        }
    }

    if (MODE == 0) {
        ek_wave_argmax(bestv, besti);
        if (lane == 0) {
            red_v[wave] = bestv;
            red_i[wave] = besti;
        }
        __syncthreads();
        if (tid == 0) {
            float v = red_v[0];
            uint32_t i = red_i[0];
#pragma unroll
            for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
                if (ek_better(red_v[w], red_i[w], v, i)) {
                    v = red_v[w];
                    i = red_i[w];
                }
            blockmax[blockIdx.x].val = v;
            blockmax[blockIdx.x].idx = i;
        }
    }
}

int ek_step_blocks(int fpl, int64_t n)
{
    const int64_t per_block = (int64_t)EK_BLOCK * fpl;
    return (int)((n + per_block - 1) / per_block);
}

template <int FPL, int MODE, bool NT>
static void ek_launch_step_t(const float *tiles, const double *G, float *dist,
                             int32_t *assign, float *out_dist,
                             const unsigned char *recs, int n_recs, int64_t n,
                             int A, int label, double cutoff,
                             EkBlockMax *blockmax, EkHist *hist, EkCtl *ctl,
                             const uint8_t *tile_skip, hipStream_t s)
{
    const int blocks = ek_step_blocks(FPL, n);
    if (blocks <= 0)
        return;
    const size_t lds = (size_t)3 * ((A + EK_TRIP - 1) / EK_TRIP * EK_TRIP) * sizeof(float);
    hipLaunchKernelGGL((ek_step_kernel<FPL, MODE, NT>), dim3(blocks),
                       dim3(EK_BLOCK), lds, s, tiles, G, dist, assign,
                       out_dist, recs, n_recs, n, A, label, cutoff, blockmax,
                       hist, ctl, tile_skip);
}

void ek_launch_step(int fpl, int mode, int nt, const float *tiles, const double *G,
                    float *dist, int32_t *assign, float *out_dist,
                    const unsigned char *recs, int n_recs, int64_t n, int A,
                    int label, double cutoff, EkBlockMax *blockmax,
                    EkHist *hist, EkCtl *ctl, hipStream_t s,
                    const uint8_t *tile_skip)
{
#define EK_GO(F, M)                                                            \
    do {                                                                       \
        if (nt)                                                                \
            ek_launch_step_t<F, M, true>(tiles, G, dist, assign, out_dist,     \
                                         recs, n_recs, n, A, label, cutoff,    \
                                         blockmax, hist, ctl, tile_skip, s);   \
        else                                                                   \
            ek_launch_step_t<F, M, false>(tiles, G, dist, assign, out_dist,    \
                                          recs, n_recs, n, A, label, cutoff,   \
                                          blockmax, hist, ctl, tile_skip, s);  \
    } while (0)
    if (mode == 0) {
        if (fpl == 4) EK_GO(4, 0);
        else if (fpl == 2) EK_GO(2, 0);
        else EK_GO(1, 0);
    } else {
        if (fpl == 4) EK_GO(4, 1);
        else if (fpl == 2) EK_GO(2, 1);
        else EK_GO(1, 1);
    }
#undef EK_GO
}

// ---------------------------------------------------------------------------
// pick: reduce to the shard's (max, first index) and publish its record
// ---------------------------------------------------------------------------
#define EK_PICK_THREADS 1024

__global__ void __launch_bounds__(EK_PICK_THREADS)
ek_pick_kernel(const EkBlockMax *__restrict__ blockmax, int n_blocks,
               const float *__restrict__ dist, const float *__restrict__ tiles,
               const double *__restrict__ G, int64_t n, int A,
               int64_t global_offset, unsigned char *__restrict__ rec,
               EkCtl *__restrict__ ctl)
{
    __shared__ float red_v[EK_PICK_THREADS / EK_WAVE];
    __shared__ uint32_t red_i[EK_PICK_THREADS / EK_WAVE];
    __shared__ uint32_t win_i;
    __shared__ float win_v;
    const int tid = threadIdx.x;
    // a stopped step wrote no partials: leave the record as it is
    if (blockmax && ctl->stopped)
        return;
    float v = -__builtin_inff();
    uint32_t i = 0xffffffffu;
    if (blockmax) {
        for (int b = tid; b < n_blocks; b += EK_PICK_THREADS) {
            const EkBlockMax m = blockmax[b];
            if (ek_better(m.val, m.idx, v, i)) {
                v = m.val;
                i = m.idx;
            }
        }
    } else {
        for (int64_t f = tid; f < n; f += EK_PICK_THREADS) {
            const float d = dist[f];
            if (ek_better(d, (uint32_t)f, v, i)) {
                v = d;
                i = (uint32_t)f;
            }
        }
    }
    ek_wave_argmax(v, i);
    if ((tid & 63) == 0) {
        red_v[tid / 64] = v;
        red_i[tid / 64] = i;
    }
    __syncthreads();
    if (tid < 64) {
        v = (tid < EK_PICK_THREADS / EK_WAVE) ? red_v[tid] : -__builtin_inff();
        i = (tid < EK_PICK_THREADS / EK_WAVE) ? red_i[tid] : 0xffffffffu;
        ek_wave_argmax(v, i);
        if (tid == 0) {
            win_v = v;
            win_i = i;
        }
    }
    __syncthreads();
    const uint32_t wi = win_i;
    EkRecHdr *h = (EkRecHdr *)rec;
    float *coords = (float *)(h + 1);
    if (n <= 0 || wi == 0xffffffffu) {
        if (tid == 0) {
            h->maxdist = -__builtin_inff();
            h->valid = 0;
            h->gidx = -1;
            h->trace = 0.0;
            h->reserved = 0;
            ctl->last_max = -__builtin_inff();
        }
        return;
    }
    if (tid == 0) {
        h->maxdist = win_v;
        h->valid = 1;
        h->gidx = global_offset + (int64_t)wi;
        h->trace = G[wi];
        h->reserved = 0;
        ctl->last_max = win_v;
    }
    const float *p = tiles + (size_t)(wi / EK_TILE) * 3 * (size_t)A * EK_TILE +
                     (wi % EK_TILE);
    for (int r = tid; r < 3 * A; r += EK_PICK_THREADS)
        coords[r] = p[(size_t)r * EK_TILE];
}

void ek_launch_pick(const EkBlockMax *blockmax, int n_blocks,
                    const float *dist, const float *tiles, const double *G,
                    int64_t n, int A, int64_t global_offset,
                    unsigned char *rec, EkCtl *ctl, hipStream_t s)
{
    hipLaunchKernelGGL(ek_pick_kernel, dim3(1), dim3(EK_PICK_THREADS), 0, s,
                       blockmax, n_blocks, dist, tiles, G, n, A, global_offset,
                       rec, ctl);
}

// ---------------------------------------------------------------------------
// records for explicitly chosen centers
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(EK_BLOCK)
ek_record_from_frame_kernel(const float *__restrict__ tiles,
                            const double *__restrict__ G, int A, int64_t idx,
                            const int64_t *__restrict__ idx_dev,
                            int64_t global_offset,
                            unsigned char *__restrict__ rec)
{
    if (idx_dev)
        idx = *idx_dev;
    EkRecHdr *h = (EkRecHdr *)rec;
    float *coords = (float *)(h + 1);
    if (threadIdx.x == 0) {
        h->maxdist = __builtin_inff();
        h->valid = 1;
        h->gidx = global_offset + idx;
        h->trace = G[idx];
        h->reserved = 0;
    }
    const float *p = tiles + (size_t)(idx / EK_TILE) * 3 * (size_t)A * EK_TILE +
                     (idx % EK_TILE);
    for (int r = threadIdx.x; r < 3 * A; r += EK_BLOCK)
        coords[r] = p[(size_t)r * EK_TILE];
}

void ek_launch_record_from_frame(const float *tiles, const double *G, int A,
                                 int64_t local_idx, const int64_t *idx_dev,
                                 int64_t global_offset, unsigned char *rec,
                                 hipStream_t s)
{
    hipLaunchKernelGGL(ek_record_from_frame_kernel, dim3(1), dim3(EK_BLOCK), 0,
                       s, tiles, G, A, local_idx, idx_dev, global_offset, rec);
}

__global__ void __launch_bounds__(EK_BLOCK)
ek_record_from_center_kernel(const float *__restrict__ center_aos,
                             const double *__restrict__ Gc, int A,
                             unsigned char *__restrict__ rec)
{
    EkRecHdr *h = (EkRecHdr *)rec;
    float *coords = (float *)(h + 1);
    if (threadIdx.x == 0) {
        h->maxdist = __builtin_inff();
        h->valid = 1;
        h->gidx = -1;
        h->trace = Gc[0];
        h->reserved = 0;
    }
    for (int r = threadIdx.x; r < 3 * A; r += EK_BLOCK)
        coords[r] = center_aos[r];
}

void ek_launch_record_from_center(const float *center_aos, const double *Gc,
                                  int A, unsigned char *rec, hipStream_t s)
{
    hipLaunchKernelGGL(ek_record_from_center_kernel, dim3(1), dim3(EK_BLOCK), 0,
                       s, center_aos, Gc, A, rec);
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(EK_BLOCK)
ek_fill_state_kernel(float *__restrict__ dist, int32_t *__restrict__ assign,
                     int64_t n, float d, int32_t a)
{
    const int64_t i = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (i < n) {
        dist[i] = d;
        assign[i] = a;
    }
}

void ek_launch_fill_state(float *dist, int32_t *assign, int64_t n, float d,
                          int32_t a, hipStream_t s)
{
    if (n <= 0)
        return;
    const int64_t blocks = (n + EK_BLOCK - 1) / EK_BLOCK;
    hipLaunchKernelGGL(ek_fill_state_kernel, dim3((unsigned)blocks),
                       dim3(EK_BLOCK), 0, s, dist, assign, n, d, a);
}

// ---------------------------------------------------------------------------
// triangle inequality for the one-center step (reference kcenters.py:287-296:
// `use_triangle_inequality`): a frame whose own center is at least twice its
// distance away from the new center cannot come closer to the new center than
// it is to its own, so its distance need not be computed.  On the device the
// unit that can be left out is a tile of 256 frames (its 3 A KiB of coordinates
// are then not read): ek_ti_center_kernel gives the distance of every existing
// center to the new one, ek_ti_tiles_kernel marks the tiles all of whose frames
// pass the test.  Minimal RMSD is a metric; the computed values carry rounding
// error, so the test keeps a margin (0.1 % + 1e-3) far above it -- a tile that
// is not skipped is simply computed, results never depend on the marks.
// ---------------------------------------------------------------------------
// one wave per existing center (its summation order is free: the value is only
// compared, with a margin)
__global__ void __launch_bounds__(EK_BLOCK)
ek_ti_center_kernel(const float *__restrict__ aos, const double *__restrict__ G,
                    int A, const EkHist *__restrict__ hist, int k, int64_t goff,
                    const unsigned char *__restrict__ rec,
                    float *__restrict__ Dnew)
{
    const int lane = threadIdx.x & (EK_WAVE - 1);
    const int l = blockIdx.x * (EK_BLOCK / EK_WAVE) + threadIdx.x / EK_WAVE;
    if (l >= k)
        return;
    const int64_t f = hist[l].gidx - goff;
    const float *x = aos + (size_t)f * 3 * A;
    const float *y = (const float *)(rec + sizeof(EkRecHdr));
    float S[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = lane; a < A; a += EK_WAVE) {
        const float x0 = x[3 * a], x1 = x[3 * a + 1], x2 = x[3 * a + 2];
        const float y0 = y[3 * a], y1 = y[3 * a + 1], y2 = y[3 * a + 2];
        S[0] += x0 * y0; S[1] += x0 * y1; S[2] += x0 * y2;
        S[3] += x1 * y0; S[4] += x1 * y1; S[5] += x1 * y2;
        S[6] += x2 * y0; S[7] += x2 * y1; S[8] += x2 * y2;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            S[q] += __shfl_xor(S[q], off, 64);
    if (lane == 0)
        Dnew[l] = ek_rmsd_from_S(S, G[f], ((const EkRecHdr *)rec)->trace, A);
}

// tile t is skipped iff every frame f of it has a center (label >= 0) with
// Dnew[label] >= 2 dist[f] (1 + 1e-3) + 1e-3;  stats[0] += tiles, [1] += skipped
__global__ void __launch_bounds__(EK_BLOCK)
ek_ti_tiles_kernel(const float *__restrict__ dist,
                   const int32_t *__restrict__ assign, int64_t n, int k,
                   const float *__restrict__ Dnew,
                   const EkCtl *__restrict__ ctl, uint8_t *__restrict__ tile_skip,
                   unsigned long long *__restrict__ stats)
{
    __shared__ int any_needed;
    if (threadIdx.x == 0)
        any_needed = 0;
    __syncthreads();
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    bool need = false;
    if (f < n) {
        const int32_t a = assign[f];
        need = a < 0 || a >= k ||
               !(Dnew[a] >= 2.0f * dist[f] * 1.001f + 1e-3f);
    }
    if (need)
        any_needed = 1;         // benign race: everybody writes 1
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool live = !ctl->stopped;
        tile_skip[blockIdx.x] = (live && !any_needed) ? 1 : 0;
        if (live) {
            atomicAdd(&stats[0], 1ull);
            if (!any_needed)
                atomicAdd(&stats[1], 1ull);
        }
    }
}

// The same for a SHARDED iteration (reference kcenters.py:351-364): the existing
// centers are other shards' frames as often as this one's, so they come from a
// table of the centers accepted so far (centred coordinates + trace, a row per
// label, filled here as they are accepted), and the new center is the winner
// among the gathered records (largest distance, lowest rank on ties: the rule
// ek_step_kernel applies).  Wave l < k: Dnew[l]; wave k: the winner's row -> table.
__global__ void __launch_bounds__(EK_BLOCK)
ek_ti_center_tab_kernel(float *__restrict__ tab, double *__restrict__ tabG, int A,
                        int k, const unsigned char *__restrict__ recs, int n_recs,
                        float *__restrict__ Dnew)
{
    const int lane = threadIdx.x & (EK_WAVE - 1);
    const int l = blockIdx.x * (EK_BLOCK / EK_WAVE) + threadIdx.x / EK_WAVE;
    if (l > k)
        return;
    const size_t rstride = ek_rec_bytes(A);
    int win = 0;
    float wmax = ((const EkRecHdr *)recs)->valid ? ((const EkRecHdr *)recs)->maxdist
                                                  : -__builtin_inff();
    for (int r = 1; r < n_recs; ++r) {
        const EkRecHdr *h = (const EkRecHdr *)(recs + (size_t)r * rstride);
        const float m = h->valid ? h->maxdist : -__builtin_inff();
        if (m > wmax) {
            wmax = m;
            win = r;
        }
    }
    const EkRecHdr *wh = (const EkRecHdr *)(recs + (size_t)win * rstride);
    const float *y = (const float *)(wh + 1);
    if (l == k) {               // the new center's own row
        for (int q = lane; q < 3 * A; q += EK_WAVE)
            tab[(size_t)k * 3 * A + q] = y[q];
        if (lane == 0)
            tabG[k] = wh->trace;
        return;
    }
    const float *x = tab + (size_t)l * 3 * A;
    float S[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = lane; a < A; a += EK_WAVE) {
        const float x0 = x[3 * a], x1 = x[3 * a + 1], x2 = x[3 * a + 2];
        const float y0 = y[3 * a], y1 = y[3 * a + 1], y2 = y[3 * a + 2];
        S[0] += x0 * y0; S[1] += x0 * y1; S[2] += x0 * y2;
        S[3] += x1 * y0; S[4] += x1 * y1; S[5] += x1 * y2;
        S[6] += x2 * y0; S[7] += x2 * y1; S[8] += x2 * y2;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            S[q] += __shfl_xor(S[q], off, 64);
    if (lane == 0)
        Dnew[l] = ek_rmsd_from_S(S, tabG[l], wh->trace, A);
}

void ek_launch_ti_tab(float *tab, double *tabG, int A, int k,
                      const unsigned char *recs, int n_recs, float *Dnew,
                      const float *dist, const int32_t *assign, int64_t n,
                      const EkCtl *ctl, uint8_t *tile_skip, unsigned long long *stats,
                      hipStream_t s)
{
    const int per = EK_BLOCK / EK_WAVE;
    hipLaunchKernelGGL(ek_ti_center_tab_kernel, dim3((k + 1 + per - 1) / per),
                       dim3(EK_BLOCK), 0, s, tab, tabG, A, k, recs, n_recs, Dnew);
    if (n > 0 && k > 0)
        hipLaunchKernelGGL(ek_ti_tiles_kernel,
                           dim3((unsigned)((n + EK_BLOCK - 1) / EK_BLOCK)),
                           dim3(EK_BLOCK), 0, s, dist, assign, n, k, Dnew, ctl,
                           tile_skip, stats);
}

void ek_launch_ti(const float *aos, const double *G, int A, const EkHist *hist,
                  int k, int64_t goff, const unsigned char *rec, float *Dnew,
                  const float *dist, const int32_t *assign, int64_t n,
                  const EkCtl *ctl, uint8_t *tile_skip, unsigned long long *stats,
                  hipStream_t s)
{
    if (n <= 0 || k <= 0)
        return;
    const int per = EK_BLOCK / EK_WAVE;
    hipLaunchKernelGGL(ek_ti_center_kernel, dim3((k + per - 1) / per),
                       dim3(EK_BLOCK), 0, s, aos, G, A, hist, k, goff, rec, Dnew);
    hipLaunchKernelGGL(ek_ti_tiles_kernel,
                       dim3((unsigned)((n + EK_BLOCK - 1) / EK_BLOCK)),
                       dim3(EK_BLOCK), 0, s, dist, assign, n, k, Dnew, ctl,
                       tile_skip, stats);
}
