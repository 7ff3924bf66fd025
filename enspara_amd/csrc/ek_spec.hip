// ek_spec.hip -- k-centers with several candidate centers per pass over the
// frames ("speculative" k-centers).  Exactly the sequential algorithm of
// enspara/cluster/kcenters.py:217-231 (same centers, same order, same labels
// and distances, bit for bit), reorganised around the fact that the
// one-center pass is HBM-bound with ~90 % of the vector ALUs idle:
//
//   round:  the frames are streamed ONCE against T candidate centers: the
//           true farthest point (candidate 0, applied immediately, exactly as
//           ek_step_kernel does) and T-1 further far points, whose distance
//           vectors are kept (4 bytes per frame each).
//   then:   the farthest point of the updated distances is found again
//           (kcenters.py:282).  If it is one of the stored candidates its
//           distances are already there: applying them is a 16-byte-per-frame
//           update instead of a (12 A + 20)-byte pass.  This repeats
//           until the farthest point is not a stored candidate; then the next
//           round starts.
//
// Every accepted center is the first-index arg-max of the current distances and
// its distances come from the same per-pair FMA chain + quartic solve as the
// one-center kernel, so nothing about the result depends on which candidates
// were guessed -- only the number of passes does.  All decisions are taken on
// the device by single-workgroup kernels that are the sole writers of the plan
// the following kernels read, so no kernel reads state another workgroup of the
// same launch writes.
#include "ek_common.h"
#include "ek_qcp.h"
#include "ek_reduce.h"

// atoms per trip of the pass kernel's main loop, and how many trips ahead of
// the FMAs the row loads are issued (register ring of DIST + 1 trips)
#ifndef EK_SPEC_TRIP
#define EK_SPEC_TRIP 4
#endif
#ifndef EK_SPEC_DIST
#define EK_SPEC_DIST 1
#endif
// waves per SIMD asked of the register allocator for the 8-candidate kernel
#ifndef EK_SPEC_WAVES8
#define EK_SPEC_WAVES8 3
#endif

// ---------------------------------------------------------------------------
// plan: choose the round's candidates among the records on offer
// (n_recs = ranks x per-rank candidates), ordered by (distance desc, global
// index asc) -- position 0 is the global first-index arg-max (kcenters.py:337:
// lowest rank wins ties; ranks own ascending contiguous blocks).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(EK_WAVE)
ek_plan_kernel(const unsigned char *__restrict__ recs, int n_recs, int A, int T,
               double cutoff, EkPlan *__restrict__ plan,
               EkHist *__restrict__ hist, EkCtl *__restrict__ ctl)
{
    if (threadIdx.x != 0)
        return;
    const size_t rstride = ek_rec_bytes(A);
    plan->go = 0;
    plan->apply = -1;
    plan->miss = 1;
    plan->used = 0;
    plan->teff = 0;
    if (ctl->stopped || ctl->n_done >= ctl->limit)
        return;
    unsigned long long taken = 0;          // n_recs <= 64
    int teff = 0;
    for (int j = 0; j < T; ++j) {
        int best = -1;
        float bv = 0.f;
        long long bg = 0;
        for (int r = 0; r < n_recs; ++r) {
            if (taken & (1ull << r))
                continue;
            const EkRecHdr *h = (const EkRecHdr *)(recs + (size_t)r * rstride);
            if (!h->valid)
                continue;
            if (best < 0 || h->maxdist > bv ||
                (h->maxdist == bv && h->gidx < bg)) {
                best = r;
                bv = h->maxdist;
                bg = h->gidx;
            }
        }
        if (best < 0)
            break;
        taken |= 1ull << best;
        plan->src[j] = best;
        plan->gidx[j] = bg;
        plan->maxdist[j] = bv;
        plan->trace[j] =
            ((const EkRecHdr *)(recs + (size_t)best * rstride))->trace;
        ++teff;
    }
    if (teff == 0)
        return;
    // stop rule of kcenters.py:217
    if (!((double)plan->maxdist[0] > cutoff)) {
        ctl->stopped = 1;
        return;
    }
    const int label = ctl->n_done;
    plan->go = 1;
    plan->teff = teff;
    plan->label = label;
    plan->used = 1;
    plan->miss = 0;
    hist[label].gidx = plan->gidx[0];
    hist[label].dist = plan->maxdist[0];
    hist[label].set = 1;
    ctl->n_done = label + 1;
    ctl->n_rounds = ctl->n_rounds + 1;
}

void ek_launch_plan(const unsigned char *recs, int n_recs, int A, int T,
                    double cutoff, EkPlan *plan, EkHist *hist, EkCtl *ctl,
                    hipStream_t s)
{
    hipLaunchKernelGGL(ek_plan_kernel, dim3(1), dim3(EK_WAVE), 0, s, recs,
                       n_recs, A, T, cutoff, plan, hist, ctl);
}

// ---------------------------------------------------------------------------
// the pass: every frame against the round's T candidates
// one lane = one frame; candidates staged in LDS as ctile[atom][cand][xyz]
// ---------------------------------------------------------------------------
// waves per SIMD asked of the register allocator: 5 at T = 4, 3 at T = 8
// UPD = true: a k-centers round (candidate 0 updates the state, the others'
// distances go to vecs[0..T-2]).  UPD = false: distances only, all T of them to
// vecs[0..T-1] (PAM proposal prefetch); dist / assign / blockmax are not touched.
template <int T, bool UPD>
__global__ void __launch_bounds__(EK_BLOCK, (T <= 4) ? 5 : EK_SPEC_WAVES8)
ek_pass_kernel(const float *__restrict__ tiles, const double *__restrict__ G,
               float *__restrict__ dist, int32_t *__restrict__ assign,
               float *__restrict__ vecs,   // [T-1][n_pad] stored distance vectors
               int64_t n, int64_t n_pad, int A,
               const unsigned char *__restrict__ recs,
               const EkPlan *__restrict__ plan,
               EkBlockMax *__restrict__ blockmax)
{
    extern __shared__ __attribute__((aligned(16))) float ctile[];
    __shared__ double gtile[T];
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];
    if (!plan->go)
        return;
    const int tid = threadIdx.x;
    const int teff = plan->teff;
    const int label = plan->label;
    const size_t rstride = ek_rec_bytes(A);

    // stage the candidates: global reads run along each record (coalesced)
    for (int j = tid; j < 3 * A * T; j += EK_BLOCK) {
        const int c = j / (3 * A), r = j % (3 * A);
        float v = 0.f;
        if (c < teff) {
            const float *src = (const float *)(recs +
                (size_t)plan->src[c] * rstride + sizeof(EkRecHdr));
            v = src[r];
        }
        // candidates in pairs: [atom][pair][xyz][2], so that one packed FMA
        // (v_pk_fma_f32) serves candidates 2p and 2p+1
        ctile[(r / 3) * (3 * T) + (c / 2) * 6 + (r % 3) * 2 + (c & 1)] = v;
    }
    if (tid < T) {
        double g = 0.0;
        if (tid < teff)
            g = ((const EkRecHdr *)(recs + (size_t)plan->src[tid] * rstride))
                    ->trace;
        gtile[tid] = g;
    }
    __syncthreads();

    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + tid;
    // one workgroup = one tile: a wave-uniform base (scalar registers) plus the
    // lane's 32-bit offset, so the row addresses cost no vector registers
    static_assert(EK_BLOCK == EK_TILE, "one workgroup per tile");
    const float *tb = tiles + (size_t)blockIdx.x * 3 * (size_t)A * EK_TILE;
    // s2[p][j] = (S_j of candidate 2p, S_j of candidate 2p+1): each component is
    // its own IEEE FMA chain in ascending atom order, exactly as the one-center
    // kernels compute it; packing only doubles the FMA rate (the T = 8 pass was
    // bound by the vector ALU, not by HBM, with scalar FMAs).
    ek_v2f s2[T / 2][9];
#pragma unroll
    for (int c = 0; c < T / 2; ++c)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            s2[c][j] = (ek_v2f){0.f, 0.f};

    const float4 *ct4 = (const float4 *)ctile;
    constexpr int Q = 3 * T / 4;            // float4 per atom (T multiple of 4)
    // Occupancy is only 3-4 waves per SIMD here (9*T accumulators), and a trip's
    // FMAs take a fraction of the HBM latency, so the rows of trip t + DIST are
    // requested before the FMAs of trip t are issued: a ring of DIST + 1
    // register buffers, unrolled so that every buffer index is a constant.
    constexpr int TRIP = EK_SPEC_TRIP;      // atoms per trip
    constexpr int DIST = EK_SPEC_DIST;
    constexpr int NB = DIST + 1;
    const int n_trip = (A + TRIP - 1) / TRIP;
    float bx[NB][TRIP], by[NB][TRIP], bz[NB][TRIP];
#define EK_ROWS(BUF, TRIP_INDEX)                                               \
    _Pragma("unroll") for (int u = 0; u < TRIP; ++u) {                         \
        int au = (TRIP_INDEX) * TRIP + u;                                      \
        au = (au < A) ? au : A - 1;     /* clamp: in-bounds, masked below */   \
        const float *row = tb + (size_t)(3 * au) * EK_TILE;                    \
        bx[BUF][u] = __builtin_nontemporal_load(row + tid);                    \
        by[BUF][u] = __builtin_nontemporal_load(row + EK_TILE + tid);          \
        bz[BUF][u] = __builtin_nontemporal_load(row + 2 * EK_TILE + tid);      \
    }
#pragma unroll
    for (int k = 0; k < DIST; ++k)
        EK_ROWS(k, k)
    for (int t0 = 0; t0 < n_trip; t0 += NB) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int t = t0 + k;
            EK_ROWS((k + DIST) % NB, t + DIST)
            const int a0 = t * TRIP;
#pragma unroll
            for (int u = 0; u < TRIP; ++u) {
                const int a = a0 + u;
                if (a < A) {                        // wave-uniform
                    float cc[3 * T];
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const float4 v = ct4[a * Q + q];
                        cc[4 * q + 0] = v.x;
                        cc[4 * q + 1] = v.y;
                        cc[4 * q + 2] = v.z;
                        cc[4 * q + 3] = v.w;
                    }
                    const ek_v2f x = (ek_v2f){bx[k][u], bx[k][u]},
                                 y = (ek_v2f){by[k][u], by[k][u]},
                                 z = (ek_v2f){bz[k][u], bz[k][u]};
#pragma unroll
                    for (int c = 0; c < T / 2; ++c) {
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
            }
        }
    }
#undef EK_ROWS

    if (!UPD) {
        if (f < n) {
            const double Gf = G[f];
#pragma unroll
            for (int c = 0; c < T; ++c) {
                __builtin_amdgcn_sched_barrier(0);
                if (c < teff) {
                    float S[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j)
                        S[j] = s2[c / 2][j][c & 1];
                    vecs[(size_t)c * n_pad + f] =
                        ek_rmsd_from_S(S, Gf, gtile[c], A);
                }
            }
        }
        return;
    }
    float bestv = -__builtin_inff();
    uint32_t besti = 0xffffffffu;
    if (f < n) {
        const double Gf = G[f];
        // candidate 0: the new center of this iteration (kcenters.py:298-306)
        float S0[9];
#pragma unroll
        for (int j = 0; j < 9; ++j)
            S0[j] = s2[0][j][0];
        const float d0 = ek_rmsd_from_S(S0, Gf, gtile[0], A);
        float cur = dist[f];
        if (d0 < cur) {
            cur = d0;
            dist[f] = d0;
            assign[f] = label;
        }
        bestv = cur;
        besti = (uint32_t)f;
        // the guesses: keep their distances for later (one quartic solve at a
        // time: interleaving them costs registers)
#pragma unroll
        for (int c = 1; c < T; ++c) {
            __builtin_amdgcn_sched_barrier(0);
            if (c < teff) {
                float S[9];
#pragma unroll
                for (int j = 0; j < 9; ++j)
                    S[j] = s2[c / 2][j][c & 1];
                // only "< the frame's distance" will ever be asked of it, and
                // that distance never grows: stop the solve once that is settled
                vecs[(size_t)(c - 1) * n_pad + f] =
                    ek_rmsd_from_S_above(S, Gf, gtile[c], A, cur);
            }
        }
    }
    ek_wave_argmax(bestv, besti);
    const int lane = tid & (EK_WAVE - 1), wave = tid / EK_WAVE;
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

size_t ek_pass_lds_bytes(int T, int A) { return (size_t)3 * A * T * sizeof(float); }

void ek_launch_pass(int T, const float *tiles, const double *G, float *dist,
                    int32_t *assign, float *vecs, int64_t n, int64_t n_pad,
                    int A, const unsigned char *recs, const EkPlan *plan,
                    EkBlockMax *blockmax, hipStream_t s)
{
    if (n <= 0)
        return;
    const unsigned blocks = (unsigned)((n + EK_BLOCK - 1) / EK_BLOCK);
    const size_t lds = ek_pass_lds_bytes(T, A);
#define EK_PASS(TT)                                                            \
    do {                                                                       \
        if (lds > 48 * 1024)                                                   \
            (void)hipFuncSetAttribute(                                         \
                (const void *)ek_pass_kernel<TT, true>,                        \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL((ek_pass_kernel<TT, true>), dim3(blocks),           \
                           dim3(EK_BLOCK), lds, s, tiles, G, dist, assign,     \
                           vecs, n, n_pad, A, recs, plan, blockmax);           \
    } while (0)
    if (T == 8)
        EK_PASS(8);
    else
        EK_PASS(4);
#undef EK_PASS
}

// distances of every frame to `count` records (count <= 8), nothing else:
// vecs[j][f] = rmsd(frame f, record j).  Used to prefetch PAM proposals.
__global__ void ek_plan_fixed_kernel(EkPlan *__restrict__ plan, int count)
{
    if (threadIdx.x != 0)
        return;
    plan->go = 1;
    plan->teff = count;
    plan->label = 0;
    for (int j = 0; j < EK_MAX_CANDS; ++j)
        plan->src[j] = j;
}

void ek_launch_pass_dist(int count, const float *tiles, const double *G,
                         float *vecs, int64_t n, int64_t n_pad, int A,
                         const unsigned char *recs, EkPlan *plan, hipStream_t s)
{
    if (n <= 0 || count <= 0)
        return;
    hipLaunchKernelGGL(ek_plan_fixed_kernel, dim3(1), dim3(EK_WAVE), 0, s, plan,
                       count);
    const unsigned blocks = (unsigned)((n + EK_BLOCK - 1) / EK_BLOCK);
    const int T = (count <= 4) ? 4 : 8;
    const size_t lds = ek_pass_lds_bytes(T, A);
#define EK_PASSD(TT)                                                           \
    do {                                                                       \
        if (lds > 48 * 1024)                                                   \
            (void)hipFuncSetAttribute(                                         \
                (const void *)ek_pass_kernel<TT, false>,                       \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL((ek_pass_kernel<TT, false>), dim3(blocks),          \
                           dim3(EK_BLOCK), lds, s, tiles, G, nullptr, nullptr, \
                           vecs, n, n_pad, A, recs, plan, nullptr);            \
    } while (0)
    if (T == 8)
        EK_PASSD(8);
    else
        EK_PASSD(4);
#undef EK_PASSD
}

// ---------------------------------------------------------------------------
// per-workgroup maxima of the current distances (used to seed a run)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(EK_BLOCK)
ek_blockmax_kernel(const float *__restrict__ dist, int64_t n,
                   EkBlockMax *__restrict__ blockmax)
{
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];
    const int tid = threadIdx.x;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + tid;
    float v = -__builtin_inff();
    uint32_t i = 0xffffffffu;
    if (f < n) {
        v = dist[f];
        i = (uint32_t)f;
    }
    ek_wave_argmax(v, i);
    if ((tid & 63) == 0) {
        red_v[tid / 64] = v;
        red_i[tid / 64] = i;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
            if (ek_better(red_v[w], red_i[w], v, i)) {
                v = red_v[w];
                i = red_i[w];
            }
        blockmax[blockIdx.x].val = v;
        blockmax[blockIdx.x].idx = i;
    }
}

void ek_launch_blockmax(const float *dist, int64_t n, EkBlockMax *blockmax,
                        hipStream_t s)
{
    if (n <= 0)
        return;
    hipLaunchKernelGGL(ek_blockmax_kernel,
                       dim3((unsigned)((n + EK_BLOCK - 1) / EK_BLOCK)),
                       dim3(EK_BLOCK), 0, s, dist, n, blockmax);
}

// ---------------------------------------------------------------------------
// single-workgroup reductions over the per-workgroup maxima
// ---------------------------------------------------------------------------
#define EK_RED_THREADS 1024

// (max, first index) over blockmax[0..nb) skipping entries whose block is
// marked in `skip` (LDS bitmap, may be null) -> all threads get the result
__device__ __forceinline__ void ek_block_argmax(const EkBlockMax *blockmax,
                                                int nb, const uint32_t *skip,
                                                float &out_v, uint32_t &out_i,
                                                int &out_b)
{
    __shared__ float r_v[EK_RED_THREADS / EK_WAVE];
    __shared__ uint32_t r_i[EK_RED_THREADS / EK_WAVE];
    __shared__ int r_b[EK_RED_THREADS / EK_WAVE];
    __shared__ float w_v;
    __shared__ uint32_t w_i;
    __shared__ int w_b;
    const int tid = threadIdx.x;
    float v = -__builtin_inff();
    uint32_t i = 0xffffffffu;
    int bsel = -1;
    for (int b = tid; b < nb; b += EK_RED_THREADS) {
        if (skip && (skip[b >> 5] & (1u << (b & 31))))
            continue;
        const EkBlockMax m = blockmax[b];
        if (m.idx == 0xffffffffu)
            continue;
        if (ek_better(m.val, m.idx, v, i)) {
            v = m.val;
            i = m.idx;
            bsel = b;
        }
    }
    // wave reduce carrying the block id along
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const uint32_t oi = __shfl_xor(i, off, 64);
        const int ob = __shfl_xor(bsel, off, 64);
        if (ek_better(ov, oi, v, i)) {
            v = ov;
            i = oi;
            bsel = ob;
        }
    }
    if ((tid & 63) == 0) {
        r_v[tid / 64] = v;
        r_i[tid / 64] = i;
        r_b[tid / 64] = bsel;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < EK_RED_THREADS / EK_WAVE; ++w)
            if (ek_better(r_v[w], r_i[w], v, i)) {
                v = r_v[w];
                i = r_i[w];
                bsel = r_b[w];
            }
        w_v = v;
        w_i = i;
        w_b = bsel;
    }
    __syncthreads();
    out_v = w_v;
    out_i = w_i;
    out_b = w_b;
    __syncthreads();
}

// local farthest point -> 16-byte header (what a rank contributes to the
// per-center exchange)
__global__ void __launch_bounds__(EK_RED_THREADS)
ek_localmax_kernel(const EkBlockMax *__restrict__ blockmax, int nb,
                   int64_t global_offset, EkMaxHdr *__restrict__ out)
{
    float v;
    uint32_t i;
    int b;
    ek_block_argmax(blockmax, nb, nullptr, v, i, b);
    if (threadIdx.x == 0) {
        out->maxdist = v;
        out->valid = (i != 0xffffffffu) ? 1 : 0;
        out->gidx = (i != 0xffffffffu) ? global_offset + (int64_t)i : -1;
    }
}

void ek_launch_localmax(const EkBlockMax *blockmax, int nb,
                        int64_t global_offset, EkMaxHdr *out, hipStream_t s)
{
    hipLaunchKernelGGL(ek_localmax_kernel, dim3(1), dim3(EK_RED_THREADS), 0, s,
                       blockmax, nb, global_offset, out);
}

// decide whether the global farthest point is a stored candidate
__global__ void __launch_bounds__(EK_WAVE)
ek_check_kernel(const EkMaxHdr *__restrict__ hdrs, int n_hdrs, double cutoff,
                EkPlan *__restrict__ plan, EkHist *__restrict__ hist,
                EkCtl *__restrict__ ctl)
{
    if (threadIdx.x != 0)
        return;
    plan->apply = -1;
    if (!plan->go || plan->miss)
        return;
    plan->miss = 1;                      // until proven a hit
    if (ctl->stopped || ctl->n_done >= ctl->limit)
        return;
    int best = -1;
    float bv = 0.f;
    long long bg = 0;
    for (int r = 0; r < n_hdrs; ++r) {
        if (!hdrs[r].valid)
            continue;
        if (best < 0 || hdrs[r].maxdist > bv ||
            (hdrs[r].maxdist == bv && hdrs[r].gidx < bg)) {
            best = r;
            bv = hdrs[r].maxdist;
            bg = hdrs[r].gidx;
        }
    }
    if (best < 0)
        return;
    ctl->last_max = bv;
    if (!((double)bv > cutoff)) {        // kcenters.py:217
        ctl->stopped = 1;
        return;
    }
    for (int j = 1; j < plan->teff; ++j) {
        if (plan->gidx[j] == bg && !(plan->used & (1u << j))) {
            const int label = ctl->n_done;
            plan->apply = j;
            plan->apply_label = label;
            plan->used |= 1u << j;
            plan->miss = 0;
            hist[label].gidx = bg;
            hist[label].dist = bv;
            hist[label].set = 1;
            ctl->n_done = label + 1;
            return;
        }
    }
}

// single shard: local arg-max and the decision in one launch
__global__ void __launch_bounds__(EK_RED_THREADS)
ek_localmax_check_kernel(const EkBlockMax *__restrict__ blockmax, int nb,
                         int64_t global_offset, double cutoff,
                         EkPlan *__restrict__ plan, EkHist *__restrict__ hist,
                         EkCtl *__restrict__ ctl)
{
    // nothing to decide: the round did not run or already missed
    if (!plan->go || plan->miss) {
        if (threadIdx.x == 0)
            plan->apply = -1;
        return;
    }
    float v;
    uint32_t i;
    int b;
    ek_block_argmax(blockmax, nb, nullptr, v, i, b);
    if (threadIdx.x != 0)
        return;
    plan->apply = -1;
    plan->miss = 1;
    if (ctl->stopped || ctl->n_done >= ctl->limit || i == 0xffffffffu)
        return;
    ctl->last_max = v;
    if (!((double)v > cutoff)) {
        ctl->stopped = 1;
        return;
    }
    const long long bg = global_offset + (long long)i;
    for (int j = 1; j < plan->teff; ++j) {
        if (plan->gidx[j] == bg && !(plan->used & (1u << j))) {
            const int label = ctl->n_done;
            plan->apply = j;
            plan->apply_label = label;
            plan->used |= 1u << j;
            plan->miss = 0;
            hist[label].gidx = bg;
            hist[label].dist = v;
            hist[label].set = 1;
            ctl->n_done = label + 1;
            return;
        }
    }
}

void ek_launch_localmax_check(const EkBlockMax *blockmax, int nb,
                              int64_t global_offset, double cutoff,
                              EkPlan *plan, EkHist *hist, EkCtl *ctl,
                              hipStream_t s)
{
    hipLaunchKernelGGL(ek_localmax_check_kernel, dim3(1), dim3(EK_RED_THREADS),
                       0, s, blockmax, nb, global_offset, cutoff, plan, hist,
                       ctl);
}

void ek_launch_check(const EkMaxHdr *hdrs, int n_hdrs, double cutoff,
                     EkPlan *plan, EkHist *hist, EkCtl *ctl, hipStream_t s)
{
    hipLaunchKernelGGL(ek_check_kernel, dim3(1), dim3(EK_WAVE), 0, s, hdrs,
                       n_hdrs, cutoff, plan, hist, ctl);
}

// apply a stored distance vector (kcenters.py:304-306 on kept distances)
__global__ void __launch_bounds__(EK_BLOCK)
ek_apply_kernel(const float *__restrict__ vecs, const double *__restrict__ G,
                int64_t n, int64_t n_pad, int A, float *__restrict__ dist,
                int32_t *__restrict__ assign, const EkPlan *__restrict__ plan,
                EkBlockMax *__restrict__ blockmax)
{
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];
    const int j = plan->apply;
    if (j < 1)
        return;
    const int label = plan->apply_label;
    const int tid = threadIdx.x;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + tid;
    float v = -__builtin_inff();
    uint32_t i = 0xffffffffu;
    if (f < n) {
        const float d = vecs[(size_t)(j - 1) * n_pad + f];
        float cur = dist[f];
        if (d < cur) {
            cur = d;
            dist[f] = d;
            assign[f] = label;
        }
        v = cur;
        i = (uint32_t)f;
    }
    ek_wave_argmax(v, i);
    if ((tid & 63) == 0) {
        red_v[tid / 64] = v;
        red_i[tid / 64] = i;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
            if (ek_better(red_v[w], red_i[w], v, i)) {
                v = red_v[w];
                i = red_i[w];
            }
        blockmax[blockIdx.x].val = v;
        blockmax[blockIdx.x].idx = i;
    }
}

void ek_launch_apply(const float *vecs, const double *G, int64_t n,
                     int64_t n_pad, int A, float *dist, int32_t *assign,
                     const EkPlan *plan, EkBlockMax *blockmax, hipStream_t s)
{
    if (n <= 0)
        return;
    hipLaunchKernelGGL(ek_apply_kernel,
                       dim3((unsigned)((n + EK_BLOCK - 1) / EK_BLOCK)),
                       dim3(EK_BLOCK), 0, s, vecs, G, n, n_pad, A, dist, assign,
                       plan, blockmax);
}

// the shard's T candidate records for the next round: record 0 is its
// first-index arg-max, records 1.. the best frames of the next-best
// workgroups, preferring frames whose current label differs from the labels of
// the candidates already chosen (two far frames of one cluster tend to be
// near each other, and then only one of them can become a center).  Any far
// frames would do: correctness never depends on the guesses.
__global__ void __launch_bounds__(EK_RED_THREADS)
ek_pickT_kernel(const EkBlockMax *__restrict__ blockmax, int nb,
                const float *__restrict__ tiles, const double *__restrict__ G,
                const int32_t *__restrict__ assign, int A, int T,
                int64_t global_offset, unsigned char *__restrict__ recs,
                EkCtl *__restrict__ ctl)
{
    extern __shared__ uint32_t skip[];       // bitmap over workgroups
    __shared__ uint32_t sel_i[EK_MAX_CANDS];
    __shared__ float sel_v[EK_MAX_CANDS];
    __shared__ int32_t sel_lab[EK_MAX_CANDS];
    __shared__ int n_sel;
    __shared__ uint32_t top_i[EK_MAX_CANDS + 4];
    __shared__ float top_v[EK_MAX_CANDS + 4];
    __shared__ int32_t top_lab[EK_MAX_CANDS + 4];
    __shared__ int n_top;
    const int tid = threadIdx.x;
    for (int w = tid; w < (nb + 31) / 32; w += EK_RED_THREADS)
        skip[w] = 0;
    if (tid == 0) {
        n_sel = 0;
        n_top = 0;
    }
    __syncthreads();
    // at most T + 4 looks: a few candidates may be passed over for carrying a
    // label that is already represented, as long as T can still be filled.
    // The per-workgroup maxima are read once: every thread keeps its (up to
    // PICK_PER) entries in registers across the looks; larger shards fall back
    // to re-reading them.
    constexpr int PICK_PER = 8;
    const bool cached = nb <= PICK_PER * EK_RED_THREADS;
    float cv[PICK_PER];
    uint32_t ci[PICK_PER];
    if (cached) {
#pragma unroll
        for (int k = 0; k < PICK_PER; ++k) {
            const int bb = tid + k * EK_RED_THREADS;
            cv[k] = -__builtin_inff();
            ci[k] = 0xffffffffu;
            if (bb < nb) {
                const EkBlockMax m = blockmax[bb];
                cv[k] = m.val;
                ci[k] = m.idx;
            }
        }
    }
    constexpr int NWV = EK_RED_THREADS / EK_WAVE;
    constexpr int LMAX = EK_MAX_CANDS + 4;
    __shared__ float wt_v[NWV * LMAX];
    __shared__ uint32_t wt_i[NWV * LMAX];
    const int max_looks = T + 4;
    if (cached) {
        // Two levels, no workgroup barrier inside the looks: every wave takes the
        // top max_looks of its own entries with wave-wide arg-max steps, then
        // wave 0 takes the top max_looks of those NWV lists -- the same ordered
        // list as max_looks arg-max passes over all entries.
        const int lane = tid & (EK_WAVE - 1), wv = tid / EK_WAVE;
        for (int look = 0; look < max_looks; ++look) {
            float v = -__builtin_inff();
            uint32_t i = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < PICK_PER; ++k)
                if (ci[k] != 0xffffffffu && ek_better(cv[k], ci[k], v, i)) {
                    v = cv[k];
                    i = ci[k];
                }
            ek_wave_argmax(v, i);            // every lane holds the winner
            if (i != 0xffffffffu) {          // its owner retires it (indices are unique)
#pragma unroll
                for (int k = 0; k < PICK_PER; ++k)
                    if (ci[k] == i)
                        ci[k] = 0xffffffffu;
            }
            if (lane == 0) {
                wt_v[wv * LMAX + look] = v;
                wt_i[wv * LMAX + look] = i;
            }
        }
        __syncthreads();
        if (wv == 0) {
            constexpr int PER2 = (NWV * LMAX + EK_WAVE - 1) / EK_WAVE;
            float ev[PER2];
            uint32_t ei[PER2];
#pragma unroll
            for (int k = 0; k < PER2; ++k) {
                const int e = lane + k * EK_WAVE;
                const int w = e / LMAX, l = e % LMAX;
                const bool ok = w < NWV && l < max_looks;
                ev[k] = ok ? wt_v[w * LMAX + l] : -__builtin_inff();
                ei[k] = ok ? wt_i[w * LMAX + l] : 0xffffffffu;
            }
            int cnt = 0;
            for (int look = 0; look < max_looks; ++look) {
                float v = -__builtin_inff();
                uint32_t i = 0xffffffffu;
#pragma unroll
                for (int k = 0; k < PER2; ++k)
                    if (ei[k] != 0xffffffffu && ek_better(ev[k], ei[k], v, i)) {
                        v = ev[k];
                        i = ei[k];
                    }
                ek_wave_argmax(v, i);
                if (i == 0xffffffffu)
                    break;
#pragma unroll
                for (int k = 0; k < PER2; ++k)
                    if (ei[k] == i)
                        ei[k] = 0xffffffffu;
                if (lane == 0) {
                    top_i[cnt] = i;
                    top_v[cnt] = v;
                }
                ++cnt;
            }
            if (lane == 0)
                n_top = cnt;
        }
    } else {
        for (int look = 0; look < max_looks; ++look) {
            float v;
            uint32_t i;
            int b;
            ek_block_argmax(blockmax, nb, skip, v, i, b);
            if (b < 0)
                break;
            if (tid == 0) {
                skip[b >> 5] |= 1u << (b & 31);
                top_i[n_top] = i;
                top_v[n_top] = v;
                n_top = n_top + 1;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // labels of the looked-at frames, fetched in parallel
    if (tid < n_top)
        top_lab[tid] = assign[top_i[tid]];
    __syncthreads();
    if (tid == 0) {
        const int nt = n_top;
        for (int look = 0; look < nt && n_sel < T; ++look) {
            const int32_t lab = top_lab[look];
            bool dup = false;
            const bool can_skip = (nt - look - 1) >= (T - n_sel);
            if (can_skip && lab >= 0)
                for (int j = 0; j < n_sel; ++j)
                    dup = dup || (sel_lab[j] == lab);
            if (!dup || n_sel == 0) {
                sel_i[n_sel] = top_i[look];
                sel_v[n_sel] = top_v[look];
                sel_lab[n_sel] = lab;
                n_sel = n_sel + 1;
            }
        }
    }
    __syncthreads();
    const int ns = n_sel;
    const size_t rstride = ek_rec_bytes(A);
    if (tid < T) {
        EkRecHdr *h = (EkRecHdr *)(recs + (size_t)tid * rstride);
        if (tid < ns) {
            h->maxdist = sel_v[tid];
            h->valid = 1;
            h->gidx = global_offset + (int64_t)sel_i[tid];
            h->trace = G[sel_i[tid]];
            h->reserved = 0;
        } else {
            h->maxdist = -__builtin_inff();
            h->valid = 0;
            h->gidx = -1;
            h->trace = 0.0;
            h->reserved = 0;
        }
    }
    if (tid == 0)
        ctl->last_max = (ns > 0) ? sel_v[0] : -__builtin_inff();
}

// the candidates' coordinates: every read is its own cache line, so they are
// spread over many workgroups (one thread per value) instead of being queued
// behind the selection in its single workgroup
__global__ void __launch_bounds__(EK_BLOCK)
ek_record_gather_kernel(const float *__restrict__ tiles, int A,
                        int64_t global_offset, unsigned char *__restrict__ recs)
{
    const size_t rstride = ek_rec_bytes(A);
    const int j = blockIdx.y;
    const EkRecHdr *h = (const EkRecHdr *)(recs + (size_t)j * rstride);
    if (!h->valid)
        return;
    const int r = blockIdx.x * EK_BLOCK + threadIdx.x;
    if (r >= 3 * A)
        return;
    const int64_t i = h->gidx - global_offset;
    const float *p = tiles + (size_t)(i / EK_TILE) * 3 * (size_t)A * EK_TILE +
                     (i % EK_TILE);
    float *coords = (float *)(recs + (size_t)j * rstride + sizeof(EkRecHdr));
    coords[r] = p[(size_t)r * EK_TILE];
}

void ek_launch_pickT(const EkBlockMax *blockmax, int nb, const float *tiles,
                     const double *G, const int32_t *assign, int A, int T,
                     int64_t global_offset, unsigned char *recs, EkCtl *ctl,
                     hipStream_t s)
{
    const size_t lds = (size_t)((nb + 31) / 32 + 1) * sizeof(uint32_t);
    hipLaunchKernelGGL(ek_pickT_kernel, dim3(1), dim3(EK_RED_THREADS), lds, s,
                       blockmax, nb, tiles, G, assign, A, T, global_offset, recs,
                       ctl);
    hipLaunchKernelGGL(ek_record_gather_kernel,
                       dim3((unsigned)((3 * A + EK_BLOCK - 1) / EK_BLOCK),
                            (unsigned)T),
                       dim3(EK_BLOCK), 0, s, tiles, A, global_offset, recs);
}
