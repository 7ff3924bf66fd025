// ek_top_dev.h -- device-side pieces of the candidate pick (ek_spec.hip) shared
// with the fused single-shard round (ek_round.hip): the farthest per-workgroup
// maxima, and the greedy choice of the next round's candidates among them.
#pragma once
#include "ek_common.h"
#include "ek_reduce.h"

#define EK_RED_THREADS 1024

// (max, first index) over blockmax[0..nb) skipping entries whose block is
// marked in `skip` (LDS bitmap, may be null) -> all threads get the result
template <bool COH = false>
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
        const EkBlockMax m = ek_ld_bm<COH>(&blockmax[b]);
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

#define EK_TOP_M 64

struct EkTop {
    int32_t n;
    int32_t pad;
    uint32_t idx[EK_TOP_M];
    float val[EK_TOP_M];
};

__device__ __forceinline__ float *ek_top_coords(unsigned char *scr)
{
    return (float *)(scr + 1024);
}
__device__ __forceinline__ double *ek_top_traces(unsigned char *scr, int A)
{
    return (double *)(scr + 1024 + (size_t)EK_TOP_M * 3 * A * sizeof(float));
}
__device__ __forceinline__ float *ek_top_D(unsigned char *scr, int A)
{
    return (float *)(scr + 1024 + (size_t)EK_TOP_M * 3 * A * sizeof(float) +
                     EK_TOP_M * sizeof(double));
}

// the EK_TOP_M largest per-workgroup maxima, ordered (value desc, index asc)
// (called by all EK_RED_THREADS threads of a workgroup; `skip`: LDS bitmap over
// the workgroups, (nb + 31) / 32 words, used only when nb > 8 * EK_RED_THREADS)
#ifdef EK_ROUND_STAMPS
__device__ unsigned long long ek_pick_st[8];
#define EK_PSTAMP(k) if (threadIdx.x == 0) ek_pick_st[k] = __builtin_amdgcn_s_memrealtime()
#else
#define EK_PSTAMP(k)
#endif
template <bool COH = false>
__device__ __forceinline__ void ek_pick_top_body(const EkBlockMax *blockmax, int nb,
                                                 EkTop *top, uint32_t *skip)
{
    EK_PSTAMP(0);
    __shared__ uint32_t top_i[EK_TOP_M];
    __shared__ float top_v[EK_TOP_M];
    __shared__ int n_top;
    const int tid = threadIdx.x;
    constexpr int PICK_PER = 8;
    const bool cached = nb <= PICK_PER * EK_RED_THREADS;
    if (!cached)
        for (int w = tid; w < (nb + 31) / 32; w += EK_RED_THREADS)
            skip[w] = 0;
    if (tid == 0)
        n_top = 0;
    __syncthreads();
    // The per-workgroup maxima are read once: every thread keeps its (up to
    // PICK_PER) entries in registers across the looks; larger shards fall back
    // to re-reading them.
    float cv[PICK_PER];
    uint32_t ci[PICK_PER];
    if (cached) {
#pragma unroll
        for (int k = 0; k < PICK_PER; ++k) {
            const int bb = tid + k * EK_RED_THREADS;
            cv[k] = -__builtin_inff();
            ci[k] = 0xffffffffu;
            if (bb < nb) {
                const EkBlockMax m = ek_ld_bm<COH>(&blockmax[bb]);
                cv[k] = m.val;
                ci[k] = m.idx;
            }
        }
    }
    EK_PSTAMP(1);
    constexpr int NWV = EK_RED_THREADS / EK_WAVE;
    constexpr int LW = 8;                // looks per wave
    __shared__ float wt_v[NWV * LW];
    __shared__ uint32_t wt_i[NWV * LW];
    const int max_looks = EK_TOP_M;
    if (cached) {
        // Level 1: every wave takes the LW best of its own entries with wave-wide
        // arg-max steps (no workgroup barrier).  Level 2: the NWV * LW survivors
        // are ranked against one another, one thread each, and the best EK_TOP_M
        // land in order.  The workgroups' maxima are spread over the waves at
        // random, so this is the true top list except when more than LW of it
        // fall into one wave -- which only costs a slightly worse guess; entry 0
        // is always the overall first-index arg-max.
        const int lane = tid & (EK_WAVE - 1), wv = tid / EK_WAVE;
        // entries per thread actually in use (uniform: the rest is skipped)
        const int per = (nb + EK_RED_THREADS - 1) / EK_RED_THREADS;
        for (int look = 0; look < LW; ++look) {
            float v = -__builtin_inff();
            uint32_t i = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < PICK_PER; ++k) {
                if (k >= per)
                    break;
                if (ci[k] != 0xffffffffu && ek_better(cv[k], ci[k], v, i)) {
                    v = cv[k];
                    i = ci[k];
                }
            }
            ek_wave_argmax(v, i);            // every lane holds the winner
            if (i != 0xffffffffu) {          // its owner retires it (indices are unique)
#pragma unroll
                for (int k = 0; k < PICK_PER; ++k) {
                    if (k >= per)
                        break;
                    if (ci[k] == i)
                        ci[k] = 0xffffffffu;
                }
            }
            if (lane == 0) {
                wt_v[wv * LW + look] = v;
                wt_i[wv * LW + look] = i;
            }
        }
        __syncthreads();
        EK_PSTAMP(2);
        // every survivor's rank among the NWV * LW of them, the comparisons
        // spread over the whole workgroup (EK_RED_THREADS / (NWV * LW) threads
        // per survivor, partial counts added up in LDS)
        {
            constexpr int NS = NWV * LW;                // survivors
            constexpr int SPLIT = EK_RED_THREADS / NS;  // threads per survivor
            static_assert(EK_RED_THREADS % NS == 0 && NS % SPLIT == 0, "even split");
            __shared__ int rank_acc[NS];
            if (tid < NS)
                rank_acc[tid] = 0;
            __syncthreads();
            const int e0 = tid % NS, part = tid / NS;
            const float v = wt_v[e0];
            const uint32_t i = wt_i[e0];
            if (i != 0xffffffffu) {
                int rank = 0;
#pragma unroll
                for (int q = 0; q < NS / SPLIT; ++q) {
                    const int e = part * (NS / SPLIT) + q;
                    const uint32_t oi = wt_i[e];
                    if (oi != 0xffffffffu && ek_better(wt_v[e], oi, v, i))
                        ++rank;
                }
                if (rank)
                    atomicAdd(&rank_acc[e0], rank);
            }
            __syncthreads();
            if (tid < NS && i != 0xffffffffu) {
                const int rank = rank_acc[tid];
                if (rank < EK_TOP_M) {
                    top_i[rank] = i;
                    top_v[rank] = v;
                }
                atomicAdd(&n_top, 1);
            }
        }
        __syncthreads();
        EK_PSTAMP(3);
        if (tid == 0 && n_top > EK_TOP_M)
            n_top = EK_TOP_M;
    } else {
        for (int look = 0; look < max_looks; ++look) {
            float v;
            uint32_t i;
            int b;
            ek_block_argmax<COH>(blockmax, nb, skip, v, i, b);
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
    if (tid < EK_TOP_M) {
        top->idx[tid] = (tid < n_top) ? top_i[tid] : 0xffffffffu;
        top->val[tid] = (tid < n_top) ? top_v[tid] : -__builtin_inff();
    }
    if (tid == 0)
        top->n = n_top;
}

// the greedy order, and the records of its first T frames
// (called by all threads of a workgroup of >= EK_WAVE threads)
template <bool COH = false>
__device__ __forceinline__ void ek_top_records_body(int A, int T,
                                                    int64_t global_offset,
                                                    unsigned char *scr,
                                                    unsigned char *recs,
                                                    EkCtl *ctl,
                                                    const float *aos = nullptr,
                                                    const double *G = nullptr)
{
    // aos / G given: the frames' coordinates and traces are read from the
    // frame-major copy of the shard, not from the scratch area
    const int nthr = blockDim.x;
    __shared__ int sel[EK_MAX_CANDS];
    __shared__ int n_sel;
    __shared__ float sD[EK_TOP_M * EK_TOP_M];
    __shared__ float sval[EK_TOP_M];
    __shared__ uint32_t sidx[EK_TOP_M];
    const EkTop *top = (const EkTop *)scr;
    const int tid = threadIdx.x;
    {   // the table into LDS first: the greedy loop is one thread's dependent reads
        const float *D = ek_top_D(scr, A);
        for (int k = tid; k < EK_TOP_M * EK_TOP_M; k += nthr)
            sD[k] = COH ? ek_coh_load(&D[k]) : D[k];
        if (tid < EK_TOP_M) {
            sval[tid] = top->val[tid];
            sidx[tid] = top->idx[tid];
        }
    }
    __syncthreads();
    if (tid < EK_WAVE) {
        // the greedy order, one wave: lane l holds entry l's remaining distance
        static_assert(EK_TOP_M <= EK_WAVE, "one lane per entry");
        const int nt = top->n;
        const int lane = tid;
        bool open = lane < nt;
        float cur = open ? sval[lane] : 0.f;
        const uint32_t ix = (lane < EK_TOP_M) ? sidx[lane] : 0xffffffffu;
        int ns = 0;
        while (ns < T) {
            float v = open ? cur : -__builtin_inff();
            uint32_t i = open ? ix : 0xffffffffu;
            ek_wave_argmax(v, i);
            if (i == 0xffffffffu)
                break;
            const unsigned long long who = __ballot(open && ix == i);
            const int best = __ffsll((long long)who) - 1;
            if (lane == best)
                open = false;
            if (lane == 0)
                sel[ns] = best;
            ++ns;
            if (lane < EK_TOP_M) {
                const float d = sD[best * EK_TOP_M + lane];
                if (open && d < cur)
                    cur = d;
            }
        }
        if (lane == 0) {
            n_sel = ns;
            ctl->last_max = (ns > 0) ? sval[sel[0]] : -__builtin_inff();
        }
    }
    __syncthreads();
    const int ns = n_sel;
    const size_t rstride = ek_rec_bytes(A);
    if (tid < T) {
        EkRecHdr *h = (EkRecHdr *)(recs + (size_t)tid * rstride);
        if (tid < ns) {
            const int s = sel[tid];
            h->maxdist = top->val[s];
            h->valid = 1;
            h->gidx = global_offset + (int64_t)top->idx[s];
            h->trace = aos ? G[top->idx[s]] : ek_top_traces(scr, A)[s];
            h->reserved = 0;
        } else {
            h->maxdist = -__builtin_inff();
            h->valid = 0;
            h->gidx = -1;
            h->trace = 0.0;
            h->reserved = 0;
        }
    }
    const float *tc = ek_top_coords(scr);
    for (int k = tid; k < ns * 3 * A; k += nthr) {
        const int j = k / (3 * A), r = k % (3 * A);
        float *coords = (float *)(recs + (size_t)j * rstride + sizeof(EkRecHdr));
        coords[r] = aos ? aos[(size_t)top->idx[sel[j]] * 3 * A + r]
                        : tc[(size_t)sel[j] * 3 * A + r];
    }
}

