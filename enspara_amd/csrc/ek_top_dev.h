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
// the list the fused single-shard round chooses its candidates from (ek_round.hip):
// EK_TOP_M, or 128 in measurement builds (-DEK_LIST_M=128: the greedy choice then
// runs over 128 entries with 8128 pairwise distances; the one-launch-per-step forms
// and the rounds across shards keep 64)
#ifndef EK_LIST_M
#define EK_LIST_M 64
#endif
static_assert(EK_LIST_M == 64 || EK_LIST_M == 128, "one or two list entries per lane");
#define EK_TOP_HEAD 4096    // bytes of the scratch area before the coordinates

struct EkTop {
    int32_t n;
    int32_t pad;
    uint32_t idx[EK_LIST_M];
    float val[EK_LIST_M];
};
static_assert(sizeof(EkTop) <= 2048, "EkMsPub sits at + 2048");

__device__ __forceinline__ float *ek_top_coords(unsigned char *scr)
{
    return (float *)(scr + EK_TOP_HEAD);
}
__device__ __forceinline__ double *ek_top_traces(unsigned char *scr, int A)
{
    return (double *)(scr + EK_TOP_HEAD + (size_t)EK_TOP_M * 3 * A * sizeof(float));
}
__device__ __forceinline__ float *ek_top_D(unsigned char *scr, int A)
{
    return (float *)(scr + EK_TOP_HEAD + (size_t)EK_TOP_M * 3 * A * sizeof(float) +
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
// The list only steers the round's guesses, but a guess counts only if it is
// exactly the farthest frame of the state at its turn, so what the list must hold
// is the top frame of as many different regions as it can.  Taking the 64 largest
// maxima does not do that: a region far from every center floods the ranking
// with its own frames (measured on the bench data: the 64 entries carried 24
// different labels on average, fewer than 10 in one round of nine -- the rounds
// that ran out of candidates).  Frames of one region share their nearest center
// (but several regions may share one), so only the EK_PICK_PER_LABEL (4) largest
// maxima of every current label compete at all: LDS atomic max on a table row per
// label, one round per place, with the value's leading bits and the entry's
// number as the key.  Of the survivors a pool of about EK_PICK_POOL is cut off by
// value (a histogram of their distance below the maximum: no sequential looks)
// and ranked exactly; the first EK_TOP_M make the list.  (Two per label is too
// strict -- more passes than without any of this --, four or eight do equally
// well; giving every label's first a head start changes nothing.)  Entry 0 is
// always the overall first-index arg-max: it holds slot 0 of the pool by
// construction.
#ifndef EK_PICK_POOL
#define EK_PICK_POOL (2 * EK_LIST_M)
#endif
#ifndef EK_PICK_PER_LABEL
#define EK_PICK_PER_LABEL 4     // maxima kept per label
#endif
#define EK_PICK_BUCKETS 1024
#ifndef EK_PICK_SLOTS
#define EK_PICK_SLOTS 2048      // labels modulo this share a table row
#endif

template <bool COH = false>
__device__ __forceinline__ void ek_pick_top_body(const EkBlockMax *blockmax, int nb,
                                                 EkTop *top, uint32_t *skip,
                                                 const int32_t *assign = nullptr,
                                                 int cap = EK_PICK_PER_LABEL,
                                                 int list_m = EK_TOP_M)
{
    // `cap`: maxima kept per label, 1 .. 16 (round 5: a run-time choice.  Four is
    // right where frames come in clouds around templates -- more floods the list
    // with one cloud's frames --, sixteen where they lie on a continuous landscape
    // and a label's region holds many far directions: bench.py --data walk 1.3e10 ->
    // 2.5e10 pairs/s; the drivers choose by the yield they see, ek_run_rounds)
    cap = cap < 1 ? 1 : (cap > 16 ? 16 : cap);
    const unsigned int slots = (unsigned int)(EK_PICK_PER_LABEL * EK_PICK_SLOTS) / (unsigned int)cap;
    // (`skip` is not used any more: round 5 keeps a thread's PICK_PER largest
    // entries whatever the number of entries, where a list of more than 8192 used
    // to fall back to 64 sequential looks)
    EK_PSTAMP(0);
    __shared__ uint32_t top_i[EK_LIST_M];
    __shared__ float top_v[EK_LIST_M];
    __shared__ int n_top;
    const int tid = threadIdx.x;
    constexpr int PICK_PER = 8;
    if (tid == 0)
        n_top = 0;
    __syncthreads();
    {
        static_assert(EK_RED_THREADS % EK_PICK_POOL == 0 &&
                          EK_PICK_BUCKETS == EK_RED_THREADS &&
                          PICK_PER * EK_RED_THREADS <= (1 << 13),
                      "pool comparisons and buckets are split over the workgroup; "
                      "13 bits number an entry");
        __shared__ unsigned int s_maxbits, s_thresh, s_nsel, s_first, s_have;
        __shared__ unsigned int hist[EK_PICK_BUCKETS];
        __shared__ unsigned int tab[EK_PICK_PER_LABEL * EK_PICK_SLOTS];
        __shared__ float sel_v[EK_PICK_POOL];
        __shared__ uint32_t sel_i[EK_PICK_POOL];
        __shared__ int sel_rank[EK_PICK_POOL];
        // The maxima are read once, PICK_PER per thread, and with them the labels
        // of those frames.  A list longer than PICK_PER x 1024 entries (maxima per
        // 64 frames of a million-frame shard: 15 625): a thread keeps the PICK_PER
        // LARGEST of its strided share -- the pool below holds ~128 entries of the
        // whole list, and more than eight of them in one thread's share of sixteen
        // do not happen; if they did it would cost a guess, never a result.
        // entries per thread actually in use (uniform: the loops skip the rest)
        const int per = nb >= PICK_PER * EK_RED_THREADS
                            ? PICK_PER : (nb + EK_RED_THREADS - 1) / EK_RED_THREADS;
        float cv[PICK_PER];
        uint32_t ci[PICK_PER];
        int32_t lab[PICK_PER];
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
        for (int b0 = tid + PICK_PER * EK_RED_THREADS; b0 < nb;
             b0 += PICK_PER * EK_RED_THREADS) {
            EkBlockMax m[PICK_PER];     // (a batch of loads in flight)
#pragma unroll
            for (int u = 0; u < PICK_PER; ++u) {
                const int bb = b0 + u * EK_RED_THREADS;
                m[u] = ek_ld_bm<COH>(&blockmax[bb < nb ? bb : nb - 1]);
                if (bb >= nb)
                    m[u].idx = 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < PICK_PER; ++u) {
                if (m[u].idx == 0xffffffffu)
                    continue;
                // the slot holding the smallest value (an empty one first)
                int lo = 0;
                float lv = ci[0] == 0xffffffffu ? -__builtin_inff() : cv[0];
#pragma unroll
                for (int k = 1; k < PICK_PER; ++k) {
                    const float kv = ci[k] == 0xffffffffu ? -__builtin_inff() : cv[k];
                    if (kv < lv) {
                        lv = kv;
                        lo = k;
                    }
                }
                if (m[u].val > lv) {
#pragma unroll
                    for (int k = 0; k < PICK_PER; ++k)
                        if (k == lo) {
                            cv[k] = m[u].val;
                            ci[k] = m[u].idx;
                        }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PICK_PER && k < per; ++k)
            lab[k] = (ci[k] != 0xffffffffu && assign) ? assign[ci[k]]
                                                      : tid + k * EK_RED_THREADS;
        hist[tid] = 0;
        for (int q = tid; q < EK_PICK_PER_LABEL * EK_PICK_SLOTS; q += EK_RED_THREADS)
            tab[q] = 0;
        if (tid < EK_PICK_POOL)
            sel_rank[tid] = 0;
        if (tid == 0) {
            s_maxbits = 0;
            s_nsel = 0;
            s_first = 0xffffffffu;
            s_have = 0;
        }
        __syncthreads();
        EK_PSTAMP(1);
        // the EK_PICK_PER_LABEL best of every label, round by round: who is not
        // the best of a round competes in the next (distances are >= 0: their
        // bits order like they do; + 1 keeps a key apart from an empty slot)
        unsigned int key[PICK_PER];
        int won[PICK_PER];              // the round an entry won, -1: none
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < PICK_PER; ++k) {
            key[k] = 0;
            won[k] = -1;
            if (ci[k] != 0xffffffffu) {
                const unsigned int e = (unsigned int)(tid + k * EK_RED_THREADS);
                key[k] = (((__float_as_uint(cv[k]) >> 12) + 1u) << 13) | (8191u - e);
                mine = cv[k] > mine ? cv[k] : mine;
            }
        }
        if (mine > 0.f)
            atomicMax(&s_maxbits, __float_as_uint(mine));
        for (int r = 0; r < cap; ++r) {
#pragma unroll
            for (int k = 0; k < PICK_PER && k < per; ++k)
                if (key[k] && won[k] < 0)
                    atomicMax(&tab[(unsigned int)cap * ((unsigned int)lab[k] % slots) + r],
                              key[k]);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PICK_PER && k < per; ++k)
                if (key[k] && won[k] < 0 &&
                    tab[(unsigned int)cap * ((unsigned int)lab[k] % slots) + r] == key[k])
                    won[k] = r;
        }
        // the overall first-index arg-max, exactly: the keys round the values,
        // and entry 0 of the list is a center without further checks
        const float vmax = __uint_as_float(s_maxbits);
#pragma unroll
        for (int k = 0; k < PICK_PER && k < per; ++k)
            if (key[k] && cv[k] == vmax)
                atomicMin(&s_first, ci[k]);
        __syncthreads();
        // the arg-max takes slot 0 of the pool, whatever else happens
#pragma unroll
        for (int k = 0; k < PICK_PER && k < per; ++k)
            if (key[k] && cv[k] == vmax && ci[k] == s_first) {
                won[k] = -1;
                sel_v[0] = cv[k];
                sel_i[0] = ci[k];
                s_have = 1;
            }
        // bucket = how far below the maximum, 1024 steps over its top quarter
        const float scale = vmax > 0.f ? (float)EK_PICK_BUCKETS / (0.25f * vmax) : 0.f;
        int bk[PICK_PER];
#pragma unroll
        for (int k = 0; k < PICK_PER; ++k) {
            bk[k] = -1;
            if (won[k] >= 0) {
                const float below = (vmax - cv[k]) * scale;
                bk[k] = below >= (float)(EK_PICK_BUCKETS - 1) ? EK_PICK_BUCKETS - 1
                                                               : (below > 0.f ? (int)below : 0);
                atomicAdd(&hist[bk[k]], 1u);
            }
        }
        __syncthreads();
        // the first bucket at which the pool is full (one wave, 16 buckets a lane)
        if (tid < EK_WAVE) {
            constexpr int PB = EK_PICK_BUCKETS / EK_WAVE;
            unsigned int loc[PB], sum = 0;
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                sum += hist[tid * PB + q];
                loc[q] = sum;
            }
            unsigned int incl = sum;
#pragma unroll
            for (int o = 1; o < EK_WAVE; o <<= 1) {
                const unsigned int v = __shfl_up(incl, o, EK_WAVE);
                if (tid >= o)
                    incl += v;
            }
            const unsigned int excl = incl - sum;
            int first = EK_PICK_BUCKETS;        // none: everything goes in
#pragma unroll
            for (int q = PB - 1; q >= 0; --q)
                if (excl + loc[q] >= (unsigned int)(EK_PICK_POOL - 1))
                    first = tid * PB + q;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {     // the lowest over the lanes
                const int v = __shfl_xor(first, o, EK_WAVE);
                first = v < first ? v : first;
            }
            if (tid == 0)
                s_thresh = (unsigned int)first;
        }
        __syncthreads();
        // everything above the threshold bucket first (fewer than the pool), then
        // that bucket's entries while there is room
        const int thresh = (int)s_thresh;
#pragma unroll
        for (int k = 0; k < PICK_PER && k < per; ++k)
            if (bk[k] >= 0 && bk[k] < thresh) {
                const unsigned int p = 1u + atomicAdd(&s_nsel, 1u);
                sel_v[p] = cv[k];
                sel_i[p] = ci[k];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PICK_PER && k < per; ++k)
            if (bk[k] >= 0 && bk[k] == thresh) {
                const unsigned int p = 1u + atomicAdd(&s_nsel, 1u);
                if (p < (unsigned int)EK_PICK_POOL) {
                    sel_v[p] = cv[k];
                    sel_i[p] = ci[k];
                }
            }
        __syncthreads();
        EK_PSTAMP(2);
        // exact rank in the pool; the comparisons of an entry are split over
        // EK_RED_THREADS / POOL threads
        const int L = !s_have ? 0
                              : (s_nsel + 1u < (unsigned int)EK_PICK_POOL
                                     ? (int)s_nsel + 1 : EK_PICK_POOL);
        constexpr int SPLIT = EK_RED_THREADS / EK_PICK_POOL;
        constexpr int SHARE = EK_PICK_POOL / SPLIT;
        const int e = tid % EK_PICK_POOL, part = tid / EK_PICK_POOL;
        if (e < L) {
            const float v = sel_v[e];
            const uint32_t i = sel_i[e];
            int rank = 0;
#pragma unroll
            for (int q = 0; q < SHARE; ++q) {
                const int o = part * SHARE + q;
                if (o < L && ek_better(sel_v[o], sel_i[o], v, i))
                    ++rank;
            }
            if (rank)
                atomicAdd(&sel_rank[e], rank);
        }
        __syncthreads();
        if (tid < L && sel_rank[tid] < list_m) {
            top_i[sel_rank[tid]] = sel_i[tid];
            top_v[sel_rank[tid]] = sel_v[tid];
        }
        if (tid == 0)
            n_top = L < list_m ? L : list_m;
        __syncthreads();
        EK_PSTAMP(3);
    }
    __syncthreads();
    if (tid < EK_LIST_M) {
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

