// ek_chain_dev.h -- device-side pieces of the chained cheap steps shared by the
// per-step kernels (ek_chain.hip: the multi-shard protocol) and the fused
// single-shard round (ek_round.hip).  See ek_chain.hip for what they decide.
#pragma once
#include "ek_common.h"
#include "ek_reduce.h"

#define EK_CHAIN_THREADS 1024

// the presumed order from the candidate rows (one thread); returns its length
__device__ __forceinline__ int ek_chain_simulate(const EkPlan *plan,
                                                 const EkChainRow *rows,
                                                 int *chain_out)
{
    const int teff = plan->teff;
    float cur[EK_MAX_CANDS];
    bool open[EK_MAX_CANDS];
    for (int j = 0; j < EK_MAX_CANDS; ++j) {
        open[j] = j >= 1 && j < teff && rows[j].valid;
        cur[j] = open[j] ? rows[j].cur : 0.f;
    }
    int cn = 0;
    for (;;) {
        int best = -1;
        float bestv = 0.f;
        long long bestg = 0;
        for (int j = 1; j < teff; ++j) {
            if (!open[j])
                continue;
            const float vj = cur[j];
            const long long gj = plan->gidx[j];
            if (best < 0 || vj > bestv || (vj == bestv && gj < bestg)) {
                best = j;
                bestv = vj;
                bestg = gj;
            }
        }
        if (best < 0)
            break;
        open[best] = false;
        chain_out[cn++] = best;
        for (int j = 1; j < teff; ++j) {        // kcenters.py:304: strict <
            const float dj = rows[j].d[best];
            if (open[j] && dj < cur[j])
                cur[j] = dj;
        }
    }
    return cn;
}

// The same by one wave, lane j = candidate j (all 64 lanes of a wave call it;
// rows, chain_out, cn_out in LDS): T - 1 steps of a wave arg-max instead of
// (T - 1)^2 dependent scalar steps.  Same order as the loop above: largest
// current distance, lowest global index on ties (the candidates' ranks by
// global index stand in for the 64-bit indices).
__device__ __forceinline__ void ek_chain_simulate_wave(const EkPlan *plan,
                                                       const EkChainRow *rows,
                                                       int *chain_out, int *cn_out)
{
    const int lane = threadIdx.x & (EK_WAVE - 1);
    const int teff = plan->teff;
    const bool cand = lane >= 1 && lane < teff && lane < EK_MAX_CANDS;
    bool open = cand && rows[cand ? lane : 0].valid;
    float cur = open ? rows[lane].cur : 0.f;
    const long long g = cand ? plan->gidx[lane] : 0;
    uint32_t rank = 0;
    for (int j = 1; j < teff && j < EK_MAX_CANDS; ++j)
        if (plan->gidx[j] < g)
            ++rank;
    int cn = 0;
    for (;;) {
        float v = open ? cur : -__builtin_inff();
        uint32_t i = open ? rank : 0xffffffffu;
        ek_wave_argmax(v, i);
        if (i == 0xffffffffu)
            break;
        const unsigned long long who = __ballot(open && rank == i);
        const int best = __ffsll((long long)who) - 1;
        if (lane == best)
            open = false;
        if (lane == 0)
            chain_out[cn] = best;
        ++cn;
        if (cand) {                             // kcenters.py:304: strict <
            const float d = rows[lane].d[best];
            if (open && d < cur)
                cur = d;
        }
    }
    if (lane == 0)
        *cn_out = cn;
}

// this shard's rows into LDS (all threads of the workgroup call it)
__device__ __forceinline__ void ek_chain_rows_local(const EkPlan *plan,
                                                    const float *dist,
                                                    const float *vecs, int64_t n,
                                                    int64_t n_pad,
                                                    int64_t global_offset,
                                                    EkChainRow *rows, int tid)
{
  for (int e = tid; e < EK_MAX_CANDS * EK_MAX_CANDS; e += blockDim.x) {
    const int j = e / EK_MAX_CANDS, u = e % EK_MAX_CANDS;
    const bool live = plan->go && j >= 1 && j < plan->teff;
    const int64_t local = live ? plan->gidx[j] - global_offset : -1;
    const bool mine = live && local >= 0 && local < n;
    float val = 0.f;
    if (mine) {
        if (u == 0)
            val = dist[local];
        else if (u < plan->teff)
            val = vecs[(size_t)(u - 1) * n_pad + local];
    }
    if (u == 0) {
        rows[j].cur = val;
        rows[j].valid = mine ? 1 : 0;
        rows[j].d[0] = 0.f;
    } else {
        rows[j].d[u] = val;
    }
  }
}

// the workgroup's 16 waves reduce the per-workgroup maxima of states 0 .. cn - 1:
// two waves per state while there are at most 8 states, one each up to 16, and
// beyond (rounds of 32: cn <= EK_MAX_CANDS - 1 = 31) a wave takes states w and
// w + 16 one after the other; the loads of a trip are issued together (the
// entries are independent).  Called by 1024 threads.
template <bool COH = false>
__device__ __forceinline__ void ek_chain_reduce(const EkBlockMax *blockmax,
                                                const EkBlockMax *pm, int nb,
                                                int nbp, int cn, float *out_v,
                                                uint32_t *out_i)
{
    __shared__ float half_v[2 * EK_MAX_CANDS];
    __shared__ uint32_t half_i[2 * EK_MAX_CANDS];
    const int tid = threadIdx.x;
    const int wv = tid / EK_WAVE, lane = tid & (EK_WAVE - 1);
    constexpr int NW = EK_CHAIN_THREADS / EK_WAVE;
    static_assert(EK_MAX_CANDS <= 2 * NW, "two states per wave at most");
    const int parts = cn <= NW / 2 ? 2 : 1;
    constexpr int U = 16;
    for (int w = wv / parts; w < cn; w += NW / parts) {
        const int part = wv % parts;
        const EkBlockMax *src = (w == 0) ? blockmax : pm + (size_t)(w - 1) * nbp;
        const int cnt = (w == 0) ? nb : nbp;
        float v = -__builtin_inff();
        uint32_t i = 0xffffffffu;
        for (int b0 = part * EK_WAVE + lane; b0 < cnt; b0 += parts * EK_WAVE * U) {
            EkBlockMax m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + u * parts * EK_WAVE;
                m[u] = ek_ld_bm<COH>(&src[b < cnt ? b : cnt - 1]);
                if (b >= cnt)
                    m[u].idx = 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (m[u].idx != 0xffffffffu && ek_better(m[u].val, m[u].idx, v, i)) {
                    v = m[u].val;
                    i = m[u].idx;
                }
        }
        ek_wave_argmax(v, i);
        if (lane == 0) {
            half_v[parts * w + part] = v;
            half_i[parts * w + part] = i;
        }
    }
    __syncthreads();
    if (tid < cn) {
        float v = half_v[parts * tid];
        uint32_t i = half_i[parts * tid];
        const float v2 = parts == 2 ? half_v[2 * tid + 1] : 0.f;
        const uint32_t i2 = parts == 2 ? half_i[2 * tid + 1] : 0xffffffffu;
        if (i2 != 0xffffffffu && (i == 0xffffffffu || ek_better(v2, i2, v, i))) {
            v = v2;
            i = i2;
        }
        out_v[tid] = v;
        out_i[tid] = i;
    }
}

// state k's global maximum is (v[k], g[k]) (ok[k] false: no frames anywhere)
__device__ __forceinline__ void ek_chain_walk(const float *v, const long long *g,
                                              const bool *ok, double cutoff,
                                              EkPlan *plan, EkHist *hist,
                                              EkCtl *ctl)
{
    const int cn = plan->chain_n;
    int napply = 0;
    plan->chain_label0 = ctl->n_done;
    for (int k = 0; k < cn; ++k) {
        if (ctl->stopped || ctl->n_done >= ctl->limit || !ok[k])
            break;
        ctl->last_max = v[k];
        if (!((double)v[k] > cutoff)) {     // kcenters.py:217
            ctl->stopped = 1;
            break;
        }
        const int j = plan->chain[k];
        if (g[k] != plan->gidx[j])
            break;                          // the farthest point is not stored
        const int label = ctl->n_done;
        hist[label].gidx = g[k];
        hist[label].dist = v[k];
        hist[label].set = 1;
        ctl->n_done = label + 1;
        plan->used |= 1u << j;
        ++napply;
    }
    plan->napply = napply;
}

