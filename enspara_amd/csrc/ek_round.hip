// ek_round.hip -- a k-centers round of ONE shard in three launches.
//
// What a round of candidates does (reference enspara/cluster/kcenters.py:217-231,
// :282, :298-306, reorganised as ek_spec.hip / ek_chain.hip describe) was ten
// launches on a single shard: plan, candidate tile, pass, per-prefix maxima,
// decide, apply, and four for the next round's candidate pick.  Seven of them are
// single-workgroup or tiny and each cost 5-19 us of mostly latency -- 13 % of a
// round at 10^6 frames, half of it at the 125 k frames per GPU of an 8-way
// split.  Here the single-workgroup steps ride at the end of the launch that
// produces their input: the LAST workgroup of a launch to finish (arrival
// counter; what it reads of the others' results travels as coherent stores /
// loads, ek_reduce.h) does them, so no workgroup ever waits for another.  A round is then
//
//   pass    ek_pass2_kernel<T, true, true> (ek_spec.hip): applies the chain the
//           previous round accepted on the way in (no apply pass), candidate 0,
//           stores the guesses' distances; its last workgroup: presumed
//           acceptance order of the guesses (ek_chain.hip step 1).
//   chain   per-256-frame maxima of the states every prefix of that order would
//           leave (one pass over the distance vectors); last workgroup: decide
//           the accepted prefix (steps 2-3, exactly ek_chain_walk), then the 64
//           farthest block maxima of the resulting state.
//   next    their pairwise distances (one wave per pair); last workgroup: greedy
//           choice of the next candidates, their records, the next plan (stop
//           rule kcenters.py:217, label, history) and the candidate tile the
//           next pass reads.
//
// Same decisions as the per-step kernels, taken from the same numbers: centers,
// labels and distances are unchanged.  The accepted chain stays pending between
// rounds (EkPend); ek_launch_round_flush applies it at the end of a run.
#include "ek_common.h"
#include "ek_qcp.h"
#include "ek_reduce.h"
#include "ek_chain_dev.h"
#include "ek_top_dev.h"

// measurement builds (-DEK_ROUND_STAMPS): the last workgroup of the chain kernel
// prints where its time went (100 MHz ticks), once, a few hundred rounds in
#ifdef EK_ROUND_STAMPS
__device__ unsigned int ek_stamp_count;
#define EK_STAMP(k) if (threadIdx.x == 0) st[k] = __builtin_amdgcn_s_memrealtime()
#else
#define EK_STAMP(k)
#endif

#define EK_ROUND_THREADS 1024       // chain kernel: also the width of its tail

// ---------------------------------------------------------------------------
// chain: states after the prefixes, decide, farthest block maxima
// ---------------------------------------------------------------------------
// pm[(k - 1) * nb + w] = first-index arg-max over frames [256 w, 256 w + 256) of
// min(dist, vec[order[0]], .., vec[order[k-1]]), k = 1 .. cn (state 0 is what the
// pass left in blockmax).  A wave covers 256 consecutive frames.
template <int NV>
__global__ void __launch_bounds__(EK_ROUND_THREADS)
ek_round_chain_kernel(EkRound r, int bootstrap)
{
    constexpr int EK_ROUND_FPT = 4;     // frames per thread (16-byte loads)
    __shared__ float sv[EK_MAX_CANDS];
    __shared__ uint32_t si[EK_MAX_CANDS];
    __shared__ int s_napply, s_fine;
    extern __shared__ uint32_t skip[];      // (unused since round 5)
    const int tid = threadIdx.x;
    const int nb = (int)((r.n + EK_BLOCK - 1) / EK_BLOCK);
    // (the maxima per 64 frames pay on shards of up to ~half a million frames -- 8 %
    // fewer rounds at 125 000 --; at 10^6 they changed 358 rounds to 355 and cost the
    // pick 10 us per round, two batches of loads instead of one: there it reads the
    // maxima per 256)
    const bool fine_ok = r.fm != nullptr && 4 * nb <= 8 * EK_RED_THREADS;
#ifdef EK_ROUND_STAMPS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    EK_STAMP(0);
    // (round 6) the pass took the per-prefix maxima itself: one workgroup, no sweep,
    // no ticket; candidate 0's entry in the history is made here
    const bool swept = r.sweep && NV == 16 && r.T == 16;
    if (!bootstrap) {
        if (!r.plan->go)
            return;             // the run is over: nothing changes any more
        const int cn = r.ord->n;
        if (cn > 0 && !swept) {
            const int64_t f0 = ((int64_t)blockIdx.x * EK_ROUND_THREADS + tid) *
                               EK_ROUND_FPT;
            const bool whole = f0 + EK_ROUND_FPT <= r.n;
            float run[EK_ROUND_FPT];
            const uint32_t vm = f0 < r.n ? r.vmask[f0 >> 6] : 0u;
            if (whole) {
                const float4 t = *(const float4 *)(r.dist + f0);
                run[0] = t.x; run[1] = t.y; run[2] = t.z; run[3] = t.w;
            } else {
#pragma unroll
                for (int q = 0; q < EK_ROUND_FPT; ++q)
                    run[q] = (f0 + q < r.n) ? r.dist[f0 + q] : 0.f;
            }
            const int64_t wg = ((int64_t)blockIdx.x * EK_ROUND_THREADS + tid) /
                               EK_WAVE;        // = 256-frame block of this wave
            // sixteen prefixes at a time (a thread holds a value per frame and
            // vector of them: rounds of 32 take two turns)
#pragma unroll
            for (int kb = 0; kb < NV; kb += 16) {
                if (kb >= cn + 1)               // uniform
                    break;
                float dv[16][EK_ROUND_FPT];
                // all loads first: the running minimum would serialise them
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const int k = kb + kk;
#pragma unroll
                    for (int q = 0; q < EK_ROUND_FPT; ++q)
                        dv[kk][q] = __builtin_inff();
                    // (a vector the pass did not store for these frames is all +inf)
                    if (k >= 1 && k <= cn && ((vm >> r.ord->cand[k - 1]) & 1u)) {
                        const float *v = r.vecs +
                                         (size_t)(r.ord->cand[k - 1] - 1) * r.n_pad + f0;
                        if (whole) {
                            const float4 t = *(const float4 *)v;
                            dv[kk][0] = t.x; dv[kk][1] = t.y; dv[kk][2] = t.z; dv[kk][3] = t.w;
                        } else {
#pragma unroll
                            for (int q = 0; q < EK_ROUND_FPT; ++q)
                                if (f0 + q < r.n)
                                    dv[kk][q] = v[q];
                        }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const int k = kb + kk;
                    if (k >= 1 && k <= cn) {        // uniform
                        float v = -__builtin_inff();
                        uint32_t i = 0xffffffffu;
#pragma unroll
                        for (int q = 0; q < EK_ROUND_FPT; ++q) {
                            if (f0 + q < r.n) {
                                if (dv[kk][q] < run[q])     // kcenters.py:304
                                    run[q] = dv[kk][q];
                                if (ek_better(run[q], (uint32_t)(f0 + q), v, i)) {
                                    v = run[q];
                                    i = (uint32_t)(f0 + q);
                                }
                            }
                        }
                        // (the rows of 16 lanes first: 64 frames each -- of the state the
                        // whole chain would leave, the finer maxima are kept for the
                        // candidate pick, "hidden frames" in ek_top_dev.h)
                        ek_row_argmax(v, i);
                        if (k == cn && fine_ok && (tid & 15) == 0 && wg < nb)
                            ek_coh_store_bm(&r.fm[4 * (size_t)wg + ((tid >> 4) & 3)], v, i);
                        ek_rows_to_wave_argmax(v, i);
                        // (read by the last workgroup of this launch: coherent store)
                        if ((tid & (EK_WAVE - 1)) == 0 && wg < nb)
                            ek_coh_store_bm(&r.pm[(size_t)(k - 1) * nb + wg], v, i);
                    }
                }
            }
        }
        EK_STAMP(1);
        if (!swept && !ek_arrive_last(r.tick + 1))
            return;
        EK_STAMP(2);
        // ---- decide (ek_chain.hip steps 2-3) ------------------------------------
        // states 0 .. cn - 1 are what the walk looks at
        __shared__ long long s_gidx[EK_MAX_CANDS];
        __shared__ int s_ord[EK_MAX_CANDS];
        if (tid < EK_MAX_CANDS) {
            s_gidx[tid] = r.plan->gidx[tid];
            s_ord[tid] = tid < cn ? r.ord->cand[tid] : 0;
        }
        ek_chain_reduce<true>(r.blockmax, r.pm, nb, nb, cn, sv, si);
        __syncthreads();
        EK_STAMP(3);
        if (tid == 0) {
            // ek_chain_walk on a register copy of the control words: one read,
            // one write-back, no dependent global reads in between
            EkCtl c = *r.ctl;
            if (swept) {
                // candidate 0 is a center now (kcenters.py:306-309): what the pass's
                // last workgroup wrote when it worked the order out
                const int lb = r.plan->label;
                r.hist[lb].gidx = r.plan->gidx[0];
                r.hist[lb].dist = r.plan->maxdist[0];
                r.hist[lb].set = 1;
                c.n_done = lb + 1;
                r.ctl->n_rounds = c.n_rounds + 1;
            }
            const int label0 = c.n_done;
            uint32_t used = r.plan->used;
            int na = 0;
            for (int k = 0; k < cn; ++k) {
                const bool ok = si[k] != 0xffffffffu;
                if (c.stopped || c.n_done >= c.limit || !ok)
                    break;
                c.last_max = sv[k];
                if (!((double)sv[k] > r.cutoff)) {      // kcenters.py:217
                    c.stopped = 1;
                    break;
                }
                const int j = s_ord[k];
                const long long g = r.goff + (long long)si[k];
                if (g != s_gidx[j]) {           // the farthest point is not stored
#ifdef EK_ROUND_STAMPS
                    // where was it?  (-1: not among the 64 block maxima the
                    // guesses were chosen from; else its rank there)
                    const EkTop *tp = (const EkTop *)r.top;
                    int pos = -1;
                    for (int q = 0; q < tp->n; ++q)
                        if (tp->idx[q] == si[k])
                            pos = q;
                    int other = -1;             // another candidate of this round?
                    for (int q = 1; q < EK_MAX_CANDS; ++q)
                        if (s_gidx[q] == g)
                            other = q;
                    // hidden behind a farther frame of its own workgroup of 256?
                    // (state 0 of this round: after candidate 0 alone)
                    const bool hidden = r.blockmax[si[k] / EK_BLOCK].idx != si[k];
                    // above the list's smallest value (it was cut by the per-label
                    // cap or the pool) or below it (the list was too short)?
                    const int above = tp->n > 0 && sv[k] > tp->val[tp->n - 1] ? 1 : 0;
                    printf("miss at %d of %d: farthest point rank %d in the list, "
                           "candidate %d, hidden %d, above the list's last %d\n", k, cn,
                           pos, other, hidden ? 1 : 0, above);
#endif
                    break;
                }
                const int label = c.n_done;
                r.hist[label].gidx = g;
                r.hist[label].dist = sv[k];
                r.hist[label].set = 1;
                c.n_done = label + 1;
                used |= 1u << j;
                r.pend->slot[na] = j - 1;
                ++na;
            }
            r.ctl->n_done = c.n_done;
            r.ctl->stopped = c.stopped;
            r.ctl->last_max = c.last_max;
            r.plan->used = used;
            r.plan->napply = na;
            r.plan->chain_label0 = label0;
            r.pend->n = na;
            r.pend->label0 = label0;
            s_napply = na;
            s_fine = (fine_ok && na > 0 && na == cn) ? 1 : 0;
            r.tick[1] = 0;
        }
        __syncthreads();
    } else {
        if (blockIdx.x != 0)
            return;
        if (tid == 0) {
            r.pend->n = 0;
            s_napply = 0;
            s_fine = 0;
        }
        __syncthreads();
    }
    // ---- the farthest block maxima of the state the accepted prefix leaves -------
    const int na = s_napply;
    // (the whole chain accepted -- most rounds --: the maxima per 64 frames of that
    // state; else the maxima per 256 of the state the accepted prefix leaves)
    const bool fine = s_fine != 0;
    const EkBlockMax *state = fine ? r.fm
                                   : (na == 0 ? r.blockmax : r.pm + (size_t)(na - 1) * nb);
    EkTop *top = (EkTop *)r.top;
    EK_STAMP(4);
    ek_pick_top_body<true>(state, fine ? 4 * nb : nb, top, skip, r.assign, r.pick_cap, EK_LIST_M);
#ifdef EK_ROUND_STAMPS
    __syncthreads();
    EK_STAMP(5);
    if (tid == 0 && !bootstrap) {       // how many different labels hold the list?
        int distinct = 0;
        for (int a = 0; a < top->n; ++a) {
            bool seen = false;
            for (int b = 0; b < a; ++b)
                seen = seen || r.assign[top->idx[b]] == r.assign[top->idx[a]];
            distinct += seen ? 0 : 1;
        }
        printf("list: %d entries, %d labels\n", top->n, distinct);
    }
    if (tid == 0 && !bootstrap && atomicAdd(&ek_stamp_count, 1u) % 200 == 150)
        printf("chain last wg (x10 ns): body %llu ticket %llu reduce %llu walk %llu "
               "pick %llu (setup+issue %llu, looks %llu, rank %llu, rest %llu)\n",
               st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3],
               st[5] - st[4], ek_pick_st[1] - ek_pick_st[0],
               ek_pick_st[2] - ek_pick_st[1], ek_pick_st[3] - ek_pick_st[2],
               st[5] - ek_pick_st[3]);
#endif
}

void ek_launch_round_chain(const EkRound &r, int bootstrap, hipStream_t s)
{
    if (r.n <= 0)
        return;
    const int64_t per = (int64_t)EK_ROUND_THREADS * 4;
    const unsigned blocks = (bootstrap || (r.sweep && r.T == 16))
                                ? 1u : (unsigned)((r.n + per - 1) / per);
    const int nb = (int)((r.n + EK_BLOCK - 1) / EK_BLOCK);
    const size_t lds = (size_t)((nb + 31) / 32 + 1) * sizeof(uint32_t);
    if (r.T > 16)
        hipLaunchKernelGGL(ek_round_chain_kernel<32>, dim3(blocks), dim3(EK_ROUND_THREADS),
                           lds, s, r, bootstrap);
    else
        hipLaunchKernelGGL(ek_round_chain_kernel<16>, dim3(blocks), dim3(EK_ROUND_THREADS),
                           lds, s, r, bootstrap);
}

// ---------------------------------------------------------------------------
// next: pairwise distances of the farthest frames, the next round's plan
// ---------------------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(EK_BLOCK)
ek_round_next_kernel(EkRound r, int bootstrap)
{
    // the run is over (a stop rule, or the goal was reached): nothing to prepare.
    // (go is cleared below, by the last workgroup of the launch that notices.)
    if (!bootstrap && !r.plan->go)
        return;
    const int tid = threadIdx.x;
#ifdef EK_P16_STATS     // (measurement build: what ek_pass16_kernel's waves counted)
    if (blockIdx.x == 0 && tid == 0) {
        unsigned int *t = r.tick + 256, v[13];
        for (int k = 0; k < 13; ++k) {
            v[k] = 0;
            for (int sl = 0; sl < 64; ++sl)
                v[k] += t[16 * sl + k];
        }
        if (v[0] >= 16 * 15000u) {
            printf("pass16 label %d: waves %u overflow %u queued %u drains %u; cycles per "
                   "wave: head %u loop %u certificates %u drain %u tail %u; a wave lives %u ns; "
                   "first to last wave of a workgroup %u ns; from a workgroup's end to the "
                   "next one's start on its CU %u ns (%u samples)\n",
                   r.ctl->n_done, v[0], v[1], v[2], v[3], 64 * (v[4] / v[0]),
                   64 * (v[5] / v[0]), 64 * (v[6] / v[0]), 64 * (v[7] / v[0]),
                   64 * (v[8] / v[0]), 10 * (v[9] / v[0]), 10 * (v[10] / (v[0] / 4)),
                   10 * (v[11] / (v[12] ? v[12] : 1)), v[12]);
            for (int k = 0; k < 64 * 16; ++k)
                t[k] = 0;
        }
    }
#endif
    const EkTop *top = (const EkTop *)r.top;
    const bool done = r.ctl->stopped || r.ctl->n_done >= r.ctl->limit;
    if (!done) {
        // one wave per pair; the values only steer the guesses, so the lanes may
        // stride over the atoms (summation order is free)
        const int lane = tid & (EK_WAVE - 1);
        const int w = blockIdx.x * (EK_BLOCK / EK_WAVE) + tid / EK_WAVE;
        const int i = w / EK_LIST_M, j = w % EK_LIST_M;
        if (i < j && j < top->n) {
            const int A = r.A;
            // straight from the frame-major copy: 12 A contiguous bytes each
            const float *x = r.aos + (size_t)top->idx[i] * 3 * A;
            const float *y = r.aos + (size_t)top->idx[j] * 3 * A;
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
            if (lane == 0) {
                const float d = ek_rmsd_from_S(S, r.G[top->idx[i]],
                                               r.G[top->idx[j]], A);
                // (read by the last workgroup of this launch: coherent stores)
                float *D = ek_top_D(r.top, A);
                ek_coh_store(&D[i * EK_LIST_M + j], d);
                ek_coh_store(&D[j * EK_LIST_M + i], d);
            }
        }
    }
    if (!ek_arrive_last_tree(r.tick + 2, r.tick + 64))
        return;
    // ---- the next round ----------------------------------------------------------
    extern __shared__ float sD[];               // [EK_LIST_M][EK_LIST_M]
    __shared__ float sval[EK_LIST_M];
    __shared__ uint32_t sidx[EK_LIST_M];
    __shared__ int sel[EK_MAX_CANDS];
    __shared__ int n_sel;
    {
        // the table of pairwise distances: coherent loads, all in flight before the
        // first is used
        const float *D = ek_top_D(r.top, r.A);
        constexpr int PER = EK_LIST_M * EK_LIST_M / EK_BLOCK;
        constexpr int CHUNK = PER < 16 ? PER : 16;
        for (int u0 = 0; u0 < PER; u0 += CHUNK) {
            float dreg[CHUNK];
#pragma unroll
            for (int u = 0; u < CHUNK; ++u)
                dreg[u] = ek_coh_load(&D[tid + (u0 + u) * EK_BLOCK]);
#pragma unroll
            for (int u = 0; u < CHUNK; ++u)
                sD[tid + (u0 + u) * EK_BLOCK] = dreg[u];
        }
        if (tid < EK_LIST_M) {
            sval[tid] = top->val[tid];
            sidx[tid] = top->idx[tid];
        }
        if (tid < EK_MAX_CANDS)
            sel[tid] = 0;
    }
    __syncthreads();
    if (tid < EK_WAVE) {
        // greedy order (ek_top_records_kernel): the frame with the largest
        // remaining distance, then every other one's is lowered by its distance
        // to it.  Entry 0 is the shard's first-index arg-max whatever D says.
        // (a lane holds entries lane, lane + 64, ..)
        constexpr int EPL = EK_LIST_M / EK_WAVE;
        const int nt = top->n;
        const int lane = tid;
        bool open[EPL];
        float cur[EPL];
        uint32_t ix[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            open[e] = lane + e * EK_WAVE < nt;
            cur[e] = open[e] ? sval[lane + e * EK_WAVE] : 0.f;
            ix[e] = sidx[lane + e * EK_WAVE];
        }
        int ns = 0;
        while (ns < T) {
            float v = -__builtin_inff();
            uint32_t i = 0xffffffffu;
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                if (open[e] && ek_better(cur[e], ix[e], v, i)) {
                    v = cur[e];
                    i = ix[e];
                }
            ek_wave_argmax(v, i);
            if (i == 0xffffffffu)
                break;
            int best = -1;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const unsigned long long who = __ballot(open[e] && ix[e] == i);
                if (who && best < 0)
                    best = e * EK_WAVE + (__ffsll((long long)who) - 1);
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                if (lane + e * EK_WAVE == best)
                    open[e] = false;
            if (lane == 0)
                sel[ns] = best;
            ++ns;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const float d = sD[best * EK_LIST_M + lane + e * EK_WAVE];
                if (open[e] && d < cur[e])
                    cur[e] = d;
            }
        }
        if (lane == 0)
            n_sel = ns;
    }
    __syncthreads();
    const int ns = n_sel;
    const size_t rstride = ek_rec_bytes(r.A);
    const float first_max = ns > 0 ? sval[sel[0]] : -__builtin_inff();
    // the round runs unless a stop rule says otherwise (kcenters.py:217)
    const bool go = !done && ns > 0 && (double)first_max > r.cutoff;
    // the records (kept for the other entry points: [0] = the shard's farthest
    // point, its distance = distances.max(), kcenters.py:226) ...
    if (tid < T) {
        EkRecHdr *h = (EkRecHdr *)(r.recs + (size_t)tid * rstride);
        EkPlan *plan = r.plan;
        if (tid < ns) {
            const uint32_t fi = sidx[sel[tid]];
            const double tr = r.G[fi];
            h->maxdist = sval[sel[tid]];
            h->valid = 1;
            h->gidx = r.goff + (int64_t)fi;
            h->trace = tr;
            h->reserved = 0;
            plan->src[tid] = tid;
            plan->gidx[tid] = r.goff + (int64_t)fi;
            plan->maxdist[tid] = sval[sel[tid]];
            plan->trace[tid] = tr;
            r.ctrace[tid] = tr;
        } else {
            h->maxdist = -__builtin_inff();
            h->valid = 0;
            h->gidx = -1;
            h->trace = 0.0;
            h->reserved = 0;
            r.ctrace[tid] = 0.0;
        }
    }
    // ... and, from the same reads, the candidate tile of the next pass:
    // (ek_ctile_index), zeros for unused slots and the atoms of padding.
    // All of a trip's loads (T candidates x 4 rows per thread) go out before
    // the first store: the stores may alias them as far as the compiler knows.
    // (T >= 16: the coordinates are ek_round_ctile16_kernel's, a launch of its own
    // -- written by this one workgroup they were 20 of the kernel's 39 us)
    const int A3 = 3 * r.A;
    if (T >= 16 && tid == 0)
        r.plan->n_rec = ns;
    for (int k0 = 0; T < 16 && k0 * EK_BLOCK < A3; k0 += 4) {
        float v[T][4];
#pragma unroll
        for (int c = 0; c < T; ++c) {
            const float *src = r.aos + (size_t)sidx[sel[c < ns ? c : 0]] * A3;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = tid + (k0 + u) * EK_BLOCK;
                v[c][u] = (c < ns && row < A3) ? src[row] : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < T; ++c) {
            float *rec = (float *)(r.recs + (size_t)c * rstride + sizeof(EkRecHdr));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = tid + (k0 + u) * EK_BLOCK;
                if (row >= A3)
                    continue;
                if (c < ns)
                    rec[row] = v[c][u];
                if (go)         // (zero for a slot without a candidate)
                    r.ctile[ek_ctile_index(T, row / 3, c, row % 3)] = v[c][u];
            }
        }
    }
    if (go && T < 16)                           // the atoms of padding
        for (int k = tid; k < (ek_ctile_atoms(r.A) - r.A) * 3 * T; k += EK_BLOCK)
            r.ctile[ek_ctile_index(T, r.A + k / (3 * T), (k % (3 * T)) / 3, k % 3)] =
                0.f;
    if (tid == 0) {
        EkPlan *plan = r.plan;
        r.ctl->last_max = first_max;
        plan->apply = -1;
        plan->chain_n = 0;
        plan->napply = 0;
        if (go) {
            // (the pass counts candidate 0 and writes its history entry once
            // its distances are in)
            plan->go = 1;
            plan->teff = ns;
            if (r.sweep && T == 16) {
                // the presumed order of the round's chain = the order the greedy
                // choice above took the candidates in (round 6: known before the
                // pass, which takes the per-prefix maxima along it)
                r.ord->n = ns - 1;
                for (int k = 0; k + 1 < ns; ++k)
                    r.ord->cand[k] = k + 1;
            }
            plan->label = r.ctl->n_done;
            plan->used = 1;
            plan->miss = 0;
        } else {
            plan->go = 0;
            plan->teff = 0;
            plan->used = 0;
            plan->miss = 1;
            if (!done && ns > 0)
                r.ctl->stopped = 1;     // maxdist <= cutoff
        }
    }
}

// The chosen frames' coordinates of a round of 16 or 32: into the round's records
// and, in 16-byte pieces, into the candidate tile(s) of the next pass
// (ek_ctile_index: one piece = the four trips of (16 atoms, axis, lane); candidates
// 16 .. 31 in a second tile).  Some thirty workgroups (sixty) side by side instead
// of the planning kernel's last one alone.  Grid: [tile blocks of the first
// tile | of the second | one workgroup per record].
__global__ void __launch_bounds__(EK_BLOCK)
ek_round_ctile16_kernel(EkRound r, int halves)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t sfr[EK_MAX_CANDS];
    const EkPlan *plan = r.plan;
    const int tid = threadIdx.x;
    const int ns = plan->n_rec, go = plan->go;
    if (ns <= 0)
        return;
    if (tid < EK_MAX_CANDS)
        sfr[tid] = tid < ns ? (uint32_t)(plan->gidx[tid] - r.goff) : 0u;
    __syncthreads();
    const int A = r.A, A3 = 3 * A;
    const int n_ct = ek_ctile_atoms(A) / 16 * 3;        // (16 atoms, axis) blocks of 1 KB
    const int ct_wgs = (n_ct + 3) / 4;
    if ((int)blockIdx.x < halves * ct_wgs) {
        const int half = blockIdx.x / ct_wgs;
        const int blk = (blockIdx.x % ct_wgs) * 4 + (tid >> 6), lane = tid & 63;
        if (!go || blk >= n_ct || 16 * half >= ns)
            return;
        const int S = blk / 3, k = blk % 3, kk = lane >> 4, c = 16 * half + (lane & 15);
        const float *src = r.aos + (size_t)sfr[c] * A3 + k;
        v4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = 16 * S + 4 * q + kk;
            v[q] = (a < A && c < ns) ? src[3 * a] : 0.f;
        }
        *(v4 *)(r.ctile + half * ek_ctile_half_floats(A) +
                ek_ctile_index(16, 16 * S + kk, c, k)) = v;
    } else {
        const int c = blockIdx.x - halves * ct_wgs;     // one record per workgroup
        if (c >= ns)
            return;
        const float *src = r.aos + (size_t)sfr[c] * A3;
        float *rec = (float *)(r.recs + (size_t)c * ek_rec_bytes(A) + sizeof(EkRecHdr));
        for (int row = tid; row < A3; row += EK_BLOCK)
            rec[row] = src[row];
    }
}

void ek_launch_round_next(const EkRound &r, int bootstrap, hipStream_t s)
{
    if (r.n <= 0)
        return;
    const unsigned blocks = (unsigned)(EK_LIST_M * EK_LIST_M / (EK_BLOCK / EK_WAVE));
    const size_t lds = (size_t)EK_LIST_M * EK_LIST_M * sizeof(float);
#define EK_NEXT_LDS(TT)                                                        \
    if (lds > 48 * 1024)                                                       \
        (void)hipFuncSetAttribute((const void *)ek_round_next_kernel<TT>,      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
    if (r.T >= 16) {
        const int halves = r.T / 16;
        if (r.T == 32) {
            EK_NEXT_LDS(32);
            hipLaunchKernelGGL((ek_round_next_kernel<32>), dim3(blocks), dim3(EK_BLOCK), lds,
                               s, r, bootstrap);
        } else {
            EK_NEXT_LDS(16);
            hipLaunchKernelGGL((ek_round_next_kernel<16>), dim3(blocks), dim3(EK_BLOCK), lds,
                               s, r, bootstrap);
        }
        const unsigned wgs =
            (unsigned)(halves * ((ek_ctile_atoms(r.A) / 16 * 3 + 3) / 4) + r.T);
        hipLaunchKernelGGL(ek_round_ctile16_kernel, dim3(wgs), dim3(EK_BLOCK), 0, s, r,
                           halves);
    } else if (r.T == 8) {
        EK_NEXT_LDS(8);
        hipLaunchKernelGGL((ek_round_next_kernel<8>), dim3(blocks), dim3(EK_BLOCK), lds,
                           s, r, bootstrap);
    } else {
        EK_NEXT_LDS(4);
        hipLaunchKernelGGL((ek_round_next_kernel<4>), dim3(blocks), dim3(EK_BLOCK), lds,
                           s, r, bootstrap);
    }
#undef EK_NEXT_LDS
}

// ---------------------------------------------------------------------------
// flush: apply what is pending, leave blockmax describing the state
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(EK_BLOCK)
ek_round_flush_kernel(EkRound r)
{
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];
    const int na = r.pend->n;
    const int label0 = r.pend->label0;
    const int tid = threadIdx.x;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + tid;
    float v = -__builtin_inff();
    uint32_t i = 0xffffffffu;
    if (f < r.n) {
        float cur = r.dist[f];
        int32_t lab = -1;
        const uint32_t vm = na > 0 ? r.vmask[f >> 6] : 0u;
        for (int k = 0; k < na; ++k) {      // kcenters.py:304-306, in order
            if (!((vm >> (r.pend->slot[k] + 1)) & 1u))
                continue;                   // not stored: +inf
            const float d = r.vecs[(size_t)r.pend->slot[k] * r.n_pad + f];
            if (d < cur) {
                cur = d;
                lab = label0 + k;
            }
        }
        if (lab >= 0) {
            r.dist[f] = cur;
            r.assign[f] = lab;
        }
        v = cur;
        i = (uint32_t)f;
    }
    ek_wave_argmax(v, i);
    if ((tid & (EK_WAVE - 1)) == 0) {
        red_v[tid / EK_WAVE] = v;
        red_i[tid / EK_WAVE] = i;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
            if (ek_better(red_v[w], red_i[w], v, i)) {
                v = red_v[w];
                i = red_i[w];
            }
        r.blockmax[blockIdx.x].val = v;
        r.blockmax[blockIdx.x].idx = i;
    }
}

void ek_launch_round_flush(const EkRound &r, hipStream_t s)
{
    if (r.n <= 0)
        return;
    hipLaunchKernelGGL(ek_round_flush_kernel,
                       dim3((unsigned)((r.n + EK_BLOCK - 1) / EK_BLOCK)),
                       dim3(EK_BLOCK), 0, s, r);
    // nothing is pending any more
    (void)hipMemsetAsync(&r.pend->n, 0, sizeof(int32_t), s);
}

// ---------------------------------------------------------------------------
// triangle inequality for rounds (reference kcenters.py:287-296, `use_triangle_
// inequality`; round 5)
// ---------------------------------------------------------------------------
// The reference skips, per new center, the frames whose own center is at least
// twice their distance away from it.  For a ROUND of T candidates the unit that
// can be left out is a (tile of 256 frames, candidate) pair: candidate c cannot
// change any frame of the tile if D(center of f, candidate c) >= 2 d(f) for every
// frame f of it (margin 0.1 % + 1e-3, as in the one-center form) -- neither as
// the new center (candidate 0) nor as a kept distance (a kept distance that is
// not below the frame's own is +inf for every reader).  The certificate holds
// for a label and distance that are STALE too (the accepted chain still pending,
// EkPend): d(f) is the frame's distance to the center its label names whatever
// came later, and the frame's current distance is no larger.
//   ek_round_ti_centers_kernel   D[label][c] for all centers so far x the round's
//                                candidates: a wave per center, lane = (candidate,
//                                quarter of the atoms); summation order free
//   ek_round_ti_tiles_kernel     per tile the candidates that may still matter
// A tile with an empty mask is not read by the pass (ek_pass16_kernel: pending
// chain and maxima only); in a partly masked tile the masked candidates' pairs
// skip their certificates and solves.  Masks never change a result: a pair that
// is not masked is simply computed.
__global__ void __launch_bounds__(EK_BLOCK)
ek_round_ti_centers_kernel(EkRound r)
{
    const EkPlan *plan = r.plan;
    if (!plan->go)
        return;
    const int lane = threadIdx.x & (EK_WAVE - 1);
    const int l = blockIdx.x * (EK_BLOCK / EK_WAVE) + threadIdx.x / EK_WAVE;
    const int k = plan->label;          // labels 0 .. k - 1 exist (candidate 0 gets k)
    if (l >= k)
        return;
    const int A = r.A, teff = plan->teff;
    const int64_t f = r.hist[l].gidx - r.goff;
    const float *x = r.aos + (size_t)f * 3 * A;
    const double Gx = r.G[f];
    const size_t rstride = ek_rec_bytes(A);
    const int c16 = lane & 15, s = lane >> 4;
    for (int cb = 0; cb < teff; cb += 16) {
        const int c = cb + c16;
        const bool live = c < teff;
        const float *y = (const float *)(r.recs + (size_t)(live ? c : 0) * rstride +
                                         sizeof(EkRecHdr));
        float S[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int a = s; a < A; a += 4) {
            const float x0 = x[3 * a], x1 = x[3 * a + 1], x2 = x[3 * a + 2];
            const float y0 = y[3 * a], y1 = y[3 * a + 1], y2 = y[3 * a + 2];
            S[0] += x0 * y0; S[1] += x0 * y1; S[2] += x0 * y2;
            S[3] += x1 * y0; S[4] += x1 * y1; S[5] += x1 * y2;
            S[6] += x2 * y0; S[7] += x2 * y1; S[8] += x2 * y2;
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            S[q] += __shfl_xor(S[q], 16, 64);
            S[q] += __shfl_xor(S[q], 32, 64);
        }
        if (s == 0 && live)
            r.ti_tab[(size_t)l * EK_MAX_CANDS + c] =
                ek_rmsd_from_S(S, Gx, plan->trace[c], A);
    }
}

__global__ void __launch_bounds__(EK_BLOCK)
ek_round_ti_tiles_kernel(EkRound r)
{
    __shared__ uint32_t s_need;
    const EkPlan *plan = r.plan;
    if (!plan->go)
        return;
    if (threadIdx.x == 0)
        s_need = 0;
    __syncthreads();
    const int k = plan->label, teff = plan->teff;
    const uint32_t all = teff >= 32 ? 0xffffffffu : ((1u << teff) - 1u);
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    uint32_t need = 0;
    if (f < r.n) {
        const int32_t a = r.assign[f];
        if (a < 0 || a >= k) {
            need = all;         // (no center yet, or a label this table does not have)
        } else {
            const float thr = 2.0f * r.dist[f] * 1.001f + 1e-3f;
            const float4 *row = (const float4 *)(r.ti_tab + (size_t)a * EK_MAX_CANDS);
#pragma unroll
            for (int q = 0; q < EK_MAX_CANDS / 4; ++q) {
                if (4 * q >= teff)
                    break;
                const float4 v = row[q];
                need |= (!(v.x >= thr) ? 1u : 0u) << (4 * q);
                need |= (!(v.y >= thr) ? 2u : 0u) << (4 * q);
                need |= (!(v.z >= thr) ? 4u : 0u) << (4 * q);
                need |= (!(v.w >= thr) ? 8u : 0u) << (4 * q);
            }
            need &= all;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        need |= (uint32_t)__shfl_xor((int)need, off, 64);
    if ((threadIdx.x & (EK_WAVE - 1)) == 0 && need)
        atomicOr(&s_need, need);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t m = s_need;
        r.tmask[blockIdx.x] = m;
        // (honest counts: a partly masked tile is still streamed for all its
        // candidates; only a tile nobody can change is left out)
        atomicAdd(&r.ti_stats[0], (unsigned long long)teff);
        if (m == 0)
            atomicAdd(&r.ti_stats[1], (unsigned long long)teff);
    }
}

void ek_launch_round_ti(const EkRound &r, int max_labels, hipStream_t s)
{
    if (r.n <= 0 || !r.tmask || max_labels <= 0)
        return;
    const int per = EK_BLOCK / EK_WAVE;
    hipLaunchKernelGGL(ek_round_ti_centers_kernel, dim3((max_labels + per - 1) / per),
                       dim3(EK_BLOCK), 0, s, r);
    hipLaunchKernelGGL(ek_round_ti_tiles_kernel,
                       dim3((unsigned)((r.n + EK_BLOCK - 1) / EK_BLOCK)), dim3(EK_BLOCK), 0,
                       s, r);
}
