// ek_chain.hip -- the cheap steps of a multi-candidate k-centers round, chained.
//
// After the pass of a round (ek_spec.hip) the shard holds the distance vectors
// of candidates 1..T-1.  The sequential algorithm (reference
// enspara/cluster/kcenters.py:217-231, :282, :298-306) now repeats: take the
// first-index arg-max of the current distances; if it is a stored, unused
// candidate, apply its vector with the next label; otherwise the round ends.
// One launch pair (and, across GPUs, one exchange) per accepted center is what
// that costs when done literally.  Here the whole chain is decided at once:
//
//  1. order   -- if every next farthest point IS a stored candidate, the order in
//     which the candidates are taken depends only on the candidates' own current
//     distances and their distances to one another (a (T-1) x (T-1) table read
//     from the stored vectors): take the unused candidate with the largest
//     current distance, lower the others' by their distance to it, repeat.
//  2. maxima  -- one pass over the distance vectors gives, for every prefix of
//     that order, the first-index arg-max of the state the prefix would leave.
//  3. decide  -- walk the prefixes: the k-th candidate of the order is accepted
//     iff the arg-max of the state before it is that candidate's frame (and the
//     stop rules allow); the first failure ends the round.  If the true farthest
//     point is an unused candidate it is necessarily the one step 1 presumed,
//     so this accepts exactly what the literal loop accepts.
//  4. apply   -- one pass applies the accepted prefix (strict <, labels in order)
//     and leaves the per-workgroup maxima for the next round's candidate pick.
//
// Across shards steps 1 and 3 work on all-gathered rows / per-prefix maxima: two
// small exchanges per round instead of one per accepted center.
#include "ek_common.h"
#include "ek_reduce.h"
#include "ek_chain_dev.h"
#include <algorithm>

// ---- 1. order ---------------------------------------------------------------------
// the rows of the candidate frames this shard owns
__device__ __forceinline__ void ek_chain_row(const EkPlan *plan, int j,
                                             const float *dist, const float *vecs,
                                             int64_t n, int64_t n_pad,
                                             int64_t global_offset,
                                             EkChainRow *row)
{
    row->valid = 0;
    row->cur = 0.f;
    for (int u = 0; u < EK_MAX_CANDS; ++u)
        row->d[u] = 0.f;
    if (!plan->go || j < 1 || j >= plan->teff)
        return;
    const int64_t local = plan->gidx[j] - global_offset;
    if (local < 0 || local >= n)
        return;
    row->valid = 1;
    row->cur = dist[local];
    for (int u = 1; u < plan->teff; ++u)
        row->d[u] = vecs[(size_t)(u - 1) * n_pad + local];
}

__global__ void __launch_bounds__(EK_WAVE)
ek_chain_rows_kernel(const EkPlan *__restrict__ plan,
                     const float *__restrict__ dist,
                     const float *__restrict__ vecs, int64_t n, int64_t n_pad,
                     int64_t global_offset, EkChainRow *__restrict__ rows_out)
{
    const int j = threadIdx.x;
    if (j < EK_MAX_CANDS)
        ek_chain_row(plan, j, dist, vecs, n, n_pad, global_offset, &rows_out[j]);
}

void ek_launch_chain_rows(const EkPlan *plan, const float *dist,
                          const float *vecs, int64_t n, int64_t n_pad,
                          int64_t global_offset, EkChainRow *rows_out,
                          hipStream_t s)
{
    hipLaunchKernelGGL(ek_chain_rows_kernel, dim3(1), dim3(EK_WAVE), 0, s, plan,
                       dist, vecs, n, n_pad, global_offset, rows_out);
}

// ---- 2. maxima of the states the prefixes would leave --------------------------------
// pm[(k - 1) * nb + workgroup] = first-index arg-max over the workgroup's frames
// of min(dist, vec[chain[0]], .., vec[chain[k-1]]), k = 1 .. chain_n - 1.
// (State 0, before any of them, is what the pass kernel left in blockmax.)
#define EK_CHAIN_FPT 4      // frames per thread in ek_chain_max_kernel

__global__ void __launch_bounds__(EK_BLOCK)
ek_chain_max_kernel(const float *__restrict__ dist,
                    const float *__restrict__ vecs, int64_t n, int64_t n_pad,
                    EkPlan *plan, EkBlockMax *__restrict__ pm, int local_order,
                    int64_t global_offset)
{
    __shared__ float red_v[EK_MAX_CANDS][EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_MAX_CANDS][EK_BLOCK / EK_WAVE];
    __shared__ EkChainRow rows[EK_MAX_CANDS];
    __shared__ int s_chain[EK_MAX_CANDS];
    __shared__ int s_cn;
    const int tid = threadIdx.x;
    if (local_order) {
        // single shard: every workgroup works the presumed order out for itself
        // (64 reads and a few hundred scalar steps) instead of waiting for a
        // launch that does it once; workgroup 0 records it for the kernels that
        // follow.  Nobody reads plan->chain* in this launch.
        ek_chain_rows_local(plan, dist, vecs, n, n_pad, global_offset, rows, tid);
        __syncthreads();
        if (tid == 0) {
            int chain[EK_MAX_CANDS];
            const int c = plan->go ? ek_chain_simulate(plan, rows, chain) : 0;
            for (int k = 0; k < c; ++k)
                s_chain[k] = chain[k];
            s_cn = c;
            if (blockIdx.x == 0) {
                for (int k = 0; k < c; ++k)
                    plan->chain[k] = chain[k];
                plan->chain_n = c;
                plan->napply = 0;
                plan->chain_label0 = 0;
            }
        }
        __syncthreads();
    } else {
        if (tid == 0) {
            s_cn = plan->chain_n;
            for (int k = 0; k < EK_MAX_CANDS; ++k)
                s_chain[k] = plan->chain[k];
        }
        __syncthreads();
    }
    const int cn = s_cn;
    if (cn <= 1)
        return;
    const int64_t f0 = ((int64_t)blockIdx.x * EK_BLOCK + tid) * EK_CHAIN_FPT;
    const int nbp = gridDim.x;
    const bool whole = f0 + EK_CHAIN_FPT <= n;      // 16-byte loads
    // all loads first: the running minimum would otherwise serialise them
    float run[EK_CHAIN_FPT];
    float dv[EK_LEGACY_CANDS][EK_CHAIN_FPT];    // (these forms run <= 16 candidates)
#pragma unroll
    for (int k = 1; k < EK_LEGACY_CANDS - 1; ++k) {
#pragma unroll
        for (int q = 0; q < EK_CHAIN_FPT; ++q)
            dv[k][q] = __builtin_inff();
        if (k < cn) {
            const float *v = vecs + (size_t)(s_chain[k - 1] - 1) * n_pad + f0;
            if (whole) {
                const float4 t = *(const float4 *)v;
                dv[k][0] = t.x; dv[k][1] = t.y; dv[k][2] = t.z; dv[k][3] = t.w;
            } else {
#pragma unroll
                for (int q = 0; q < EK_CHAIN_FPT; ++q)
                    if (f0 + q < n)
                        dv[k][q] = v[q];
            }
        }
    }
    if (whole) {
        const float4 t = *(const float4 *)(dist + f0);
        run[0] = t.x; run[1] = t.y; run[2] = t.z; run[3] = t.w;
    } else {
#pragma unroll
        for (int q = 0; q < EK_CHAIN_FPT; ++q)
            run[q] = (f0 + q < n) ? dist[f0 + q] : 0.f;
    }
#pragma unroll
    for (int k = 1; k < EK_LEGACY_CANDS - 1; ++k) {
        if (k < cn) {                       // uniform
            float v = -__builtin_inff();
            uint32_t i = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < EK_CHAIN_FPT; ++q) {
                if (f0 + q < n) {
                    if (dv[k][q] < run[q])
                        run[q] = dv[k][q];
                    if (ek_better(run[q], (uint32_t)(f0 + q), v, i)) {
                        v = run[q];
                        i = (uint32_t)(f0 + q);
                    }
                }
            }
            ek_wave_argmax(v, i);
            if ((tid & (EK_WAVE - 1)) == 0) {
                red_v[k][tid / EK_WAVE] = v;
                red_i[k][tid / EK_WAVE] = i;
            }
        }
    }
    __syncthreads();
    if (tid >= 1 && tid < cn) {
        const int k = tid;
        float v = red_v[k][0];
        uint32_t i = red_i[k][0];
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
            if (ek_better(red_v[k][w], red_i[k][w], v, i)) {
                v = red_v[k][w];
                i = red_i[k][w];
            }
        pm[(size_t)(k - 1) * nbp + blockIdx.x].val = v;
        pm[(size_t)(k - 1) * nbp + blockIdx.x].idx = i;
    }
}

// workgroups of ek_chain_max_kernel (= entries per prefix in pm)
int ek_chain_max_blocks(int64_t n)
{
    const int64_t per = (int64_t)EK_BLOCK * EK_CHAIN_FPT;
    return (int)((n + per - 1) / per);
}

// local_order != 0: single shard, the order is worked out here
void ek_launch_chain_max(const float *dist, const float *vecs, int64_t n,
                         int64_t n_pad, EkPlan *plan, EkBlockMax *pm,
                         int local_order, int64_t global_offset, hipStream_t s)
{
    if (n <= 0)
        return;
    hipLaunchKernelGGL(ek_chain_max_kernel, dim3((unsigned)ek_chain_max_blocks(n)),
                       dim3(EK_BLOCK), 0, s, dist, vecs, n, n_pad, plan, pm,
                       local_order, global_offset);
}

// ---- 1 + 2 + this shard's part of 3, one launch (the multi-shard round) ----------------
// Steps 1, 2 and the shard-local half of 3: every workgroup works the presumed order out of the gathered
// rows for itself (64 reads and a few hundred scalar steps; workgroup 0 records
// it in the plan for the apply step), a wave covers 256 consecutive frames and
// leaves their first-index arg-max after every prefix of the order in pm (as
// coherent stores), and the last workgroup to finish (arrival counter,
// ek_reduce.h) reduces them to this shard's (max distance, global index) per
// prefix: the 128 bytes that go into the second exchange.
#define EK_CM2_THREADS 1024
#define EK_CM2_FPT 4

__global__ void __launch_bounds__(EK_CM2_THREADS)
ek_chain_max2_kernel(const float *__restrict__ dist, const float *__restrict__ vecs,
                     int64_t n, int64_t n_pad, EkPlan *plan,
                     const EkChainRow *__restrict__ rows_all, int n_shards,
                     const EkBlockMax *__restrict__ blockmax, EkBlockMax *pm,
                     int64_t global_offset, EkMaxHdr *__restrict__ hdrs_out,
                     unsigned int *tick)
{
    __shared__ EkChainRow rows[EK_MAX_CANDS];
    __shared__ int s_chain[EK_MAX_CANDS];
    __shared__ int s_cn;
    __shared__ float sv[EK_MAX_CANDS];
    __shared__ uint32_t si[EK_MAX_CANDS];
    const int tid = threadIdx.x;
    const int nb = (int)((n + EK_BLOCK - 1) / EK_BLOCK);
    if (tid < EK_MAX_CANDS) {           // the owner's row among the shards'
        rows[tid].valid = 0;
        for (int sh = 0; sh < n_shards; ++sh) {
            const EkChainRow *r = &rows_all[(size_t)sh * EK_MAX_CANDS + tid];
            if (r->valid) {
                rows[tid] = *r;
                break;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        int chain[EK_MAX_CANDS];
        const int c = plan->go ? ek_chain_simulate(plan, rows, chain) : 0;
        for (int k = 0; k < c; ++k)
            s_chain[k] = chain[k];
        s_cn = c;
        if (blockIdx.x == 0) {          // (nobody reads plan->chain* in this launch)
            for (int k = 0; k < c; ++k)
                plan->chain[k] = chain[k];
            plan->chain_n = c;
            plan->napply = 0;
            plan->chain_label0 = 0;
        }
    }
    __syncthreads();
    const int cn = s_cn;
    if (cn > 1) {
        const int64_t f0 = ((int64_t)blockIdx.x * EK_CM2_THREADS + tid) * EK_CM2_FPT;
        const bool whole = f0 + EK_CM2_FPT <= n;
        float run[EK_CM2_FPT];
        float dv[EK_LEGACY_CANDS][EK_CM2_FPT];
        // all loads first: the running minimum would otherwise serialise them
#pragma unroll
        for (int k = 1; k < EK_LEGACY_CANDS - 1; ++k) {
#pragma unroll
            for (int q = 0; q < EK_CM2_FPT; ++q)
                dv[k][q] = __builtin_inff();
            if (k < cn) {
                const float *v = vecs + (size_t)(s_chain[k - 1] - 1) * n_pad + f0;
                if (whole) {
                    const float4 t = *(const float4 *)v;
                    dv[k][0] = t.x; dv[k][1] = t.y; dv[k][2] = t.z; dv[k][3] = t.w;
                } else {
#pragma unroll
                    for (int q = 0; q < EK_CM2_FPT; ++q)
                        if (f0 + q < n)
                            dv[k][q] = v[q];
                }
            }
        }
        if (whole) {
            const float4 t = *(const float4 *)(dist + f0);
            run[0] = t.x; run[1] = t.y; run[2] = t.z; run[3] = t.w;
        } else {
#pragma unroll
            for (int q = 0; q < EK_CM2_FPT; ++q)
                run[q] = (f0 + q < n) ? dist[f0 + q] : 0.f;
        }
        const int64_t wg = ((int64_t)blockIdx.x * EK_CM2_THREADS + tid) / EK_WAVE;
#pragma unroll
        for (int k = 1; k < EK_LEGACY_CANDS - 1; ++k) {
            if (k < cn) {                       // uniform
                float v = -__builtin_inff();
                uint32_t i = 0xffffffffu;
#pragma unroll
                for (int q = 0; q < EK_CM2_FPT; ++q) {
                    if (f0 + q < n) {
                        if (dv[k][q] < run[q])      // kcenters.py:304
                            run[q] = dv[k][q];
                        if (ek_better(run[q], (uint32_t)(f0 + q), v, i)) {
                            v = run[q];
                            i = (uint32_t)(f0 + q);
                        }
                    }
                }
                ek_wave_argmax(v, i);
                if ((tid & (EK_WAVE - 1)) == 0 && wg < nb)
                    ek_coh_store_bm(&pm[(size_t)(k - 1) * nb + wg], v, i);
            }
        }
    }
    if (!ek_arrive_last(tick))
        return;
    // ---- the last workgroup: this shard's maximum after every prefix ------------------
    // (state 0 is what the pass left in blockmax: written by the launch before)
    ek_chain_reduce<true>(blockmax, pm, nb, nb, cn, sv, si);
    __syncthreads();
    if (tid < EK_MAX_CANDS) {
        const bool ok = tid < cn && si[tid] != 0xffffffffu;
        hdrs_out[tid].maxdist = ok ? sv[tid] : -__builtin_inff();
        hdrs_out[tid].valid = ok ? 1 : 0;
        hdrs_out[tid].gidx = ok ? global_offset + (int64_t)si[tid] : -1;
    }
    if (tid == 0)
        *tick = 0;
}

void ek_launch_chain_max2(const float *dist, const float *vecs, int64_t n,
                          int64_t n_pad, EkPlan *plan, const EkChainRow *rows_all,
                          int n_shards, const EkBlockMax *blockmax, EkBlockMax *pm,
                          int64_t global_offset, EkMaxHdr *hdrs_out,
                          unsigned int *tick, hipStream_t s)
{
    const int64_t per = (int64_t)EK_CM2_THREADS * EK_CM2_FPT;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, (n + per - 1) / per);
    hipLaunchKernelGGL(ek_chain_max2_kernel, dim3(blocks), dim3(EK_CM2_THREADS), 0, s,
                       dist, vecs, n, n_pad, plan, rows_all, n_shards, blockmax, pm,
                       global_offset, hdrs_out, tick);
}

// ---- 3. decide ----------------------------------------------------------------------
__global__ void __launch_bounds__(EK_WAVE)
ek_chain_decide_kernel(const EkMaxHdr *__restrict__ hdrs_all, int n_shards,
                       double cutoff, EkPlan *__restrict__ plan,
                       EkHist *__restrict__ hist, EkCtl *__restrict__ ctl)
{
    if (threadIdx.x != 0)
        return;
    plan->napply = 0;
    if (!plan->go)
        return;
    float v[EK_MAX_CANDS];
    long long g[EK_MAX_CANDS];
    bool ok[EK_MAX_CANDS];
    for (int k = 0; k < EK_MAX_CANDS; ++k) {
        ok[k] = false;
        v[k] = 0.f;
        g[k] = 0;
        for (int sh = 0; sh < n_shards; ++sh) {   // largest, lowest index on ties
            const EkMaxHdr h = hdrs_all[(size_t)sh * EK_MAX_CANDS + k];
            if (!h.valid)
                continue;
            if (!ok[k] || h.maxdist > v[k] ||
                (h.maxdist == v[k] && h.gidx < g[k])) {
                ok[k] = true;
                v[k] = h.maxdist;
                g[k] = h.gidx;
            }
        }
    }
    ek_chain_walk(v, g, ok, cutoff, plan, hist, ctl);
}

void ek_launch_chain_decide(const EkMaxHdr *hdrs_all, int n_shards, double cutoff,
                            EkPlan *plan, EkHist *hist, EkCtl *ctl, hipStream_t s)
{
    hipLaunchKernelGGL(ek_chain_decide_kernel, dim3(1), dim3(EK_WAVE), 0, s,
                       hdrs_all, n_shards, cutoff, plan, hist, ctl);
}

__global__ void __launch_bounds__(EK_CHAIN_THREADS)
ek_chain_decide_local_kernel(const EkBlockMax *__restrict__ blockmax,
                             const EkBlockMax *__restrict__ pm, int nb, int nbp,
                             int64_t global_offset, double cutoff,
                             EkPlan *__restrict__ plan, EkHist *__restrict__ hist,
                             EkCtl *__restrict__ ctl)
{
    __shared__ float sv[EK_MAX_CANDS];
    __shared__ uint32_t si[EK_MAX_CANDS];
    if (!plan->go) {
        if (threadIdx.x == 0)
            plan->napply = 0;
        return;
    }
    const int cn = plan->chain_n;
    ek_chain_reduce(blockmax, pm, nb, nbp, cn, sv, si);
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    float v[EK_MAX_CANDS];
    long long g[EK_MAX_CANDS];
    bool ok[EK_MAX_CANDS];
    for (int k = 0; k < EK_MAX_CANDS; ++k) {
        ok[k] = k < cn && si[k] != 0xffffffffu;
        v[k] = ok[k] ? sv[k] : 0.f;
        g[k] = ok[k] ? global_offset + (long long)si[k] : -1;
    }
    ek_chain_walk(v, g, ok, cutoff, plan, hist, ctl);
}

void ek_launch_chain_decide_local(const EkBlockMax *blockmax, const EkBlockMax *pm,
                                  int nb, int nbp, int64_t global_offset,
                                  double cutoff, EkPlan *plan, EkHist *hist,
                                  EkCtl *ctl, hipStream_t s)
{
    hipLaunchKernelGGL(ek_chain_decide_local_kernel, dim3(1),
                       dim3(EK_CHAIN_THREADS), 0, s, blockmax, pm, nb, nbp,
                       global_offset, cutoff, plan, hist, ctl);
}

// ---- 4. apply the accepted prefix -----------------------------------------------------
__global__ void __launch_bounds__(EK_BLOCK)
ek_chain_apply_kernel(const float *__restrict__ vecs, int64_t n, int64_t n_pad,
                      float *__restrict__ dist, int32_t *__restrict__ assign,
                      const EkPlan *__restrict__ plan,
                      EkBlockMax *__restrict__ blockmax)
{
    __shared__ float red_v[EK_BLOCK / EK_WAVE];
    __shared__ uint32_t red_i[EK_BLOCK / EK_WAVE];
    const int na = plan->napply;
    if (na < 1)
        return;                 // blockmax still describes the state
    const int label0 = plan->chain_label0;
    const int tid = threadIdx.x;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + tid;
    float v = -__builtin_inff();
    uint32_t i = 0xffffffffu;
    if (f < n) {
        float cur = dist[f];
        int32_t lab = -1;
        for (int k = 0; k < na; ++k) {      // kcenters.py:304-306, in order
            const float d = vecs[(size_t)(plan->chain[k] - 1) * n_pad + f];
            if (d < cur) {
                cur = d;
                lab = label0 + k;
            }
        }
        if (lab >= 0) {
            dist[f] = cur;
            assign[f] = lab;
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
        blockmax[blockIdx.x].val = v;
        blockmax[blockIdx.x].idx = i;
    }
}

void ek_launch_chain_apply(const float *vecs, int64_t n, int64_t n_pad, float *dist,
                           int32_t *assign, const EkPlan *plan,
                           EkBlockMax *blockmax, hipStream_t s)
{
    if (n <= 0)
        return;
    hipLaunchKernelGGL(ek_chain_apply_kernel,
                       dim3((unsigned)((n + EK_BLOCK - 1) / EK_BLOCK)),
                       dim3(EK_BLOCK), 0, s, vecs, n, n_pad, dist, assign, plan,
                       blockmax);
}
