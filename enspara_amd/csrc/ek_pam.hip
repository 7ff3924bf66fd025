// ek_pam.hip -- device side of one PAM (k-medoids) proposal.
//
// Replaces, for metric 'rmsd', the O(n) numpy passes of
// _kmedoids_pam_update (reference enspara/cluster/kmedoids.py:610-690):
//   state_inds = where(assignments == cid)            :611  -> count / select
//   new_ctr_dist = metric(X, proposed_center)         :637  -> ek_step_kernel<.,1>
//   dst_dn / dst_up_assig_other / dst_up_assig_this   :644-658 -> classify
//   assign_to_nearest_center(X[ambiguous], medoids)   :666  -> subset assign
//   cost = mean(dist^2) old vs new                    :478, :680-681 -> sumsq
// The host keeps only the RNG and the accept/reject decision.
#include "ek_common.h"
#include "ek_reduce.h"
#include <algorithm>
#include "ek_qcp.h"

// ---- medoid table: centred coordinates of chosen frames, center-major --------
__global__ void __launch_bounds__(EK_BLOCK)
ek_gather_frames_kernel(const float *__restrict__ tiles,
                        const double *__restrict__ G, int A,
                        const int64_t *__restrict__ idx, int first_row,
                        float *__restrict__ out_aos, double *__restrict__ outG)
{
    const int64_t f = idx[blockIdx.x];
    const int row = first_row + blockIdx.x;
    const float *p = tiles + (size_t)(f / EK_TILE) * 3 * (size_t)A * EK_TILE +
                     (f % EK_TILE);
    float *o = out_aos + (size_t)row * 3 * A;
    for (int r = threadIdx.x; r < 3 * A; r += EK_BLOCK)
        o[r] = p[(size_t)r * EK_TILE];
    if (threadIdx.x == 0)
        outG[row] = G[f];
}

void ek_launch_gather_frames(const float *tiles, const double *G, int A,
                             const int64_t *idx_dev, int count, int first_row,
                             float *out_aos, double *outG, hipStream_t s)
{
    if (count <= 0)
        return;
    hipLaunchKernelGGL(ek_gather_frames_kernel, dim3(count), dim3(EK_BLOCK), 0,
                       s, tiles, G, A, idx_dev, first_row, out_aos, outG);
}

// same, for arbitrary destination rows (rows[i] of out_aos / outG)
__global__ void __launch_bounds__(EK_BLOCK)
ek_gather_rows_kernel(const float *__restrict__ tiles,
                      const double *__restrict__ G, int A,
                      const int64_t *__restrict__ idx,
                      const int64_t *__restrict__ rows,
                      float *__restrict__ out_aos, double *__restrict__ outG)
{
    const int64_t f = idx[blockIdx.x];
    const int64_t row = rows[blockIdx.x];
    const float *p = tiles + (size_t)(f / EK_TILE) * 3 * (size_t)A * EK_TILE +
                     (f % EK_TILE);
    float *o = out_aos + (size_t)row * 3 * A;
    for (int r = threadIdx.x; r < 3 * A; r += EK_BLOCK)
        o[r] = p[(size_t)r * EK_TILE];
    if (threadIdx.x == 0)
        outG[row] = G[f];
}

void ek_launch_gather_rows(const float *tiles, const double *G, int A,
                           const int64_t *idx_dev, const int64_t *rows_dev,
                           int count, float *out_aos, double *outG, hipStream_t s)
{
    if (count <= 0)
        return;
    hipLaunchKernelGGL(ek_gather_rows_kernel, dim3(count), dim3(EK_BLOCK), 0, s,
                       tiles, G, A, idx_dev, rows_dev, out_aos, outG);
}

// ---- members of one cluster, in ascending frame order --------------------------
__global__ void __launch_bounds__(EK_BLOCK)
ek_count_members_kernel(const int32_t *__restrict__ assign, int64_t n,
                        int32_t cid, int32_t *__restrict__ blockcnt)
{
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    const int hit = (f < n && assign[f] == cid) ? 1 : 0;
    const int c = __syncthreads_count(hit);
    if (threadIdx.x == 0)
        blockcnt[blockIdx.x] = c;
}

// exclusive scan of the per-block counts by one workgroup; total -> *total
__global__ void __launch_bounds__(1024)
ek_scan_counts_kernel(const int32_t *__restrict__ blockcnt, int nblocks,
                      int64_t *__restrict__ scan, int64_t *__restrict__ total)
{
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int lo = t * per, hi = min(nblocks, lo + per);
    int64_t s = 0;
    for (int b = lo; b < hi; ++b)
        s += blockcnt[b];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int64_t run = 0;
        for (int i = 0; i < 1024; ++i) {
            const int64_t v = part[i];
            part[i] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    int64_t run = part[t];
    for (int b = lo; b < hi; ++b) {
        scan[b] = run;
        run += blockcnt[b];
    }
}

// the j-th member (0-based, ascending frame index) -> *out
__global__ void __launch_bounds__(EK_WAVE)
ek_select_member_kernel(const int32_t *__restrict__ assign, int64_t n,
                        int32_t cid, const int64_t *__restrict__ scan,
                        int nblocks, int64_t j, int64_t *__restrict__ out)
{
    if (threadIdx.x != 0)
        return;
    int lo = 0, hi = nblocks - 1;     // last block whose scan <= j
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (scan[mid] <= j)
            lo = mid;
        else
            hi = mid - 1;
    }
    int64_t rank = j - scan[lo];
    int64_t res = -1;
    const int64_t f0 = (int64_t)lo * EK_BLOCK;
    for (int i = 0; i < EK_BLOCK; ++i) {
        const int64_t f = f0 + i;
        if (f < n && assign[f] == cid) {
            if (rank == 0) {
                res = f;
                break;
            }
            --rank;
        }
    }
    *out = res;
}

void ek_launch_count_members(const int32_t *assign, int64_t n, int32_t cid,
                             int32_t *blockcnt, int64_t *scan, int64_t *total,
                             hipStream_t s)
{
    const int nblocks = (int)((n + EK_BLOCK - 1) / EK_BLOCK);
    if (nblocks > 0)            // an empty shard: the scan alone reports 0 members
        hipLaunchKernelGGL(ek_count_members_kernel, dim3(nblocks), dim3(EK_BLOCK),
                           0, s, assign, n, cid, blockcnt);
    hipLaunchKernelGGL(ek_scan_counts_kernel, dim3(1), dim3(1024), 0, s,
                       blockcnt, nblocks, scan, total);
}

void ek_launch_select_member(const int32_t *assign, int64_t n, int32_t cid,
                             const int64_t *scan, int64_t j, int64_t *out,
                             hipStream_t s)
{
    const int nblocks = (int)((n + EK_BLOCK - 1) / EK_BLOCK);
    hipLaunchKernelGGL(ek_select_member_kernel, dim3(1), dim3(EK_WAVE), 0, s,
                       assign, n, cid, scan, nblocks, j, out);
}

// the same for `count` consecutive clusters cid0.. at once (proposal windows):
// blockcnt[j][workgroup], scan[j][workgroup], total[j]
__global__ void __launch_bounds__(EK_BLOCK)
ek_count_members_multi_kernel(const int32_t *__restrict__ assign, int64_t n,
                              int32_t cid0, int count, int nblocks,
                              int32_t *__restrict__ blockcnt)
{
    __shared__ int cnt[EK_PAM_WIN];
    if (threadIdx.x < EK_PAM_WIN)
        cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (f < n) {
        const int j = assign[f] - cid0;
        if (j >= 0 && j < count)
            atomicAdd(&cnt[j], 1);
    }
    __syncthreads();
    if ((int)threadIdx.x < count)
        blockcnt[(size_t)threadIdx.x * nblocks + blockIdx.x] = cnt[threadIdx.x];
}

__global__ void __launch_bounds__(1024)
ek_scan_counts_multi_kernel(const int32_t *__restrict__ blockcnt, int nblocks,
                            int64_t *__restrict__ scan,
                            int64_t *__restrict__ total, int64_t *total_host)
{
    __shared__ int64_t part[1024];
    const int j = blockIdx.x;
    const int32_t *bc = blockcnt + (size_t)j * nblocks;
    int64_t *sc = scan + (size_t)j * nblocks;
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int lo = t * per, hi = min(nblocks, lo + per);
    int64_t s = 0;
    for (int b = lo; b < hi; ++b)
        s += bc[b];
    part[t] = s;
    __syncthreads();
    // exclusive scan of the 1024 partial sums by the first wave
    if (t < EK_WAVE) {
        int64_t loc[1024 / EK_WAVE];
        int64_t sum = 0;
#pragma unroll
        for (int q = 0; q < 1024 / EK_WAVE; ++q) {
            loc[q] = sum;
            sum += part[t * (1024 / EK_WAVE) + q];
        }
        int64_t incl = sum;               // inclusive scan over the lanes
#pragma unroll
        for (int off = 1; off < EK_WAVE; off <<= 1) {
            const int64_t o = __shfl_up(incl, off, 64);
            if (t >= off)
                incl += o;
        }
        const int64_t excl = incl - sum;
#pragma unroll
        for (int q = 0; q < 1024 / EK_WAVE; ++q)
            part[t * (1024 / EK_WAVE) + q] = excl + loc[q];
        if (t == EK_WAVE - 1) {
            total[j] = incl;
            if (total_host)     // (mapped host memory: no copy kernel behind this one)
                total_host[j] = incl;
        }
    }
    __syncthreads();
    int64_t run = part[t];
    for (int b = lo; b < hi; ++b) {
        sc[b] = run;
        run += bc[b];
    }
}

// the js[j]-th member of cluster cid0 + j -> out[j] (-1: none; js[j] < 0: skipped)
__global__ void __launch_bounds__(EK_BLOCK)
ek_select_member_multi_kernel(const int32_t *__restrict__ assign, int64_t n,
                              int32_t cid0, const int64_t *__restrict__ scan,
                              int nblocks, const int64_t *__restrict__ js,
                              int64_t *__restrict__ out, int64_t *out_host)
{
    __shared__ int lo_s;
    __shared__ long long f_s;
    __shared__ int wcnt[EK_BLOCK / EK_WAVE];
    const int j = blockIdx.x;
    const int64_t want = js[j];
    if (want < 0) {
        if (threadIdx.x == 0) {
            out[j] = -1;
            if (out_host)
                out_host[j] = -1;
        }
        return;
    }
    const int32_t cid = cid0 + j;
    const int64_t *sc = scan + (size_t)j * nblocks;
    if (threadIdx.x == 0) {
        int lo = 0, hi = nblocks - 1;     // last workgroup whose scan <= want
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            if (sc[mid] <= want)
                lo = mid;
            else
                hi = mid - 1;
        }
        lo_s = lo;
        f_s = -1;
    }
    __syncthreads();
    const int lo = lo_s;
    const int64_t rank = want - sc[lo];
    const int64_t f = (int64_t)lo * EK_BLOCK + threadIdx.x;
    const bool hit = f < n && assign[f] == cid;
    const unsigned long long m = __ballot(hit);
    const int lane = threadIdx.x & (EK_WAVE - 1), wv = threadIdx.x / EK_WAVE;
    if (lane == 0)
        wcnt[wv] = __popcll(m);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wv; ++w)
        before += wcnt[w];
    if (hit && before + __popcll(m & ((1ull << lane) - 1ull)) == rank)
        f_s = f;
    __syncthreads();
    if (threadIdx.x == 0) {     // (one write each: the host's copy is mapped host memory)
        out[j] = f_s;
        if (out_host)
            out_host[j] = f_s;
    }
}

void ek_launch_count_members_multi(const int32_t *assign, int64_t n, int32_t cid0,
                                   int count, int32_t *blockcnt, int64_t *scan,
                                   int64_t *total, hipStream_t s, int64_t *total_host)
{
    const int nblocks = (int)((n + EK_BLOCK - 1) / EK_BLOCK);
    if (nblocks > 0)
        hipLaunchKernelGGL(ek_count_members_multi_kernel, dim3(nblocks),
                           dim3(EK_BLOCK), 0, s, assign, n, cid0, count, nblocks,
                           blockcnt);
    hipLaunchKernelGGL(ek_scan_counts_multi_kernel, dim3(count), dim3(1024), 0, s,
                       blockcnt, nblocks, scan, total, total_host);
}

void ek_launch_count_members_multi(const int32_t *assign, int64_t n, int32_t cid0,
                                   int count, int32_t *blockcnt, int64_t *scan,
                                   int64_t *total, hipStream_t s)
{
    ek_launch_count_members_multi(assign, n, cid0, count, blockcnt, scan, total, s,
                                  (int64_t *)nullptr);
}

// the scan alone (the per-workgroup counts were written by another kernel)
void ek_launch_scan_counts(const int32_t *blockcnt, int64_t n, int64_t *scan,
                           int64_t *total, hipStream_t s)
{
    const int nblocks = (int)((n + EK_BLOCK - 1) / EK_BLOCK);
    hipLaunchKernelGGL(ek_scan_counts_multi_kernel, dim3(1), dim3(1024), 0, s,
                       blockcnt, nblocks, scan, total, (int64_t *)nullptr);
}

void ek_launch_select_member_multi(const int32_t *assign, int64_t n, int32_t cid0,
                                   int count, const int64_t *scan,
                                   const int64_t *js_dev, int64_t *out,
                                   hipStream_t s, int64_t *out_host)
{
    const int nblocks = (int)((n + EK_BLOCK - 1) / EK_BLOCK);
    hipLaunchKernelGGL(ek_select_member_multi_kernel, dim3(count), dim3(EK_BLOCK),
                       0, s, assign, n, cid0, scan, nblocks, js_dev, out, out_host);
}

void ek_launch_select_member_multi(const int32_t *assign, int64_t n, int32_t cid0,
                                   int count, const int64_t *scan,
                                   const int64_t *js_dev, int64_t *out,
                                   hipStream_t s)
{
    ek_launch_select_member_multi(assign, n, cid0, count, scan, js_dev, out, s,
                                  (int64_t *)nullptr);
}

// ---- classification (kmedoids.py:639-658) ---------------------------------------
// newd: distance of every frame to the proposed medoid.
//   dist > newd                      -> (newd, cid)
//   dist <= newd and assign != cid   -> unchanged
//   dist <= newd and assign == cid   -> ambiguous: listed, resolved below
// MARK: an ambiguous member's trial label is -2 - (its position in the list)
// until the cost-sum kernel resolves it from amb_best (no scatter launch)
template <bool MARK>
__global__ void __launch_bounds__(EK_BLOCK)
ek_pam_classify_kernel(const float *__restrict__ dist,
                       const int32_t *__restrict__ assign,
                       const float *__restrict__ newd, int64_t n, int32_t cid,
                       float *__restrict__ ndist, int32_t *__restrict__ nassign,
                       uint32_t *__restrict__ amb,
                       unsigned long long *__restrict__ amb_best,
                       unsigned int *__restrict__ amb_count,
                       unsigned int *__restrict__ reach)
{
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (f >= n)
        return;
    const float d = dist[f], nd = newd[f];
    const int32_t a = assign[f];
    if (d > nd) {
        ndist[f] = nd;
        nassign[f] = cid;
    } else if (a != cid) {
        ndist[f] = d;
        nassign[f] = a;
    } else {
        const unsigned int pos = atomicAdd(amb_count, 1u);
        amb[pos] = (uint32_t)f;
        amb_best[pos] = ~0ull;
        if (MARK) {
            ndist[f] = 0.f;
            nassign[f] = -2 - (int32_t)pos;
        }
        // how far a medoid may be from the old one and still matter to this
        // frame (ek_pam_prune_kernel); non-negative floats order like their bits
        atomicMax(reach, __float_as_uint(d + nd));
    }
}

void ek_launch_pam_classify(const float *dist, const int32_t *assign,
                            const float *newd, int64_t n, int32_t cid,
                            float *ndist, int32_t *nassign, uint32_t *amb,
                            unsigned long long *amb_best,
                            unsigned int *amb_count, unsigned int *reach,
                            hipStream_t s, int mark)
{
    const int nblocks = (int)((n + EK_BLOCK - 1) / EK_BLOCK);
    if (nblocks <= 0)
        return;
    if (mark)
        hipLaunchKernelGGL(ek_pam_classify_kernel<true>, dim3(nblocks),
                           dim3(EK_BLOCK), 0, s, dist, assign, newd, n, cid, ndist,
                           nassign, amb, amb_best, amb_count, reach);
    else
        hipLaunchKernelGGL(ek_pam_classify_kernel<false>, dim3(nblocks),
                           dim3(EK_BLOCK), 0, s, dist, assign, newd, n, cid, ndist,
                           nassign, amb, amb_best, amb_count, reach);
}

// The classification inside a window (ek_pam_window_run), one launch for
//   * taking over an accepted predecessor's trial state (APPLY above),
//   * the classification itself, 4096 frames per workgroup of 1024 threads,
//   * the compacted copy of every ambiguous member's coordinates (ambt / ambG,
//     what ek_gather_amb_kernel does in a launch of its own): the wave that
//     finds one copies it, the frame-major copy of the shard makes that 12 A
//     contiguous bytes,
//   * and, by the last workgroup to finish (arrival counter; `reach` and the
//     list of ambiguous members travel as atomics / coherent stores,
//     ek_reduce.h), the list of medoids within reach of those members
//     (ek_pam_prune_kernel's test) from the window's distance tables: slot
//     `slot`'s old medoid against medoid c is O[slot][c] -- or, where an earlier
//     slot i of this window was accepted, T[i][cid]: its proposal sits in row
//     cid0 + i now and the old medoid of `cid` has not moved since the tables
//     were made.
#define EK_CLS_THREADS 1024                 // also the width of the last workgroup's tail
#define EK_CLS_FPT 4                        // consecutive frames per load
#define EK_CLS_TRIPS 1
#define EK_CLS_WG (EK_CLS_THREADS * EK_CLS_FPT * EK_CLS_TRIPS)

__global__ void __launch_bounds__(EK_CLS_THREADS)
ek_pam_classify_window_kernel(float *dist, int32_t *assign,
                              const float *__restrict__ newd, int64_t n,
                              int32_t cid, float *ndist, int32_t *nassign,
                              uint32_t *__restrict__ amb,
                              unsigned long long *__restrict__ amb_best,
                              unsigned int *__restrict__ amb_count, EkPamClsWin w)
{
    const int t = threadIdx.x;
    const bool ap = *w.prev_accept != 0;
    const int64_t base = (int64_t)blockIdx.x * EK_CLS_WG + (int64_t)t * EK_CLS_FPT;
    float d[EK_CLS_TRIPS][EK_CLS_FPT], nd[EK_CLS_TRIPS][EK_CLS_FPT];
    int32_t a[EK_CLS_TRIPS][EK_CLS_FPT];
#pragma unroll
    for (int q = 0; q < EK_CLS_TRIPS; ++q) {
        const int64_t f0 = base + (int64_t)q * EK_CLS_THREADS * EK_CLS_FPT;
        const float *ds = ap ? ndist : dist;
        const int32_t *as = ap ? nassign : assign;
        if (f0 + EK_CLS_FPT <= n) {
            const float4 dv = *(const float4 *)(ds + f0);
            const int4 av = *(const int4 *)(as + f0);
            const float4 nv = *(const float4 *)(newd + f0);
            d[q][0] = dv.x; d[q][1] = dv.y; d[q][2] = dv.z; d[q][3] = dv.w;
            a[q][0] = av.x; a[q][1] = av.y; a[q][2] = av.z; a[q][3] = av.w;
            nd[q][0] = nv.x; nd[q][1] = nv.y; nd[q][2] = nv.z; nd[q][3] = nv.w;
        } else {
#pragma unroll
            for (int k = 0; k < EK_CLS_FPT; ++k) {
                const bool in = f0 + k < n;
                d[q][k] = in ? ds[f0 + k] : 0.f;
                a[q][k] = in ? as[f0 + k] : -1;
                nd[q][k] = in ? newd[f0 + k] : 0.f;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < EK_CLS_TRIPS; ++q) {
        const int64_t f0 = base + (int64_t)q * EK_CLS_THREADS * EK_CLS_FPT;
        float od[EK_CLS_FPT];
        int32_t oa[EK_CLS_FPT];
        unsigned int ambk = 0;                  // which of the four are ambiguous
#pragma unroll
        for (int k = 0; k < EK_CLS_FPT; ++k) {
            if (d[q][k] > nd[q][k]) {
                od[k] = nd[q][k];
                oa[k] = cid;
            } else {
                od[k] = d[q][k];
                oa[k] = a[q][k];
                if (a[q][k] == cid && f0 + k < n)
                    ambk |= 1u << k;
            }
        }
        // ambiguous members (rare): listed, marked in the trial state until
        // the cost-sum launch resolves them, their coordinates compacted
        while (__ballot(ambk != 0)) {
            int k = -1;
            unsigned int pos = 0;
            if (ambk) {
                k = __ffs((int)ambk) - 1;
                ambk &= ambk - 1;
                pos = atomicAdd(amb_count, 1u);
                __hip_atomic_store(&amb[pos], (uint32_t)(f0 + k), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                amb_best[pos] = ~0ull;
                // how far a medoid may be from the old one and still matter to
                // this frame; non-negative floats order like their bits
                float dk = d[q][0], nk = nd[q][0];
#pragma unroll
                for (int kk = 1; kk < EK_CLS_FPT; ++kk)
                    if (k == kk) {
                        dk = d[q][kk];
                        nk = nd[q][kk];
                    }
                atomicMax(amb_count + 1, __float_as_uint(dk + nk));
#pragma unroll
                for (int kk = 0; kk < EK_CLS_FPT; ++kk)
                    if (k == kk) {
                        od[kk] = 0.f;
                        oa[kk] = -2 - (int32_t)pos;
                    }
            }
            unsigned long long todo = __ballot(k >= 0);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int64_t f = __shfl(f0 + k, src, EK_WAVE);
                const unsigned int ps = __shfl(pos, src, EK_WAVE);
                if ((int64_t)ps < w.cap) {
                    const float *p = w.frames_aos + (size_t)f * 3 * w.A;
                    const int lane = t & (EK_WAVE - 1);
                    for (int r0 = lane; r0 < 3 * w.A; r0 += 8 * EK_WAVE) {
                        float v[8];             // eight loads in flight
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            v[u] = (r0 + u * EK_WAVE < 3 * w.A) ? p[r0 + u * EK_WAVE]
                                                                : 0.f;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (r0 + u * EK_WAVE < 3 * w.A)
                                w.ambt[(size_t)(r0 + u * EK_WAVE) * w.cap + ps] = v[u];
                    }
                    if (lane == 0)
                        w.ambG[ps] = w.G[f];
                }
            }
        }
        if (f0 + EK_CLS_FPT <= n) {
            if (ap) {
                *(float4 *)(dist + f0) =
                    make_float4(d[q][0], d[q][1], d[q][2], d[q][3]);
                *(int4 *)(assign + f0) = make_int4(a[q][0], a[q][1], a[q][2], a[q][3]);
            }
            *(float4 *)(ndist + f0) = make_float4(od[0], od[1], od[2], od[3]);
            *(int4 *)(nassign + f0) = make_int4(oa[0], oa[1], oa[2], oa[3]);
        } else {
#pragma unroll
            for (int k = 0; k < EK_CLS_FPT; ++k)
                if (f0 + k < n) {
                    if (ap) {
                        dist[f0 + k] = d[q][k];
                        assign[f0 + k] = a[q][k];
                    }
                    ndist[f0 + k] = od[k];
                    nassign[f0 + k] = oa[k];
                }
        }
    }
    if (!w.O)
        return;
    if (!ek_arrive_last_tree(w.tick, w.tick + 1))
        return;
    // ---- the last workgroup: medoids within reach of the ambiguous members --------
    const float R = __uint_as_float(__hip_atomic_load(amb_count + 1, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT));
    const float lim = R * 1.001f + 1e-3f;
    for (int c0 = t; c0 < w.K; c0 += 8 * EK_CLS_THREADS) {
        float D[8];                             // eight table reads in flight
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * EK_CLS_THREADS;
            D[u] = (c < w.K) ? w.O[c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * EK_CLS_THREADS;
            if (c >= w.K)
                continue;
            const int i = c - w.cid0;
            if (i >= 0 && i < w.slot && w.accepted[i])
                D[u] = w.T[(size_t)i * w.K + cid];
            if (c == cid || !(D[u] > lim))
                w.list[atomicAdd(amb_count + 2, 1u)] = c;
        }
    }
}

void ek_launch_pam_classify_window(float *dist, int32_t *assign, const float *newd,
                                   int64_t n, int32_t cid, float *ndist,
                                   int32_t *nassign, uint32_t *amb,
                                   unsigned long long *amb_best,
                                   unsigned int *amb_count, const EkPamClsWin &w,
                                   hipStream_t s)
{
    const int nblocks = (int)((n + EK_CLS_WG - 1) / EK_CLS_WG);
    if (nblocks <= 0)
        return;
    hipLaunchKernelGGL(ek_pam_classify_window_kernel, dim3(nblocks),
                       dim3(EK_CLS_THREADS),
                       0, s, dist, assign, newd, n, cid, ndist, nassign, amb,
                       amb_best, amb_count, w);
}

__global__ void __launch_bounds__(EK_BLOCK)
ek_pam_apply_kernel(const int32_t *__restrict__ flag, float *__restrict__ dist,
                    const float *__restrict__ ndist, int32_t *__restrict__ assign,
                    const int32_t *__restrict__ nassign, int64_t n)
{
    if (!*flag)
        return;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (f < n) {
        dist[f] = ndist[f];
        assign[f] = nassign[f];
    }
}

void ek_launch_pam_apply(const int32_t *flag, float *dist, const float *ndist,
                         int32_t *assign, const int32_t *nassign, int64_t n,
                         hipStream_t s)
{
    const int nblocks = (int)((n + EK_BLOCK - 1) / EK_BLOCK);
    if (nblocks <= 0)
        return;
    hipLaunchKernelGGL(ek_pam_apply_kernel, dim3(nblocks), dim3(EK_BLOCK), 0, s,
                       flag, dist, ndist, assign, nassign, n);
}

// ---- which medoids can matter to the ambiguous members --------------------------------
// An ambiguous member f of cluster cid sat at distance d_old(f) from the old
// medoid and sits at nd(f) from the proposal, which is in the trial set, so its
// new nearest medoid is at most nd(f) away.  Minimal RMSD is a metric, hence a
// medoid c with  D(old medoid, c) - d_old(f) > nd(f)  for every such f -- i.e.
// D(old, c) > reach = max_f (d_old(f) + nd(f)) -- is farther from every one of
// them than the proposal and cannot be anybody's nearest: the list-all-medoids
// search of kmedoids.py:666 returns the same labels and distances without it.
// (The comparison carries a margin four orders of magnitude above the rounding
// error of a distance; D itself is only compared, so its summation order is
// free: one wave per medoid, lanes strided over the atoms.)  Row K of the table
// holds the old medoid, row cid the proposal (always kept).
__global__ void __launch_bounds__(EK_BLOCK)
ek_pam_prune_kernel(const float *__restrict__ aos, const double *__restrict__ Gm,
                    int A, int K, int cid, const unsigned int *__restrict__ reach,
                    int32_t *__restrict__ list, unsigned int *__restrict__ n_list)
{
    const int lane = threadIdx.x & (EK_WAVE - 1);
    const int c = blockIdx.x * (EK_BLOCK / EK_WAVE) + threadIdx.x / EK_WAVE;
    if (c >= K)
        return;
    bool keep = (c == cid);
    if (!keep) {
        const float *x = aos + (size_t)c * 3 * A;
        const float *y = aos + (size_t)K * 3 * A;
        float S[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int a = lane; a < A; a += EK_WAVE) {
            const float x0 = x[3 * a], x1 = x[3 * a + 1], x2 = x[3 * a + 2];
            const float y0 = y[3 * a], y1 = y[3 * a + 1], y2 = y[3 * a + 2];
            S[0] += x0 * y0; S[1] += x0 * y1; S[2] += x0 * y2;
            S[3] += x1 * y0; S[4] += x1 * y1; S[5] += x1 * y2;
            S[6] += x2 * y0; S[7] += x2 * y1; S[8] += x2 * y2;
        }
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1)
                S[j] += __shfl_xor(S[j], off, 64);
        const float D = ek_rmsd_from_S(S, Gm[c], Gm[K], A);
        const float R = __uint_as_float(*reach);
        keep = !(D > R * 1.001f + 1e-3f);
    }
    if (keep && lane == 0)
        list[atomicAdd(n_list, 1u)] = c;
}

void ek_launch_pam_prune(const float *aos, const double *Gm, int A, int K, int cid,
                         const unsigned int *reach, int32_t *list,
                         unsigned int *n_list, hipStream_t s)
{
    const int per = EK_BLOCK / EK_WAVE;
    hipLaunchKernelGGL(ek_pam_prune_kernel, dim3((K + per - 1) / per),
                       dim3(EK_BLOCK), 0, s, aos, Gm, A, K, cid, reach, list,
                       n_list);
}

// ---- listed frames x all medoids (kmedoids.py:666) ---------------------------------
// grid = (frame chunks of 256, center chunks of PCT).  Every (frame, center)
// distance is folded into amb_best[i] with a 64-bit atomic min on
// (float bits << 32 | center index): distances are >= 0 so the bit pattern is
// monotonic, and equal distances resolve to the lowest center index -- the
// result of util.py:199-203's strict-< scan in ascending center order,
// independent of the order the chunks finish in.
#define PCT 8

// compact the listed frames into a frame-minor block ambt[3A][cap] (+ traces)
// so that the pair kernel below reads them coalesced
__global__ void __launch_bounds__(EK_BLOCK)
ek_gather_amb_kernel(const float *__restrict__ tiles,
                     const double *__restrict__ G, int A,
                     const uint32_t *__restrict__ amb,
                     const unsigned int *__restrict__ n_amb_p, int64_t cap,
                     const unsigned int *__restrict__ n_list,
                     float *__restrict__ ambt, double *__restrict__ ambG)
{
    // one workgroup per listed frame: its 3A row loads go out in parallel
    const unsigned int i = blockIdx.x;
    if (i >= *n_amb_p)
        return;
    if (n_list && *n_list == 1)
        return;                 // ek_subset_assign_kernel will not need them
    const uint32_t f = amb[i];
    const float *p = tiles + (size_t)(f / EK_TILE) * 3 * (size_t)A * EK_TILE +
                     (f % EK_TILE);
    for (int r = threadIdx.x; r < 3 * A; r += EK_BLOCK)
        ambt[(size_t)r * cap + i] = p[(size_t)r * EK_TILE];
    if (threadIdx.x == 0)
        ambG[i] = G[f];
}

__global__ void __launch_bounds__(EK_BLOCK)
ek_subset_assign_kernel(const float *__restrict__ ambt,
                        const double *__restrict__ ambG, int A, int64_t cap,
                        const unsigned int *__restrict__ n_amb_p,
                        const float *__restrict__ centers,
                        const double *__restrict__ Gc, int K,
                        const int32_t *__restrict__ list,
                        const unsigned int *__restrict__ n_list,
                        const uint32_t *__restrict__ amb,
                        const float *__restrict__ newd, int cid,
                        unsigned long long *__restrict__ amb_best)
{
    extern __shared__ __attribute__((aligned(16))) float ctile[];
    __shared__ double gtile[PCT];
    __shared__ int32_t ktile[PCT];
    const int tid = threadIdx.x;
    const unsigned int n_amb = *n_amb_p;
    if (blockIdx.x * EK_BLOCK >= n_amb)
        return;                            // whole workgroup past the list
    const unsigned int i = blockIdx.x * EK_BLOCK + tid;
    // the medoids to try: all K of them, or those ek_pam_prune_kernel listed
    const int Kl = list ? (int)*n_list : K;
    const int k0 = blockIdx.y * PCT;
    if (k0 >= Kl)
        return;
    if (list && Kl == 1) {
        // only the proposal itself is within reach (the usual case): the
        // distance to it is the one the proposal's pass already computed, by the
        // same FMA chain from the same coordinates
        if (i < n_amb)
            atomicMin(&amb_best[i],
                      ((unsigned long long)__float_as_uint(newd[amb[i]]) << 32) |
                          (unsigned int)cid);
        return;
    }
    const int kc = (Kl - k0 < PCT) ? (Kl - k0) : PCT;
    if (tid < PCT)
        ktile[tid] = (tid < kc) ? (list ? list[k0 + tid] : k0 + tid) : 0;
    __syncthreads();
    // ctile[a][c][k]; global reads run along each center's row (coalesced)
    for (int j = tid; j < 3 * A * PCT; j += EK_BLOCK) {
        const int c = j / (3 * A), r = j % (3 * A);
        // centers in pairs, [atom][pair][xyz][2]: one packed FMA serves two
        ctile[(r / 3) * (3 * PCT) + (c / 2) * 6 + (r % 3) * 2 + (c & 1)] =
            (c < kc) ? centers[(size_t)ktile[c] * 3 * A + r] : 0.f;
    }
    if (tid < PCT)
        gtile[tid] = (tid < kc) ? Gc[ktile[tid]] : 0.0;
    __syncthreads();
    if (i >= n_amb)
        return;
    const float *p = ambt + i;
    ek_v2f s2[PCT / 2][9];      // (center 2p, center 2p+1): independent FMA chains
#pragma unroll
    for (int c = 0; c < PCT / 2; ++c)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            s2[c][j] = (ek_v2f){0.f, 0.f};
    const float4 *ct4 = (const float4 *)ctile;
#pragma unroll 4
    for (int a = 0; a < A; ++a) {
        const float xs = p[(size_t)(3 * a + 0) * cap];
        const float ys = p[(size_t)(3 * a + 1) * cap];
        const float zs = p[(size_t)(3 * a + 2) * cap];
        float cc[3 * PCT];
#pragma unroll
        for (int q = 0; q < 3 * PCT / 4; ++q) {
            const float4 v = ct4[a * (3 * PCT / 4) + q];
            cc[4 * q + 0] = v.x;
            cc[4 * q + 1] = v.y;
            cc[4 * q + 2] = v.z;
            cc[4 * q + 3] = v.w;
        }
        const ek_v2f x = (ek_v2f){xs, xs}, y = (ek_v2f){ys, ys},
                     z = (ek_v2f){zs, zs};
#pragma unroll
        for (int c = 0; c < PCT / 2; ++c) {
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
    const double Gf = ambG[i];
    unsigned long long best = ~0ull;
#pragma unroll
    for (int c = 0; c < PCT; ++c) {
        if (c < kc) {
            float S[9];
#pragma unroll
            for (int j = 0; j < 9; ++j)
                S[j] = s2[c / 2][j][c & 1];
            const float d = ek_rmsd_from_S(S, Gf, gtile[c], A);
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(d) << 32) |
                (unsigned int)ktile[c];
            if (key < best)
                best = key;
        }
    }
    atomicMin(&amb_best[i], best);
}

void ek_launch_subset_assign(const float *tiles, const double *G, int A,
                             const uint32_t *amb, const unsigned int *n_amb,
                             int64_t max_amb, float *ambt, double *ambG,
                             int64_t cap, const float *centers,
                             const double *Gc, int K, const int32_t *list,
                             const unsigned int *n_list, const float *newd,
                             int cid, unsigned long long *amb_best,
                             hipStream_t s, bool gathered)
{
    if (max_amb <= 0 || K <= 0)
        return;
    const unsigned gx = (unsigned)((max_amb + EK_BLOCK - 1) / EK_BLOCK);
    if (!gathered)      // (the window's classification compacts them itself)
        hipLaunchKernelGGL(ek_gather_amb_kernel, dim3((unsigned)max_amb),
                           dim3(EK_BLOCK), 0, s, tiles, G, A, amb, n_amb, cap,
                           list ? n_list : nullptr, ambt, ambG);
    const dim3 grid(gx, (K + PCT - 1) / PCT);
    const size_t lds = (size_t)3 * A * PCT * sizeof(float);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)ek_subset_assign_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL(ek_subset_assign_kernel, grid, dim3(EK_BLOCK), lds, s,
                       ambt, ambG, A, cap, n_amb, centers, Gc, K, list, n_list,
                       amb, newd, cid, amb_best);
}

__global__ void __launch_bounds__(EK_BLOCK)
ek_pam_scatter_kernel(const uint32_t *__restrict__ amb,
                      const unsigned long long *__restrict__ amb_best,
                      const unsigned int *__restrict__ n_amb_p,
                      float *__restrict__ ndist, int32_t *__restrict__ nassign)
{
    const unsigned int i = blockIdx.x * EK_BLOCK + threadIdx.x;
    if (i >= *n_amb_p)
        return;
    const unsigned long long key = amb_best[i];
    const uint32_t f = amb[i];
    ndist[f] = __uint_as_float((unsigned int)(key >> 32));
    nassign[f] = (int32_t)(key & 0xffffffffu);
}

void ek_launch_pam_scatter(const uint32_t *amb,
                           const unsigned long long *amb_best,
                           const unsigned int *n_amb, int64_t max_amb,
                           float *ndist, int32_t *nassign, hipStream_t s)
{
    if (max_amb <= 0)
        return;
    hipLaunchKernelGGL(ek_pam_scatter_kernel,
                       dim3((unsigned)((max_amb + EK_BLOCK - 1) / EK_BLOCK)),
                       dim3(EK_BLOCK), 0, s, amb, amb_best, n_amb, ndist,
                       nassign);
}

// ---- cost: sum of squares in float64, fixed reduction order -----------------------
// (kmedoids.py:478-479 takes np.square(x).mean() in float64; each square of a
// float32 is exact in float64, only the summation order differs from numpy's
// pairwise sum, by a few ulp of the total -- unless the order is numpy's own:
// see "cost sums in numpy's order" below)

// ---- the trial medoid table ---------------------------------------------------------------
// One workgroup prepares the trial medoid table and the counters of a proposal:
// (a) a row left modified by a rejected proposal is restored from row K,
// (b) row cid is saved in row K, (c) the proposal goes into row cid -- either a
// local frame (index by value, or read from idx_dev) or a center given as
// centred coordinates --, (d) the ambiguous-member counter and the moved-cluster
// mask are cleared.  Thread t owns elements t, t + 256, .. of every row, so the
// three steps need no barrier even when restore_cid == cid.
__global__ void __launch_bounds__(EK_BLOCK)
ek_pam_trial_kernel(const float *__restrict__ tiles, const double *__restrict__ G,
                    int A, float *__restrict__ aos, double *__restrict__ Gm, int K,
                    int cid, int restore_cid, int64_t frame_index,
                    const int64_t *__restrict__ idx_dev,
                    const float *__restrict__ ext_aos,
                    const double *__restrict__ ext_G,
                    unsigned int *__restrict__ amb_count,
                    unsigned int *__restrict__ moved, EkPamWin *__restrict__ win,
                    int win_slots)
{
    // amb_count[0] ambiguous members, [1] reach bits, [2] listed medoids
    const int tid = threadIdx.x;
    if (win) {          // a window starts here: nothing decided, `win_slots` to go
        static_assert(sizeof(EkPamWin) % 4 == 0 && sizeof(EkPamWin) / 4 <= 4 * EK_BLOCK,
                      "EkPamWin is cleared word by word");
        for (int i = tid; i < (int)(sizeof(EkPamWin) / 4); i += EK_BLOCK)
            ((uint32_t *)win)[i] = (i == 0) ? (uint32_t)win_slots : 0u;
    }
    const float *p = nullptr;
    int64_t f = -1;
    if (!ext_aos) {
        f = (frame_index >= 0) ? frame_index : idx_dev[0];
        p = tiles + (size_t)(f / EK_TILE) * 3 * (size_t)A * EK_TILE + (f % EK_TILE);
    }
    for (int r = tid; r < 3 * A; r += EK_BLOCK) {
        if (restore_cid >= 0)
            aos[(size_t)restore_cid * 3 * A + r] = aos[(size_t)K * 3 * A + r];
        aos[(size_t)K * 3 * A + r] = aos[(size_t)cid * 3 * A + r];
        aos[(size_t)cid * 3 * A + r] = ext_aos ? ext_aos[r] : p[(size_t)r * EK_TILE];
    }
    if (tid == 0) {
        if (restore_cid >= 0)
            Gm[restore_cid] = Gm[K];
        Gm[K] = Gm[cid];
        Gm[cid] = ext_aos ? ext_G[0] : G[f];
        amb_count[0] = 0;
        amb_count[1] = 0;
        amb_count[2] = 0;
        *moved = 0;
    }
}

void ek_launch_pam_trial(const float *tiles, const double *G, int A, float *aos,
                         double *Gm, int K, int cid, int restore_cid,
                         int64_t frame_index, const int64_t *idx_dev,
                         const float *ext_aos, const double *ext_G,
                         unsigned int *amb_count, unsigned int *moved,
                         hipStream_t s, EkPamWin *win, int win_slots)
{
    hipLaunchKernelGGL(ek_pam_trial_kernel, dim3(1), dim3(EK_BLOCK), 0, s, tiles, G,
                       A, aos, Gm, K, cid, restore_cid, frame_index, idx_dev,
                       ext_aos, ext_G, amb_count, moved, win, win_slots);
}

// ---- cost sums in numpy's order ------------------------------------------------------------
// The reference compares np.square(d).mean() of the old and the trial
// distances (kmedoids.py:478-479, :680-683).  When a proposal only permutes
// distances -- a two-member cluster swapping its medoid -- the two means are
// equal in exact arithmetic and the comparison is decided by the rounding of the
// summation, so the sum has to be taken in numpy's own order: the array is cut
// into buffer chunks of 8192 elements; each chunk is summed pairwise -- halves
// (the left one rounded down to a multiple of 8) down to leaves of <= 128
// elements, a leaf being eight interleaved running sums combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail -- and the chunk
// sums are added up left to right.  (numpy/_core/src/umath/loops_utils.h.src,
// pairwise sum; checked against numpy for lengths 1..1.3e6.)  The shapes of a
// full chunk and of the last, shorter one are tabulated by the host
// (ek_pw_build_shape) and shared by every launch.

// one 8-lane group per leaf; moved-cluster mask as a by-product of the same read
// RESOLVE: trial labels < -1 are ambiguous members still waiting for their
// result (ek_pam_classify_kernel<true>): it is read from amb_best here -- every
// frame is visited exactly once -- and written into the trial state
template <bool RESOLVE>
__device__ __forceinline__ void ek_pw_fetch(const float *a, float *b,
                                            const int32_t *assign, int32_t *nassign,
                                            const unsigned long long *amb_best,
                                            int64_t f, double &va, double &vb,
                                            int32_t &oa, int32_t &na)
{
    va = a[f];
    float fb = b[f];
    oa = assign[f];
    na = nassign[f];
    if (RESOLVE && na < -1) {
        const unsigned long long key = amb_best[-2 - na];
        fb = __uint_as_float((unsigned int)(key >> 32));
        na = (int32_t)(key & 0xffffffffu);
        b[f] = fb;
        nassign[f] = na;
    }
    vb = fb;
}

template <bool RESOLVE>
__global__ void __launch_bounds__(EK_BLOCK)
ek_pw_leaf_kernel(const float *__restrict__ a, float *__restrict__ b,
                  const int32_t *__restrict__ assign,
                  int32_t *__restrict__ nassign,
                  const unsigned long long *__restrict__ amb_best, int64_t n,
                  int32_t win_lo,
                  int32_t win_count, const EkPwShape *__restrict__ shapes,
                  int n_full, int n_leaves_total, double *__restrict__ leafsum,
                  unsigned int *__restrict__ mask)
{
    __shared__ unsigned int acc;
    if (threadIdx.x == 0)
        acc = 0;
    __syncthreads();
    const int g = blockIdx.x * (EK_BLOCK / 8) + threadIdx.x / 8;
    const int l8 = threadIdx.x & 7;
    unsigned int m = 0;
    if (g < n_leaves_total) {
        const EkPwShape *sh = &shapes[0];
        int chunk = g / EK_PW_FULL_LEAVES, leaf = g % EK_PW_FULL_LEAVES;
        if (chunk >= n_full) {
            chunk = n_full;
            leaf = g - n_full * EK_PW_FULL_LEAVES;
            sh = &shapes[1];
        }
        const int64_t off = (int64_t)chunk * EK_PW_CHUNK + sh->leaf_off[leaf];
        const int len = sh->leaf_len[leaf];
        double ra = 0.0, rb = 0.0;
        const int body = (len < 8) ? 0 : len - (len % 8);
        if (body > 0) {
            for (int i = 0; i < body; i += 8) {
                const int64_t f = off + i + l8;
                double va, vb;
                int32_t oa, na;
                ek_pw_fetch<RESOLVE>(a, b, assign, nassign, amb_best, f, va, vb, oa,
                                     na);
                if (i == 0) {
                    ra = va * va;
                    rb = vb * vb;
                } else {
                    ra = ra + va * va;
                    rb = rb + vb * vb;
                }
                if (win_count > 0) {
                    if (oa != na) {
                        const int32_t ia = oa - win_lo, ib = na - win_lo;
                        if (ia >= 0 && ia < win_count)
                            m |= 1u << ia;
                        if (ib >= 0 && ib < win_count)
                            m |= 1u << ib;
                    }
                }
            }
#pragma unroll
            for (int off8 = 1; off8 < 8; off8 <<= 1) {   // (r0+r1)+(r2+r3) ...
                ra = ra + __shfl_xor(ra, off8, 8);
                rb = rb + __shfl_xor(rb, off8, 8);
            }
        }
        if (l8 == 0) {
            for (int i = body; i < len; ++i) {          // sequential tail
                const int64_t f = off + i;
                double va, vb;
                int32_t oa, na;
                ek_pw_fetch<RESOLVE>(a, b, assign, nassign, amb_best, f, va, vb, oa,
                                     na);
                ra = ra + va * va;
                rb = rb + vb * vb;
                if (win_count > 0) {
                    if (oa != na) {
                        const int32_t ia = oa - win_lo, ib = na - win_lo;
                        if (ia >= 0 && ia < win_count)
                            m |= 1u << ia;
                        if (ib >= 0 && ib < win_count)
                            m |= 1u << ib;
                    }
                }
            }
            leafsum[2 * (size_t)g + 0] = ra;
            leafsum[2 * (size_t)g + 1] = rb;
        }
    }
    if (m)
        atomicOr(&acc, m);
    __syncthreads();
    if (threadIdx.x == 0 && acc)
        atomicOr(mask, acc);
}

// one workgroup per chunk: the pairwise tree over its leaves, level by level
__global__ void __launch_bounds__(128)
ek_pw_chunk_kernel(const double *__restrict__ leafsum,
                   const EkPwShape *__restrict__ shapes, int n_full,
                   double *__restrict__ chunksum)
{
    __shared__ double va[2 * EK_PW_MAX_LEAVES], vb[2 * EK_PW_MAX_LEAVES];
    const int chunk = blockIdx.x;
    const EkPwShape *sh = (chunk < n_full) ? &shapes[0] : &shapes[1];
    const int first = (chunk < n_full) ? chunk * EK_PW_FULL_LEAVES
                                       : n_full * EK_PW_FULL_LEAVES;
    const int nl = sh->n_leaves, t = threadIdx.x;
    for (int i = t; i < nl; i += 128) {
        va[i] = leafsum[2 * (size_t)(first + i) + 0];
        vb[i] = leafsum[2 * (size_t)(first + i) + 1];
    }
    __syncthreads();
    for (int lev = 0; lev < sh->n_levels; ++lev) {
        const int k0 = sh->level_start[lev], k1 = sh->level_start[lev + 1];
        for (int k = k0 + t; k < k1; k += 128) {
            const int l = sh->node_l[k], r = sh->node_r[k];
            va[nl + k] = va[l] + va[r];
            vb[nl + k] = vb[l] + vb[r];
        }
        __syncthreads();
    }
    if (t == 0) {
        const int root = (sh->n_nodes > 0) ? nl + sh->n_nodes - 1 : 0;
        chunksum[2 * (size_t)chunk + 0] = va[root];
        chunksum[2 * (size_t)chunk + 1] = vb[root];
    }
}

// chunk sums left to right, packed with the counters into the one record the
// host reads back (or, across shards, exchanges)
__global__ void __launch_bounds__(EK_BLOCK)
ek_pw_pack_kernel(const double *__restrict__ chunksum, int n_chunks,
                  const unsigned int *__restrict__ n_amb,
                  const unsigned int *__restrict__ moved, int64_t n,
                  EkPamOut *__restrict__ out)
{
    // the chunk sums come in with one parallel read; the additions stay in
    // numpy's left-to-right order
    __shared__ double sa_s[EK_BLOCK], sb_s[EK_BLOCK];
    const int t = threadIdx.x;
    double sa = 0.0, sb = 0.0;
    for (int c0 = 0; c0 < n_chunks; c0 += EK_BLOCK) {
        const int c = c0 + t;
        if (c < n_chunks) {
            sa_s[t] = chunksum[2 * (size_t)c + 0];
            sb_s[t] = chunksum[2 * (size_t)c + 1];
        }
        __syncthreads();
        if (t == 0) {
            const int m = (n_chunks - c0 < EK_BLOCK) ? n_chunks - c0 : EK_BLOCK;
            for (int k = 0; k < m; ++k) {
                sa = sa + sa_s[k];
                sb = sb + sb_s[k];
            }
        }
        __syncthreads();
    }
    if (t != 0)
        return;
    out->sum_old = sa;
    out->sum_new = sb;
    out->n_frames = n;
    out->n_amb = *n_amb;
    out->moved = *moved;
}

// ---------------------------------------------------------------------------
// a window of proposals decided on the device (ek_pam_window_run)
// ---------------------------------------------------------------------------
// Inside a window a proposal is three launches: classification (which also
// takes over the trial state of an accepted predecessor), the ambiguous members
// against the medoids within reach, and this one: cost sums, verdict, set-up of
// the next proposal.
//
// Cost sums: a workgroup covers 32 leaves = 4096 consecutive frames = half a
// numpy chunk, a perfect subtree of the pairwise sum, so it adds its leaves up
// itself (adjacent pairs, level by level: floating-point addition commutes, so
// a butterfly over lanes gives every lane the in-order result) and hands ONE
// pair of sums to the last workgroup to finish (arrival counter, coherent
// stores / loads: ek_reduce.h).  The leaves of the shorter last chunk go through
// individually and are summed along that chunk's own tree.
//
// Verdict, by the last workgroup once the sums are in: accept iff
// mean(new^2) < mean(old^2), both in float64 (kmedoids.py:478-479, :683) -- the
// same two divisions and comparison the host made -- and
//   accepted: the cluster's medoid index is the proposed frame, the clusters of
//             the window whose membership changes are marked stale; the first
//             stale cluster after this one is where the window stops: its
//             proposal was drawn from a member list that no longer holds
//             (kmedoids.py:611-614).  The trial state becomes the state
//             (kmedoids.py:684-690) in the next slot's classification launch,
//             or in ek_pam_apply_kernel after the last slot;
//   rejected (or past the stop): the medoid table gets its row back.
// Then the NEXT proposal's trial table is set up (what ek_pam_trial_kernel does
// in a launch of its own).  A slot past the stop still runs its kernels -- they
// only write scratch -- but nothing of it is kept.
struct EkPwTail {
    unsigned int *tick;         // [0] top, [1 ..] EK_ARRIVE_G leaves
    int n_chunks;
    const unsigned int *n_amb;
    int64_t n;
};

__device__ __forceinline__ double ek_coh_ld_f64(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ek_coh_st_f64(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define EK_PW_WG_LEAVES (EK_BLOCK / 8)      // 32: half a full chunk

__global__ void __launch_bounds__(EK_BLOCK)
ek_pw_window_kernel(const float *__restrict__ a, float *__restrict__ b,
                    const int32_t *__restrict__ assign,
                    int32_t *__restrict__ nassign,
                    const unsigned long long *__restrict__ amb_best,
                    int32_t win_lo, int32_t win_count,
                    const EkPwShape *__restrict__ shapes, int n_full,
                    int n_leaves_total, double *__restrict__ part,
                    unsigned int *__restrict__ mask, EkPwTail tl, EkPamDecide dc)
{
    static_assert(EK_PW_FULL_LEAVES == 2 * EK_PW_WG_LEAVES, "half a chunk each");
    __shared__ unsigned int acc;
    __shared__ double wsum[2][EK_BLOCK / EK_WAVE];
    const int t = threadIdx.x;
    if (t == 0)
        acc = 0;
    __syncthreads();
    const int l8 = t & 7;
    const int n_half = 2 * n_full;              // workgroups over full chunks
    // part: [2 * n_half] half-chunk sums, then the last chunk's leaf sums
    double *ragsum = part + 2 * (size_t)n_half;
    unsigned int m = 0;
    if ((int)blockIdx.x < n_half) {
        const int64_t off = ((int64_t)blockIdx.x * EK_PW_WG_LEAVES + t / 8) * 128;
        float fa[16], fb[16];
        int32_t oa[16], na[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t f = off + 8 * i + l8;
            fa[i] = a[f];
            fb[i] = b[f];
            oa[i] = assign[f];
            na[i] = nassign[f];
        }
        double ra = 0.0, rb = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (na[i] < -1) {           // marked ambiguous member: resolved now
                const int64_t f = off + 8 * i + l8;
                const unsigned long long key = amb_best[-2 - na[i]];
                fb[i] = __uint_as_float((unsigned int)(key >> 32));
                na[i] = (int32_t)(key & 0xffffffffu);
                b[f] = fb[i];
                nassign[f] = na[i];
            }
            const double va = fa[i], vb = fb[i];
            if (i == 0) {
                ra = va * va;
                rb = vb * vb;
            } else {
                ra = ra + va * va;
                rb = rb + vb * vb;
            }
            if (win_count > 0 && oa[i] != na[i]) {
                const int32_t ia = oa[i] - win_lo, ib = na[i] - win_lo;
                if (ia >= 0 && ia < win_count)
                    m |= 1u << ia;
                if (ib >= 0 && ib < win_count)
                    m |= 1u << ib;
            }
        }
        // the leaf: (r0+r1)+(r2+r3) ..; then the eight leaves of the wave, the
        // four waves of the workgroup
#pragma unroll
        for (int o = 1; o < EK_WAVE; o <<= 1) {
            ra = ra + __shfl_xor(ra, o, EK_WAVE);
            rb = rb + __shfl_xor(rb, o, EK_WAVE);
        }
        if ((t & (EK_WAVE - 1)) == 0) {
            wsum[0][t / EK_WAVE] = ra;
            wsum[1][t / EK_WAVE] = rb;
        }
        __syncthreads();
        if (t < 2)
            ek_coh_st_f64(&part[2 * (size_t)blockIdx.x + t],
                          (wsum[t][0] + wsum[t][1]) + (wsum[t][2] + wsum[t][3]));
    } else {
        // leaves of the last, shorter chunk (any shape)
        const EkPwShape *sh = &shapes[1];
        const int leaf = ((int)blockIdx.x - n_half) * EK_PW_WG_LEAVES + t / 8;
        if (leaf < sh->n_leaves) {
            const int64_t off = (int64_t)n_full * EK_PW_CHUNK + sh->leaf_off[leaf];
            const int len = sh->leaf_len[leaf];
            double ra = 0.0, rb = 0.0;
            const int body = (len < 8) ? 0 : len - (len % 8);
            for (int i = 0; i < body; i += 8) {
                double va, vb;
                int32_t oa, na;
                ek_pw_fetch<true>(a, b, assign, nassign, amb_best, off + i + l8, va,
                                  vb, oa, na);
                if (i == 0) {
                    ra = va * va;
                    rb = vb * vb;
                } else {
                    ra = ra + va * va;
                    rb = rb + vb * vb;
                }
                if (win_count > 0 && oa != na) {
                    const int32_t ia = oa - win_lo, ib = na - win_lo;
                    if (ia >= 0 && ia < win_count)
                        m |= 1u << ia;
                    if (ib >= 0 && ib < win_count)
                        m |= 1u << ib;
                }
            }
            if (body > 0) {
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {
                    ra = ra + __shfl_xor(ra, o, 8);
                    rb = rb + __shfl_xor(rb, o, 8);
                }
            }
            if (l8 == 0) {
                for (int i = body; i < len; ++i) {      // sequential tail
                    double va, vb;
                    int32_t oa, na;
                    ek_pw_fetch<true>(a, b, assign, nassign, amb_best, off + i, va,
                                      vb, oa, na);
                    ra = ra + va * va;
                    rb = rb + vb * vb;
                    if (win_count > 0 && oa != na) {
                        const int32_t ia = oa - win_lo, ib = na - win_lo;
                        if (ia >= 0 && ia < win_count)
                            m |= 1u << ia;
                        if (ib >= 0 && ib < win_count)
                            m |= 1u << ib;
                    }
                }
                ek_coh_st_f64(&ragsum[2 * (size_t)leaf + 0], ra);
                ek_coh_st_f64(&ragsum[2 * (size_t)leaf + 1], rb);
            }
        }
    }
    if (m)
        atomicOr(&acc, m);
    __syncthreads();
    if (t == 0 && acc)
        atomicOr(mask, acc);
    if (!ek_arrive_last_tree(tl.tick, tl.tick + 1))
        return;

    // ---- the last workgroup: chunk sums, then the chunks left to right ------------
    __shared__ double ca[EK_BLOCK], cb[EK_BLOCK];
    __shared__ double la[2 * EK_PW_MAX_LEAVES], lb[2 * EK_PW_MAX_LEAVES];
    __shared__ int s_accept;
    if (tl.n_chunks > n_full) {
        // the last chunk's own tree, level by level
        const EkPwShape *sh = &shapes[1];
        const int nl = sh->n_leaves;
        for (int i = t; i < nl; i += EK_BLOCK) {
            la[i] = ek_coh_ld_f64(&ragsum[2 * (size_t)i + 0]);
            lb[i] = ek_coh_ld_f64(&ragsum[2 * (size_t)i + 1]);
        }
        __syncthreads();
        for (int lev = 0; lev < sh->n_levels; ++lev) {
            const int k0 = sh->level_start[lev], k1 = sh->level_start[lev + 1];
            for (int k = k0 + t; k < k1; k += EK_BLOCK) {
                const int l = sh->node_l[k], r = sh->node_r[k];
                la[nl + k] = la[l] + la[r];
                lb[nl + k] = lb[l] + lb[r];
            }
            __syncthreads();
        }
    }
    double sa = 0.0, sb = 0.0;
    for (int c0 = 0; c0 < tl.n_chunks; c0 += EK_BLOCK) {
        const int c = c0 + t;
        if (c < n_full) {
            const double a0 = ek_coh_ld_f64(&part[4 * (size_t)c + 0]),
                         b0 = ek_coh_ld_f64(&part[4 * (size_t)c + 1]),
                         a1 = ek_coh_ld_f64(&part[4 * (size_t)c + 2]),
                         b1 = ek_coh_ld_f64(&part[4 * (size_t)c + 3]);
            ca[t] = a0 + a1;
            cb[t] = b0 + b1;
        } else if (c == n_full && c < tl.n_chunks) {
            const EkPwShape *sh = &shapes[1];
            const int root = (sh->n_nodes > 0) ? sh->n_leaves + sh->n_nodes - 1 : 0;
            ca[t] = la[root];
            cb[t] = lb[root];
        }
        __syncthreads();
        if (t == 0) {
            const int mm = (tl.n_chunks - c0 < EK_BLOCK) ? tl.n_chunks - c0 : EK_BLOCK;
            for (int k0 = 0; k0 < mm; k0 += 16) {
                // sixteen reads in flight, the additions in order
                double xa[16], xb[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    xa[j] = ca[(k0 + j) & (EK_BLOCK - 1)];
                    xb[j] = cb[(k0 + j) & (EK_BLOCK - 1)];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (k0 + j < mm) {
                        sa = sa + xa[j];
                        sb = sb + xb[j];
                    }
            }
        }
        __syncthreads();
    }
    // ---- verdict and the next proposal's trial table ---------------------------------
    EkPamWin *win = dc.win;
    EkPamOut o;
    if (t == 0) {
        o.sum_old = sa;
        o.sum_new = sb;
        o.n_frames = tl.n;
        o.n_amb = *tl.n_amb;
        // every workgroup's bits are in: they were OR-ed before its ticket
        o.moved = __hip_atomic_load(mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        win->out[dc.slot] = o;
        const bool live = dc.slot < win->stop;
        const double old_cost = o.sum_old / dc.n_total,
                     new_cost = o.sum_new / dc.n_total;
        const bool accept = live && new_cost < old_cost;
        s_accept = accept ? 1 : 0;
        if (live) {
            win->accept[dc.slot] = accept ? 1 : 0;
            if ((int64_t)o.n_amb > dc.max_amb)
                win->err = 1 + dc.slot;
            if (accept) {
                if (dc.med_idx)
                    dc.med_idx[dc.cid] = dc.frame;
                const uint32_t stale = win->stale | o.moved;
                win->stale = stale;
                const uint32_t later = (dc.slot >= 31) ? 0u : (stale >> (dc.slot + 1));
                if (later) {
                    const int first = dc.slot + 1 + (__ffs((int)later) - 1);
                    if (first < win->stop)
                        win->stop = first;
                }
            }
        }
    }
    __syncthreads();
    const bool accept = s_accept != 0;
    const int A = dc.A, K = dc.K;
    // thread t owns elements t, t + 256, .. of every row: no barrier needed
    for (int r0 = t; r0 < 3 * A; r0 += 4 * EK_BLOCK) {
        float sv[4], nx[4], pr[4];              // all loads first
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * EK_BLOCK;
            const bool in = r < 3 * A;
            sv[u] = in ? dc.aos[(size_t)K * 3 * A + r] : 0.f;
            nx[u] = (in && dc.next_cid >= 0) ? dc.aos[(size_t)dc.next_cid * 3 * A + r]
                                             : 0.f;
            pr[u] = (in && dc.next_cid >= 0)
                        ? dc.frames_aos[(size_t)dc.next_frame * 3 * A + r]
                        : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * EK_BLOCK;
            if (r >= 3 * A)
                continue;
            if (!accept)
                dc.aos[(size_t)dc.cid * 3 * A + r] = sv[u];
            if (dc.next_cid >= 0) {
                dc.aos[(size_t)K * 3 * A + r] = nx[u];
                dc.aos[(size_t)dc.next_cid * 3 * A + r] = pr[u];
            }
        }
    }
    if (t == 0) {
        if (!accept)
            dc.Gm[dc.cid] = dc.Gm[K];
        if (dc.next_cid >= 0) {
            dc.Gm[K] = dc.Gm[dc.next_cid];
            dc.Gm[dc.next_cid] = dc.G[dc.next_frame];
            dc.amb_count[0] = 0;
            dc.amb_count[1] = 0;
            dc.amb_count[2] = 0;
            *dc.moved = 0;
        }
    }
}

// amb_best != nullptr: the trial state still carries marked ambiguous members
// (ek_launch_pam_classify(.., mark = 1)); they are resolved on the way
void ek_launch_sumsq_pack(const float *a, float *b, const int32_t *assign,
                          int32_t *nassign, int64_t n, int32_t win_lo,
                          int32_t win_count, const EkPwShape *shapes, int n_full,
                          int n_leaves_total, int n_chunks, double *part,
                          const unsigned int *n_amb, unsigned int *moved,
                          EkPamOut *out, hipStream_t s,
                          const unsigned long long *amb_best, unsigned int *tick,
                          const EkPamDecide *decide)
{
    double *leafsum = part;
    double *chunksum = part + 2 * (size_t)n_leaves_total;
    if (n_leaves_total > 0 && amb_best && tick && decide) {
        // inside a window: one launch, the last workgroup finishes the sums and
        // decides (*out is decide->win->out[decide->slot])
        EkPwTail tl;
        tl.tick = tick;
        tl.n_chunks = n_chunks;
        tl.n_amb = n_amb;
        tl.n = n;
        const int n_rag = n_leaves_total - n_full * EK_PW_FULL_LEAVES;
        const int blocks = 2 * n_full +
                           (n_rag + EK_PW_WG_LEAVES - 1) / EK_PW_WG_LEAVES;
        hipLaunchKernelGGL(ek_pw_window_kernel, dim3(blocks), dim3(EK_BLOCK), 0, s,
                           a, b, assign, nassign, amb_best, win_lo, win_count,
                           shapes, n_full, n_leaves_total, part, moved, tl,
                           *decide);
        return;
    }
    if (n_leaves_total > 0) {
        const int per = EK_BLOCK / 8;
        if (amb_best)
            hipLaunchKernelGGL((ek_pw_leaf_kernel<true>),
                               dim3((n_leaves_total + per - 1) / per),
                               dim3(EK_BLOCK), 0, s, a, b, assign, nassign,
                               amb_best, n, win_lo, win_count, shapes, n_full,
                               n_leaves_total, leafsum, moved);
        else
            hipLaunchKernelGGL((ek_pw_leaf_kernel<false>),
                               dim3((n_leaves_total + per - 1) / per),
                               dim3(EK_BLOCK), 0, s, a, b, assign, nassign,
                               amb_best, n, win_lo, win_count, shapes, n_full,
                               n_leaves_total, leafsum, moved);
        hipLaunchKernelGGL(ek_pw_chunk_kernel, dim3(n_chunks), dim3(128), 0, s,
                           leafsum, shapes, n_full, chunksum);
    }
    hipLaunchKernelGGL(ek_pw_pack_kernel, dim3(1), dim3(EK_BLOCK), 0, s, chunksum,
                       n_chunks, n_amb, moved, n, out);
}

// leaf sums and chunk sums of one array alone (the tree ek_pam_sparse.hip keeps
// up to date): part[2 g] / part[2 (n_leaves + c)], the odd entries are scratch
void ek_launch_pw_tree(const float *dist, const int32_t *assign, int64_t n,
                       const EkPwShape *shapes, int n_full, int n_leaves_total,
                       int n_chunks, double *part, unsigned int *mask_scratch,
                       hipStream_t s)
{
    if (n_leaves_total <= 0)
        return;
    const int per = EK_BLOCK / 8;
    hipLaunchKernelGGL((ek_pw_leaf_kernel<false>),
                       dim3((n_leaves_total + per - 1) / per), dim3(EK_BLOCK), 0, s,
                       dist, const_cast<float *>(dist), assign,
                       const_cast<int32_t *>(assign), nullptr, n, 0, 0, shapes, n_full,
                       n_leaves_total, part, mask_scratch);
    hipLaunchKernelGGL(ek_pw_chunk_kernel, dim3(n_chunks), dim3(128), 0, s, part,
                       shapes, n_full, part + 2 * (size_t)n_leaves_total);
}

// chunk sums of both columns of `part` ([2 g + 0 / 1] leaf sums) and the totals,
// chunks added left to right -> out2[0], out2[1]  (ek_features.hip: PAM over
// float64 distances brings its own leaf kernel)
__global__ void __launch_bounds__(EK_WAVE)
ek_pw_total_kernel(const double *__restrict__ chunksum, int n_chunks,
                   double *__restrict__ out2)
{
    if (threadIdx.x >= 2)
        return;
    double s = 0.0;
    for (int c = 0; c < n_chunks; ++c)
        s = s + chunksum[2 * (size_t)c + threadIdx.x];
    out2[threadIdx.x] = s;
}

// the chunk sums alone (the caller adds them left to right itself)
void ek_launch_pw_chunks(double *part, const EkPwShape *shapes, int n_full,
                         int n_leaves_total, int n_chunks, hipStream_t s)
{
    double *chunksum = part + 2 * (size_t)n_leaves_total;
    if (n_leaves_total > 0)
        hipLaunchKernelGGL(ek_pw_chunk_kernel, dim3(n_chunks), dim3(128), 0, s, part,
                           shapes, n_full, chunksum);
}

void ek_launch_pw_chunks_total(double *part, const EkPwShape *shapes, int n_full,
                               int n_leaves_total, int n_chunks, double *out2,
                               hipStream_t s)
{
    double *chunksum = part + 2 * (size_t)n_leaves_total;
    if (n_leaves_total > 0)
        hipLaunchKernelGGL(ek_pw_chunk_kernel, dim3(n_chunks), dim3(128), 0, s, part,
                           shapes, n_full, chunksum);
    hipLaunchKernelGGL(ek_pw_total_kernel, dim3(1), dim3(EK_WAVE), 0, s, chunksum,
                       n_chunks, out2);
}

// host: the pairwise tree of a chunk of `len` elements (len <= EK_PW_CHUNK)
static int ek_pw_rec(int off, int len, EkPwShape *sh, int *node_level,
                     int *tmp_l, int *tmp_r, int *n_tmp)
{
    if (len <= 128) {
        const int id = sh->n_leaves++;
        sh->leaf_off[id] = off;
        sh->leaf_len[id] = len;
        return id;                              // leaves: ids 0..
    }
    int n2 = len / 2;
    n2 -= n2 % 8;
    const int l = ek_pw_rec(off, n2, sh, node_level, tmp_l, tmp_r, n_tmp);
    const int r = ek_pw_rec(off + n2, len - n2, sh, node_level, tmp_l, tmp_r, n_tmp);
    const int k = (*n_tmp)++;
    tmp_l[k] = l;
    tmp_r[k] = r;
    const int ll = (l >= 1000) ? node_level[l - 1000] : 0;
    const int lr = (r >= 1000) ? node_level[r - 1000] : 0;
    node_level[k] = 1 + (ll > lr ? ll : lr);
    return 1000 + k;                            // internal nodes: ids 1000..
}

void ek_pw_build_shape(int len, EkPwShape *sh)
{
    *sh = EkPwShape();
    if (len <= 0)
        return;
    int node_level[EK_PW_MAX_LEAVES], tmp_l[EK_PW_MAX_LEAVES], tmp_r[EK_PW_MAX_LEAVES];
    int n_tmp = 0;
    ek_pw_rec(0, len, sh, node_level, tmp_l, tmp_r, &n_tmp);
    // order the internal nodes by level (children before parents), root last
    int order[EK_PW_MAX_LEAVES], pos[EK_PW_MAX_LEAVES], cnt = 0, max_level = 0;
    for (int k = 0; k < n_tmp; ++k)
        if (node_level[k] > max_level)
            max_level = node_level[k];
    sh->n_levels = max_level;
    for (int lev = 1; lev <= max_level; ++lev) {
        sh->level_start[lev - 1] = cnt;
        for (int k = 0; k < n_tmp; ++k)
            if (node_level[k] == lev) {
                pos[k] = cnt;
                order[cnt++] = k;
            }
    }
    sh->level_start[max_level] = cnt;
    sh->n_nodes = cnt;
    const int nl = sh->n_leaves;
    for (int q = 0; q < cnt; ++q) {
        const int k = order[q];
        const int l = tmp_l[k], r = tmp_r[k];
        sh->node_l[q] = (l >= 1000) ? nl + pos[l - 1000] : l;
        sh->node_r[q] = (r >= 1000) ? nl + pos[r - 1000] : r;
    }
}

// ---- proposal prefetch restricted to the frames a proposal can touch ------------------
// A frame f with label a at distance d(f) from its medoid m_a cannot come closer
// to a proposed medoid p than D(m_a, p) - d(f) (minimal RMSD is a metric), so if
// D(m_a, p) >= 2 d(f) it stays where it is whatever the exact distance: the
// classification of kmedoids.py:644-658 only asks "is it closer than d(f)?".
// Exact distances are therefore needed only for the frames that fail that test
// for some proposal of the window, and for the members of the window's own
// clusters -- the only frames whose distance can GROW while the window is worked
// through (as ambiguous members of an accepted proposal), which would invalidate
// a test made at prefetch time.  Every other frame gets +inf.  The comparison
// carries a margin far above the rounding of a distance; D is only compared, so
// its summation order is free.

// Distances of a few thousand rows to a few columns -- too little work to fill
// the chip with the streaming kernels, whose one wave per 64 frames then runs
// at memory latency.  Here a workgroup takes 64 rows x EK_PAM_GROUP columns,
// wave = column, lane = row; the rows (frame-major, 12 A contiguous bytes each)
// and the columns' coordinates go through LDS in slices of EK_PAIR_CH atoms,
// read from memory once along the rows with the next slice's loads in flight
// during the FMAs.  Every (row, column) pair is still one lane's IEEE FMA
// chains in ascending atom order, S[3 r + c] += row_r * column_c, followed by
// ek_rmsd_from_S(S, G_row, G_column): bit-identical to the streaming kernels.
//
// MODE 0, the distance tables of a window (rows: the medoid table):
//   T[j * K + c] = rmsd(medoid c, proposal j), j < n_prop
//   dmin[g * K + c] = min of T[j * K + c] over the columns j of group g
//   O[i * K + c] = rmsd(medoid c, medoid old_lo + i), i < n_old -- what the
//                  pruning of slot i's ambiguous members asks for
//                  (ek_pam_prune_kernel / the classification's last workgroup)
//   `held` is the row a rejected proposal still occupies (its medoid is in row
//   K), or -1.  grid (ceil(K / 64), column groups of proposals + of old medoids).
//   Round 5, `dprop` given (proposal j is a MEMBER of cluster old_lo + j, at
//   dprop[j] from its medoid, n_old == n_prop): every use of T is a lower bound
//   -- "a medoid farther than .. cannot matter", "a frame whose medoid is at
//   least twice its distance from the proposal cannot move" -- and the triangle
//   inequality gives one from O alone: rmsd(medoid c, proposal j) >=
//   O[j][c] - dprop[j].  Only the old medoids' columns are computed (half the
//   pairs), T[j][c] = max(O[j][c] - dprop[j], 0), dmin from those; a looser bound
//   lists a few more frames and medoids, results cannot change.
// MODE 1, a window's proposals against the frames they can touch (rows: the
//   listed frames of the frame-major copy):
//   vecs[j * n_pad + list[i]] = rmsd(frame list[i], proposal j)
#ifndef EK_PAIR_CH
#define EK_PAIR_CH 48
#endif
#ifndef EK_PAIR_LD
#define EK_PAIR_LD (EK_WAVE + 8)
#endif
struct EkPairArgs {
    const float *aos;           // rows: [.][3A]
    const double *G;            // their traces
    int32_t A;
    const unsigned char *recs;  // columns: records
    int32_t n_col;
    // MODE 0
    int32_t K, held, old_lo, n_old;
    float *T, *O, *dmin;
    const float *dprop;         // != nullptr: T and dmin as lower bounds from O (below)
    // MODE 1
    const uint32_t *list;
    int64_t n_rows, n_pad;
    float *vecs;
};

static inline size_t ek_pair_lds_bytes()
{
    return (size_t)(3 * EK_PAIR_CH * EK_PAIR_LD + EK_PAM_GROUP * 3 * EK_PAIR_CH) *
           sizeof(float);
}

// A wave takes EK_PAIR_CPW columns, a workgroup is EK_PAM_GROUP / EK_PAIR_CPW waves.
// Two columns per wave (the rows' coordinates read from LDS once for both) were
// measured in round 5 and lost: 42 -> 51 us (tables), 46 -> 56 us (listed frames) --
// with half the waves the rows' loads, sixteen 16-byte pieces per instruction, are
// what the kernel waits for, not the LDS reads (profiles/r05/pam_pairs_cpw2.log).
#ifndef EK_PAIR_CPW
#define EK_PAIR_CPW 1
#endif
template <int MODE>
__global__ void __launch_bounds__(EK_PAM_GROUP / EK_PAIR_CPW *EK_WAVE)
ek_pam_pairs_kernel(EkPairArgs p)
{
    constexpr int CPW = EK_PAIR_CPW;
    constexpr int NT = EK_PAM_GROUP / CPW * EK_WAVE;
    // (a slice's element e of row m sits at e * LD + m.  Reads take consecutive rows:
    // conflict-free with any LD.  The stores of a wave are 8 rows x 8 consecutive
    // elements: with LD = 65 their banks are (e + m) % 32, fifteen of them for 64
    // lanes -- SQ_LDS_BANK_CONFLICT as large as the LDS's active cycles, round 5;
    // with LD = 64 + 8 they are (8 e + m) % 32: two lanes per bank, what 64 lanes cost
    // anyway)
    constexpr int LD = EK_PAIR_LD;
    constexpr int TPR = NT / EK_WAVE;           // threads per row
    constexpr int NLD = 3 * EK_PAIR_CH / TPR;   // loads per thread and slice
    constexpr int NY = (3 * EK_PAIR_CH + EK_WAVE - 1) / EK_WAVE;
    static_assert(EK_PAM_GROUP % CPW == 0 && (3 * EK_PAIR_CH) % TPR == 0, "tiling");
    extern __shared__ __attribute__((aligned(16))) float pair_lds[];
    float *tile = pair_lds;                                 // [3 CH][LD]
    float *ytile = pair_lds + 3 * EK_PAIR_CH * LD;          // [columns][3 CH]
    __shared__ float tmin[EK_PAM_GROUP][EK_WAVE];
    const int A = p.A;
    const int lane = threadIdx.x & (EK_WAVE - 1);
    // grid.y: the groups of EK_PAM_GROUP columns -- the proposals' first, then
    // (MODE 0) the old medoids'
    const int gp = (p.n_col + EK_PAM_GROUP - 1) / EK_PAM_GROUP;
    const bool old = MODE == 0 && (int)blockIdx.y >= gp;
    const int grp = old ? (int)blockIdx.y - gp : (int)blockIdx.y;
    const int jw = __builtin_amdgcn_readfirstlane(threadIdx.x / EK_WAVE);
    const int ncol = old ? p.n_old : p.n_col;
    // this wave's columns
    int jc[CPW];
    bool live[CPW];
    const float *y[CPW];
    double Gy[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        jc[u] = grp * EK_PAM_GROUP + jw * CPW + u;
        live[u] = jc[u] < ncol;
        y[u] = p.aos;
        Gy[u] = 0.0;
        if (live[u]) {
            if (old) {
                const int r = (p.old_lo + jc[u] == p.held) ? p.K : p.old_lo + jc[u];
                y[u] = p.aos + (size_t)r * 3 * A;
                Gy[u] = p.G[r];
            } else {
                const size_t rstride = ek_rec_bytes(A);
                y[u] = (const float *)(p.recs + (size_t)jc[u] * rstride + sizeof(EkRecHdr));
                Gy[u] = ((const EkRecHdr *)(p.recs + (size_t)jc[u] * rstride))->trace;
            }
        }
    }
    // row `m` of the workgroup: where it lives, whether it exists
    auto row_of = [&](int m, bool &ok) -> int64_t {
        const int64_t i = (int64_t)blockIdx.x * EK_WAVE + m;
        if (MODE == 0) {
            ok = i < p.K;
            return (i == p.held) ? p.K : (ok ? i : 0);
        }
        ok = i < p.n_rows;
        return ok ? (int64_t)p.list[i] : 0;
    };
    // loading: TPR threads per row, each every TPR-th element of the slice; a
    // column's coordinates by its own wave
    const int lm = threadIdx.x / TPR, le = threadIdx.x % TPR;
    bool lok;
    const float *lrow = p.aos + (size_t)row_of(lm, lok) * 3 * A;
    float S[CPW][9];
#pragma unroll
    for (int u = 0; u < CPW; ++u)
#pragma unroll
        for (int q = 0; q < 9; ++q)
            S[u][q] = 0.f;
    float v[NLD], vy[CPW][NY];  // the slice after the one being worked on
#define EK_PAIR_LOAD(A0)                                                       \
    _Pragma("unroll") for (int k = 0; k < NLD; ++k) {                          \
        const int e = le + TPR * k;                                            \
        v[k] = (3 * (A0) + e < 3 * A && lok) ? lrow[3 * (A0) + e] : 0.f;       \
    }                                                                          \
    _Pragma("unroll") for (int u = 0; u < CPW; ++u)                            \
    _Pragma("unroll") for (int k = 0; k < NY; ++k) {                           \
        const int e = lane + EK_WAVE * k;                                      \
        vy[u][k] = (live[u] && e < 3 * EK_PAIR_CH && 3 * (A0) + e < 3 * A)     \
                       ? y[u][3 * (A0) + e] : 0.f;                             \
    }
    EK_PAIR_LOAD(0)
    for (int a0 = 0; a0 < A; a0 += EK_PAIR_CH) {
        const int ch = (A - a0 < EK_PAIR_CH) ? A - a0 : EK_PAIR_CH;
        __syncthreads();                        // the slice before is done with
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            tile[(le + TPR * k) * LD + lm] = v[k];
#pragma unroll
        for (int u = 0; u < CPW; ++u)
#pragma unroll
            for (int k = 0; k < NY; ++k)
                if (lane + EK_WAVE * k < 3 * EK_PAIR_CH)
                    ytile[(jw * CPW + u) * 3 * EK_PAIR_CH + lane + EK_WAVE * k] = vy[u][k];
        __syncthreads();
        if (a0 + EK_PAIR_CH < A)                // in flight during the FMAs
            EK_PAIR_LOAD(a0 + EK_PAIR_CH)
        if (live[0]) {                          // (the wave's first column: the others follow it)
            const float *yt = ytile + jw * CPW * 3 * EK_PAIR_CH;
#pragma unroll 4
            for (int a = 0; a < ch; ++a) {
                const float x0 = tile[(3 * a + 0) * LD + lane],
                            x1 = tile[(3 * a + 1) * LD + lane],
                            x2 = tile[(3 * a + 2) * LD + lane];
#pragma unroll
                for (int u = 0; u < CPW; ++u) {
                    const float y0 = yt[u * 3 * EK_PAIR_CH + 3 * a],
                                y1 = yt[u * 3 * EK_PAIR_CH + 3 * a + 1],
                                y2 = yt[u * 3 * EK_PAIR_CH + 3 * a + 2];
                    S[u][0] = fmaf(x0, y0, S[u][0]); S[u][1] = fmaf(x0, y1, S[u][1]);
                    S[u][2] = fmaf(x0, y2, S[u][2]); S[u][3] = fmaf(x1, y0, S[u][3]);
                    S[u][4] = fmaf(x1, y1, S[u][4]); S[u][5] = fmaf(x1, y2, S[u][5]);
                    S[u][6] = fmaf(x2, y0, S[u][6]); S[u][7] = fmaf(x2, y1, S[u][7]);
                    S[u][8] = fmaf(x2, y2, S[u][8]);
                }
            }
        }
    }
#undef EK_PAIR_LOAD
    bool ok;
    const int64_t row = row_of(lane, ok);
    const double Gx = ok ? p.G[row] : 0.0;
    float D[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        D[u] = __builtin_inff();
        if (live[u] && ok)
            D[u] = ek_rmsd_from_S(S[u], Gx, Gy[u], A);
    }
    if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < CPW; ++u)
            if (live[u] && ok)
                p.vecs[(size_t)jc[u] * p.n_pad + row] = D[u];
        return;
    }
    const int c = blockIdx.x * EK_WAVE + lane;
#pragma unroll
    for (int u = 0; u < CPW; ++u)
        if (live[u] && ok)
            (old ? p.O : p.T)[(size_t)jc[u] * p.K + c] = D[u];
    if (old && !p.dprop)
        return;
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        if (old) {              // (bounds: see above)
            D[u] = (live[u] && ok) ? fmaxf(D[u] - p.dprop[jc[u]], 0.f) : __builtin_inff();
            if (live[u] && ok)
                p.T[(size_t)jc[u] * p.K + c] = D[u];
        }
        tmin[jw * CPW + u][lane] = D[u];
    }
    __syncthreads();
    if (jw == 0 && ok) {
        float m = tmin[0][lane];
#pragma unroll
        for (int q = 1; q < EK_PAM_GROUP; ++q)
            m = fminf(m, tmin[q][lane]);
        p.dmin[(size_t)grp * p.K + c] = m;
    }
}

// ---- the same distances through the matrix cores (round 5) -------------------------------------
// The kernel above spends a quarter of a wave's time in its FMAs and the rest waiting
// (slices through LDS, two barriers each; SQ counters in profiles/r05); 64 rows x 8
// columns per workgroup are barely one workgroup per CU for a window's few thousand
// rows.  Here a WAVE takes 16 rows x 16 columns and v_mfma_f32_16x16x4_f32 does four
// atoms per instruction (ek_pass16.hip: every accumulator is the oracle's FMA chain in
// atom order, atoms past the last are zeros in both operands): lane l holds atom
// l / 16 of the trip for row l % 16 (A operand) and for column l % 16 (B operand), both
// as the 12 contiguous bytes of that atom in the frame-major rows -- no LDS, no
// barrier, the loads of the trips ahead in flight; nine instructions per trip, one per
// (row coordinate, column coordinate).  A lane ends with S of four (row, column)
// pairs of ONE column.  Four waves per workgroup (64 rows), columns in groups of 16
// (grid.y: the proposals' groups first, then -- MODE 0 -- the old medoids').
typedef float ek_p4 __attribute__((ext_vector_type(4)));
struct EkF3 {
    float x, y, z;
};
#ifndef EK_PP16_DEPTH
#define EK_PP16_DEPTH 8
#endif
template <int MODE>
__global__ void __launch_bounds__(4 * EK_WAVE)
ek_pam_pairs16_kernel(EkPairArgs p)
{
    const int A = p.A;
    const int lane = threadIdx.x & (EK_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / EK_WAVE);
    const int gp = (p.n_col + 15) / 16;
    const bool old = MODE == 0 && (int)blockIdx.y >= gp;
    const int cg = old ? (int)blockIdx.y - gp : (int)blockIdx.y;
    const int ncol = old ? p.n_old : p.n_col;
    // this lane's column (B operand, and the column of its four results)
    const int jc = cg * 16 + (lane & 15);
    const bool clive = jc < ncol;
    const float *y = p.aos;
    double Gy = 0.0;
    if (clive) {
        if (old) {
            const int r = (p.old_lo + jc == p.held) ? p.K : p.old_lo + jc;
            y = p.aos + (size_t)r * 3 * A;
            Gy = p.G[r];
        } else {
            const size_t rstride = ek_rec_bytes(A);
            y = (const float *)(p.recs + (size_t)jc * rstride + sizeof(EkRecHdr));
            Gy = ((const EkRecHdr *)(p.recs + (size_t)jc * rstride))->trace;
        }
    }
    // row `i` of the launch: where it lives, whether it exists
    auto row_of = [&](int64_t i, bool &ok) -> int64_t {
        if (MODE == 0) {
            ok = i < p.K;
            return (i == p.held) ? p.K : (ok ? i : 0);
        }
        ok = i < p.n_rows;
        return ok ? (int64_t)p.list[i] : 0;
    };
    const int64_t i0 = (int64_t)blockIdx.x * EK_WAVE + wave * 16;
    if (i0 >= (MODE == 0 ? (int64_t)p.K : p.n_rows))
        return;                                 // (a whole wave: no barrier in this kernel)
    bool lok;
    const float *x = p.aos + (size_t)row_of(i0 + (lane & 15), lok) * 3 * A;
    const int ka = lane >> 4;                   // this lane's atom of a trip
    const int NQ = (A + 3) / 4;
    ek_p4 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
        acc[q] = ek_p4{0.f, 0.f, 0.f, 0.f};
    // (the loads of trip t + DEPTH are issued before the instructions of trip t and
    // stay there -- scheduling barriers: left alone the compiler sinks every load to
    // just before its use and waits for it at full latency.  Addresses are clamped to
    // the row's last atom and the value masked when it is used: no branch around a load)
    typedef const __attribute__((address_space(1))) float *ek_gf;
    const ek_gf xg = (ek_gf)x, yg = (ek_gf)y;
    EkF3 xa[EK_PP16_DEPTH], yb[EK_PP16_DEPTH];
#define EK_PP16_LOAD(D_, T_)                                                    \
    {                                                                          \
        const int at = min(4 * (T_) + ka, A - 1);                              \
        const ek_gf px = xg + 3 * at, py = yg + 3 * at;                        \
        xa[D_].x = px[0];                                                      \
        xa[D_].y = px[1];                                                      \
        xa[D_].z = px[2];                                                      \
        yb[D_].x = py[0];                                                      \
        yb[D_].y = py[1];                                                      \
        yb[D_].z = py[2];                                                      \
    }
#pragma unroll
    for (int d = 0; d < EK_PP16_DEPTH; ++d)
        EK_PP16_LOAD(d, d)
    __builtin_amdgcn_sched_barrier(0);
    for (int t0 = 0; t0 < NQ; t0 += EK_PP16_DEPTH) {
#pragma unroll
        for (int d = 0; d < EK_PP16_DEPTH; ++d) {
            const bool in = 4 * (t0 + d) + ka < A;
            EkF3 cx = xa[d], cy = yb[d];
            if (!(in && lok))
                cx.x = cx.y = cx.z = 0.f;
            if (!(in && clive))
                cy.x = cy.y = cy.z = 0.f;
            EK_PP16_LOAD(d, t0 + d + EK_PP16_DEPTH)
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cx.x, cy.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cx.x, cy.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(cx.x, cy.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(cx.y, cy.x, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(cx.y, cy.y, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(cx.y, cy.z, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(cx.z, cy.x, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(cx.z, cy.y, acc[7], 0, 0, 0);
            acc[8] = __builtin_amdgcn_mfma_f32_16x16x4f32(cx.z, cy.z, acc[8], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef EK_PP16_LOAD
    // ---- this lane's four pairs: rows 4 (l / 16) + r, column l % 16 --------------------
    float D[4];
    int64_t rowv[4];
    bool okv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + 4 * ka + r;
        rowv[r] = row_of(i, okv[r]);
        D[r] = __builtin_inff();
        if (clive && okv[r]) {
            float S[9];
#pragma unroll
            for (int q = 0; q < 9; ++q)
                S[q] = acc[q][r];
#ifdef EK_PP16_NOSOLVE          // (measurement builds: what the solves cost)
            D[r] = S[0] + (float)p.G[rowv[r]];
#else
            D[r] = ek_rmsd_from_S(S, p.G[rowv[r]], Gy, A);
#endif
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (clive && okv[r])
                p.vecs[(size_t)jc * p.n_pad + rowv[r]] = D[r];
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t c = i0 + 4 * ka + r;
        if (clive && okv[r])
            (old ? p.O : p.T)[(size_t)jc * p.K + c] = D[r];
    }
    if (old && !p.dprop)
        return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t c = i0 + 4 * ka + r;
        float m = D[r];
        if (old) {              // (bounds: see above)
            m = (clive && okv[r]) ? fmaxf(D[r] - p.dprop[jc], 0.f) : __builtin_inff();
            if (clive && okv[r])
                p.T[(size_t)jc * p.K + c] = m;
        }
        // the minimum over a group of EK_PAM_GROUP columns: eight neighbouring lanes
        m = fminf(m, __shfl_xor(m, 1, 64));
        m = fminf(m, __shfl_xor(m, 2, 64));
        m = fminf(m, __shfl_xor(m, 4, 64));
        if ((lane & 7) == 0 && okv[r] && cg * 16 + (lane & 15) < ((ncol + 7) / 8) * 8)
            p.dmin[(size_t)(cg * 2 + ((lane & 15) >> 3)) * p.K + c] = m;
    }
}

template <int MODE>
static void ek_pairs16_launch(const EkPairArgs &p, int64_t n_rows, int n_prop, int n_old,
                              hipStream_t s)
{
    const dim3 grid((unsigned)((n_rows + EK_WAVE - 1) / EK_WAVE),
                    (unsigned)((n_prop + 15) / 16 + (n_old + 15) / 16));
    hipLaunchKernelGGL(ek_pam_pairs16_kernel<MODE>, grid, dim3(4 * EK_WAVE), 0, s, p);
}

// 1: the matrix-core form above; 0: the LDS-staged kernel (ek_set_option key 21)
int ek_pam_pairs_form = 1;

template <int MODE>
static void ek_pairs_launch(const EkPairArgs &p, dim3 grid, hipStream_t s)
{
    const size_t lds = ek_pair_lds_bytes();     // above the 64 KB a kernel gets unasked
    (void)hipFuncSetAttribute((const void *)ek_pam_pairs_kernel<MODE>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ek_pam_pairs_kernel<MODE>, grid,
                       dim3(EK_PAM_GROUP / EK_PAIR_CPW * EK_WAVE), lds, s, p);
}

void ek_launch_pam_tables(const float *aos, const double *Gm, int A, int K, int held,
                          const unsigned char *recs, int n_prop, int old_lo,
                          int n_old, float *T, float *O, float *dmin,
                          hipStream_t s, const float *dprop)
{
    if (dprop && n_old == n_prop) {     // T and dmin as bounds from O: old columns only
        EkPairArgs q = {};
        q.aos = aos;
        q.G = Gm;
        q.A = A;
        q.recs = recs;
        q.n_col = 0;
        q.K = K;
        q.held = held;
        q.old_lo = old_lo;
        q.n_old = n_old;
        q.T = T;
        q.O = O;
        q.dmin = dmin;
        q.dprop = dprop;
        if (ek_pam_pairs_form)
            ek_pairs16_launch<0>(q, K, 0, n_old, s);
        else
            ek_pairs_launch<0>(q, dim3((K + EK_WAVE - 1) / EK_WAVE,
                                       (n_old + EK_PAM_GROUP - 1) / EK_PAM_GROUP), s);
        return;
    }
    EkPairArgs p = {};
    p.aos = aos;
    p.G = Gm;
    p.A = A;
    p.recs = recs;
    p.n_col = n_prop;
    p.K = K;
    p.held = held;
    p.old_lo = old_lo;
    p.n_old = n_old;
    p.T = T;
    p.O = O;
    p.dmin = dmin;
    if (ek_pam_pairs_form) {
        ek_pairs16_launch<0>(p, K, n_prop, n_old, s);
        return;
    }
    const int groups = (n_prop + EK_PAM_GROUP - 1) / EK_PAM_GROUP +
                       (n_old + EK_PAM_GROUP - 1) / EK_PAM_GROUP;
    ek_pairs_launch<0>(p, dim3((K + EK_WAVE - 1) / EK_WAVE, groups), s);
}

// vecs[j * n_pad + list[i]] = rmsd(frame list[i], record j), i < n_rows, j < count
void ek_launch_pam_list_dist(const float *aos, const double *G, int A,
                             const uint32_t *list, int64_t n_rows,
                             const unsigned char *recs, int count, float *vecs,
                             int64_t n_pad, hipStream_t s)
{
    if (n_rows <= 0 || count <= 0)
        return;
    EkPairArgs p = {};
    p.aos = aos;
    p.G = G;
    p.A = A;
    p.recs = recs;
    p.n_col = count;
    p.list = list;
    p.n_rows = n_rows;
    p.n_pad = n_pad;
    p.vecs = vecs;
    if (ek_pam_pairs_form) {
        ek_pairs16_launch<1>(p, n_rows, count, 0, s);
        return;
    }
    ek_pairs_launch<1>(p, dim3((unsigned)((n_rows + EK_WAVE - 1) / EK_WAVE),
                               (count + EK_PAM_GROUP - 1) / EK_PAM_GROUP), s);
}

// the frames that need exact distances -> list (any order), *n_list
// 4096 frames per workgroup, compacted in LDS: one atomic on *n_list each
#define EK_ACT_PER 16
__global__ void __launch_bounds__(EK_BLOCK)
ek_pam_active_kernel(const float *__restrict__ dist,
                     const int32_t *__restrict__ assign, int64_t n,
                     const float *__restrict__ dmin, int n_groups, int K,
                     int32_t win_lo, int32_t win_count,
                     uint32_t *__restrict__ list, unsigned int *__restrict__ n_list)
{
    __shared__ uint32_t found[EK_BLOCK * EK_ACT_PER];
    __shared__ unsigned int n_found, base_s;
    const int t = threadIdx.x;
    if (t == 0)
        n_found = 0;
    __syncthreads();
    const int64_t f0 = (int64_t)blockIdx.x * EK_BLOCK * EK_ACT_PER + t;
    int32_t a[EK_ACT_PER];
    float d[EK_ACT_PER], dm[EK_ACT_PER];
#pragma unroll
    for (int q = 0; q < EK_ACT_PER; ++q) {
        const int64_t f = f0 + (int64_t)q * EK_BLOCK;
        a[q] = (f < n) ? assign[f] : 0;
        d[q] = (f < n) ? dist[f] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < EK_ACT_PER; ++q)
        dm[q] = (a[q] >= 0 && a[q] < K) ? dmin[a[q]] : 0.f;
    for (int g = 1; g < n_groups; ++g)
#pragma unroll
        for (int q = 0; q < EK_ACT_PER; ++q)
            if (a[q] >= 0 && a[q] < K)
                dm[q] = fminf(dm[q], dmin[(size_t)g * K + a[q]]);
#pragma unroll
    for (int q = 0; q < EK_ACT_PER; ++q) {
        const int64_t f = f0 + (int64_t)q * EK_BLOCK;
        if (f >= n)
            continue;
        bool act;
        if (a[q] < 0 || a[q] >= K || (a[q] >= win_lo && a[q] < win_lo + win_count))
            act = true;
        else
            act = !(dm[q] > 2.f * d[q] * 1.001f + 1e-3f);
        if (act)
            found[atomicAdd(&n_found, 1u)] = (uint32_t)f;
    }
    __syncthreads();
    const unsigned int cnt = n_found;
    if (cnt == 0)
        return;
    if (t == 0)
        base_s = atomicAdd(n_list, cnt);
    __syncthreads();
    for (unsigned int i = t; i < cnt; i += EK_BLOCK)
        list[base_s + i] = found[i];
}

void ek_launch_pam_active(const float *dist, const int32_t *assign, int64_t n,
                          const float *dmin, int n_groups, int K, int32_t win_lo,
                          int32_t win_count, uint32_t *list, unsigned int *n_list,
                          hipStream_t s, bool cleared)
{
    if (!cleared)       // (ek_pam_setup_kernel clears it on its way)
        (void)hipMemsetAsync(n_list, 0, sizeof(unsigned int), s);
    if (n <= 0)
        return;
    const int64_t per = (int64_t)EK_BLOCK * EK_ACT_PER;
    hipLaunchKernelGGL(ek_pam_active_kernel, dim3((unsigned)((n + per - 1) / per)),
                       dim3(EK_BLOCK), 0, s, dist, assign, n, dmin, n_groups, K,
                       win_lo, win_count, list, n_list);
}

// The distance vectors are +inf except at the frames of the window before: those
// entries alone go back to +inf (instead of filling `cols` whole vectors).
__global__ void __launch_bounds__(EK_BLOCK)
ek_pam_vecs_reset_kernel(const uint32_t *__restrict__ list, int64_t n_rows, int64_t n_pad,
                         float *__restrict__ vecs)
{
    const int64_t i = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (i < n_rows)
        vecs[(size_t)blockIdx.y * n_pad + list[i]] = __builtin_inff();
}

void ek_launch_pam_vecs_reset(const uint32_t *list, int64_t n_rows, int cols,
                              int64_t n_pad, float *vecs, hipStream_t s)
{
    if (n_rows <= 0 || cols <= 0)
        return;
    hipLaunchKernelGGL(ek_pam_vecs_reset_kernel,
                       dim3((unsigned)((n_rows + EK_BLOCK - 1) / EK_BLOCK), cols),
                       dim3(EK_BLOCK), 0, s, list, n_rows, n_pad, vecs);
}

// the records of up to EK_PAM_WIN frames in one launch (block j = frame j)
struct EkFrameList {
    int64_t f[EK_PAM_WIN];
};
__global__ void __launch_bounds__(EK_BLOCK)
ek_records_from_frames_kernel(const float *__restrict__ aos,
                              const double *__restrict__ G, int A, EkFrameList fl,
                              int64_t global_offset,
                              unsigned char *__restrict__ recs)
{
    const int64_t idx = fl.f[blockIdx.x];
    EkRecHdr *h = (EkRecHdr *)(recs + (size_t)blockIdx.x * ek_rec_bytes(A));
    float *coords = (float *)(h + 1);
    if (threadIdx.x == 0) {
        h->maxdist = __builtin_inff();
        h->valid = 1;
        h->gidx = global_offset + idx;
        h->trace = G[idx];
        h->reserved = 0;
    }
    for (int r = threadIdx.x; r < 3 * A; r += EK_BLOCK)
        coords[r] = aos[(size_t)idx * 3 * A + r];
}

void ek_launch_records_from_frames(const float *aos, const double *G, int A,
                                   const int64_t *frames, int count,
                                   int64_t global_offset, unsigned char *recs,
                                   hipStream_t s)
{
    if (count <= 0)
        return;
    EkFrameList fl;
    for (int j = 0; j < EK_PAM_WIN; ++j)
        fl.f[j] = j < count ? frames[j] : 0;
    hipLaunchKernelGGL(ek_records_from_frames_kernel, dim3(count), dim3(EK_BLOCK), 0,
                       s, aos, G, A, fl, global_offset, recs);
}

// Everything a window's prefetch needs of its proposals, in one launch: the
// records (workgroup j: frame j, as above), the candidate tile / traces the
// pass kernel reads (ek_spec.hip, ek_ctile_kernel's layout [atom][pair][xyz][2],
// taken from the frames directly) and the fixed plan "distances to these
// records" for the first `cg` <= EK_PAM_GROUP of them (what a pass over all
// frames starts with when the restriction to touched frames does not apply),
// and the active-frame counter cleared.
__global__ void __launch_bounds__(EK_BLOCK)
ek_pam_setup_kernel(const float *__restrict__ aos, const double *__restrict__ G,
                    int A, EkFrameList fl, int count, int cg, int T,
                    int64_t global_offset,
                    unsigned char *__restrict__ recs, float *__restrict__ ctile,
                    double *__restrict__ ctrace, EkPlan *__restrict__ plan,
                    unsigned int *__restrict__ counter, const float *__restrict__ dist,
                    float *__restrict__ dprop, const int64_t *__restrict__ frames_dev)
{
    // frames_dev (round 5): the proposed frames where ek_select_member_multi_kernel
    // left them -- the sweep no longer waits for them to reach the host before the
    // window's set-up goes out (a frame that does not exist, -1: row 0, and the
    // host refuses the window when the list arrives)
#define EK_SETUP_FRAME(J) (frames_dev ? (frames_dev[J] < 0 ? 0 : frames_dev[J]) : fl.f[J])
    if ((int)blockIdx.x < count) {
        const int64_t idx = EK_SETUP_FRAME(blockIdx.x);
        if (dprop && threadIdx.x == 0)  // (how far the proposal is from its medoid)
            dprop[blockIdx.x] = dist[idx];
        EkRecHdr *h = (EkRecHdr *)(recs + (size_t)blockIdx.x * ek_rec_bytes(A));
        float *coords = (float *)(h + 1);
        if (threadIdx.x == 0) {
            h->maxdist = __builtin_inff();
            h->valid = 1;
            h->gidx = global_offset + idx;
            h->trace = G[idx];
            h->reserved = 0;
        }
        for (int r = threadIdx.x; r < 3 * A; r += EK_BLOCK)
            coords[r] = aos[(size_t)idx * 3 * A + r];
    }
    const int total = ek_ctile_atoms(A) * 3 * T;
    for (int j = blockIdx.x * EK_BLOCK + threadIdx.x; j < total;
         j += gridDim.x * EK_BLOCK) {
        const int a = j / (3 * T), c = (j % (3 * T)) / 3, k = j % 3;
        float v = 0.f;
        if (a < A && c < cg)
            v = aos[(size_t)EK_SETUP_FRAME(c) * 3 * A + 3 * a + k];
        ctile[ek_ctile_index(T, a, c, k)] = v;
    }
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < T)
            ctrace[threadIdx.x] = (int)threadIdx.x < cg ? G[EK_SETUP_FRAME(threadIdx.x)] : 0.0;
        if (threadIdx.x == 0) {
            plan->go = 1;
            plan->teff = cg;
            plan->label = 0;
            for (int j = 0; j < EK_PAM_GROUP; ++j)
                plan->src[j] = j;
            *counter = 0;
        }
    }
}

void ek_launch_pam_setup(const float *aos, const double *G, int A,
                         const int64_t *frames, int count, int64_t global_offset,
                         unsigned char *recs, float *ctile, double *ctrace,
                         EkPlan *plan, unsigned int *counter, hipStream_t s,
                         const float *dist, float *dprop)
{
    if (count <= 0)
        return;
    EkFrameList fl;
    for (int j = 0; j < EK_PAM_WIN; ++j)
        fl.f[j] = j < count ? frames[j] : 0;
    const int cg = std::min(count, EK_PAM_GROUP);
    const int T = ek_pass_dist_T(cg);
    const int cb = (ek_ctile_atoms(A) * 3 * T + EK_BLOCK - 1) / EK_BLOCK;
    hipLaunchKernelGGL(ek_pam_setup_kernel, dim3(std::max(count, cb)),
                       dim3(EK_BLOCK), 0, s, aos, G, A, fl, count, cg, T,
                       global_offset, recs, ctile, ctrace, plan, counter, dist, dprop,
                       (const int64_t *)nullptr);
}

// the same with the frames read on the device (frames_dev[count])
void ek_launch_pam_setup_dev(const float *aos, const double *G, int A,
                             const int64_t *frames_dev, int count, int64_t global_offset,
                             unsigned char *recs, float *ctile, double *ctrace,
                             EkPlan *plan, unsigned int *counter, hipStream_t s,
                             const float *dist, float *dprop)
{
    if (count <= 0)
        return;
    EkFrameList fl = {};
    const int cg = std::min(count, EK_PAM_GROUP);
    const int T = ek_pass_dist_T(cg);
    const int cb = (ek_ctile_atoms(A) * 3 * T + EK_BLOCK - 1) / EK_BLOCK;
    hipLaunchKernelGGL(ek_pam_setup_kernel, dim3(std::max(count, cb)),
                       dim3(EK_BLOCK), 0, s, aos, G, A, fl, count, cg, T,
                       global_offset, recs, ctile, ctrace, plan, counter, dist, dprop,
                       frames_dev);
}


