// ek_msm.hip -- MSM construction on the device: transition counts and
// row-normalisation (the secondary kernel of the north star).
//
// Replaces (reference paths relative to /root/reference):
//   assigns_to_counts     enspara/msm/transition_matrices.py:113-170
//     (+ _transitions_helper :310-321): per trajectory drop the -1 frames,
//     THEN pair state[t] with state[t+lag] (sliding window) or every lag-th
//     frame with the next (:316-319), count the pairs into a sparse matrix;
//   _row_normalize        enspara/msm/builders.py:171-204 (sparse branch
//     :188-196): T = diag(1/rowsum) * C with empty rows left at zero.
//
// Integer work, all of it hand-written here (no library sort / scan): the counts
// are a histogram over the dense n_states x n_states int32 table -- 100 MB at
// the 5 000 states of BASELINE.json configs[4], 1.6 GB at the 20 000 of
// configs[3], of 288 GB --: the -1 frames are squeezed out by a two-level prefix
// sum (per-workgroup counts, one workgroup scans them, every workgroup places
// its survivors), every transition is one atomic add on its cell, and the cells
// that are not zero leave the table in index order -- the same two-level
// compaction -- which is COO sorted by (row, col) = CSR order.  The labels may
// come from the host (ek_msm_counts) or be the ones a fit left in HBM
// (ek_msm_counts_ctx: the reference pipes result.assignments straight into
// assigns_to_counts; here they need not leave the device in between).
#include "ek_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <new>

extern int ek_set_error(int code, const char *fmt, ...);
// (ek_api.hip) what a count over a context's resident labels needs of it
void ek_ctx_msm_view(ek_ctx *c, int *device, int64_t *n, const int32_t **assign,
                     hipStream_t *stream, void ***scratch_slot);

#define MSM_HIP(call)                                                          \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                \
            rc = ek_set_error(EK_EHIP, "%s failed: %s at %s:%d", #call,        \
                              hipGetErrorString(e_), __FILE__, __LINE__);      \
            goto done;                                                         \
        }                                                                      \
    } while (0)

static inline unsigned msm_blocks(int64_t n)
{
    return (unsigned)std::max<int64_t>(1, (n + EK_BLOCK - 1) / EK_BLOCK);
}

// ---- two-level compaction: count per workgroup, scan the counts, place ----------
#define MSM_WG 1024

// cnt[b] = elements of [1024 b, 1024 b + 1024) that survive
// (MODE 0: frames that are not -1; MODE 1: table cells that are not 0)
template <int MODE>
__global__ void __launch_bounds__(MSM_WG)
msm_count_kernel(const int32_t *__restrict__ a, int64_t n,
                 int32_t *__restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * MSM_WG + threadIdx.x;
    const bool keep = i < n && (MODE == 0 ? a[i] != -1 : a[i] != 0);
    const int c = __syncthreads_count(keep);
    if (threadIdx.x == 0)
        cnt[blockIdx.x] = c;
}

// exclusive prefix sums of cnt[0 .. nb) by one workgroup -> off[0 .. nb], off[nb] = total
__global__ void __launch_bounds__(MSM_WG)
msm_scan_kernel(const int32_t *__restrict__ cnt, int64_t nb,
                int64_t *__restrict__ off)
{
    __shared__ int64_t part[MSM_WG];
    __shared__ int64_t wsum[MSM_WG / EK_WAVE];
    const int t = threadIdx.x;
    const int64_t per = (nb + MSM_WG - 1) / MSM_WG;
    const int64_t lo = (int64_t)t * per, hi = (lo + per < nb) ? lo + per : nb;
    int64_t s = 0;
    // (a thread's counts are read in batches of sixteen loads in flight: one after the
    // other, each waiting for the one before, the 24 of a 5000-state table's scan were
    // 25 us of nothing but latency -- twice per call)
    constexpr int UB = 16;
    for (int64_t b0 = lo; b0 < hi; b0 += UB) {
        int32_t v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u)
            v[u] = b0 + u < hi ? cnt[b0 + u] : 0;
#pragma unroll
        for (int u = 0; u < UB; ++u)
            s += v[u];
    }
    // scan of the 1024 partial sums: inside the waves, then over the 16 waves
    int64_t incl = s;
    const int lane = t & (EK_WAVE - 1), wv = t / EK_WAVE;
#pragma unroll
    for (int o = 1; o < EK_WAVE; o <<= 1) {
        const int64_t v = __shfl_up(incl, o, EK_WAVE);
        if (lane >= o)
            incl += v;
    }
    if (lane == EK_WAVE - 1)
        wsum[wv] = incl;
    __syncthreads();
    int64_t base = 0;
    for (int w = 0; w < wv; ++w)
        base += wsum[w];
    part[t] = base + incl - s;
    __syncthreads();
    int64_t run = part[t];
    for (int64_t b0 = lo; b0 < hi; b0 += UB) {
        int32_t v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u)
            v[u] = b0 + u < hi ? cnt[b0 + u] : 0;
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (b0 + u < hi) {
                off[b0 + u] = run;
                run += v[u];
            }
    }
    if (t == MSM_WG - 1)
        off[nb] = base + incl;
}

// position of this thread's element among the survivors of its workgroup
__device__ __forceinline__ int msm_wg_rank(bool keep)
{
    __shared__ int wcnt[MSM_WG / EK_WAVE];
    const int lane = threadIdx.x & (EK_WAVE - 1), wv = threadIdx.x / EK_WAVE;
    const unsigned long long m = __ballot(keep);
    if (lane == 0)
        wcnt[wv] = __popcll(m);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wv; ++w)
        before += wcnt[w];
    return before + __popcll(m & ((1ull << lane) - 1ull));
}

// the frames that are not -1, in order (transition_matrices.py:156)
__global__ void __launch_bounds__(MSM_WG)
msm_squeeze_kernel(const int32_t *__restrict__ a, int64_t n,
                   const int64_t *__restrict__ off, int32_t *__restrict__ c)
{
    const int64_t i = (int64_t)blockIdx.x * MSM_WG + threadIdx.x;
    const bool keep = i < n && a[i] != -1;
    const int r = msm_wg_rank(keep);
    if (keep)
        c[off[blockIdx.x] + r] = a[i];
}

// cstart[t] = survivors before the first frame of trajectory t; cstart[n_trj] = all
// (one wave per trajectory: what its workgroup of the squeeze placed before it)
__global__ void __launch_bounds__(EK_BLOCK)
msm_cstart2_kernel(const int32_t *__restrict__ a, int64_t n,
                   const int64_t *__restrict__ start,
                   const int64_t *__restrict__ off, int64_t nb, int64_t n_trj,
                   int64_t *__restrict__ cstart)
{
    const int64_t t = (int64_t)blockIdx.x * (EK_BLOCK / EK_WAVE) +
                      threadIdx.x / EK_WAVE;
    const int lane = threadIdx.x & (EK_WAVE - 1);
    if (t > n_trj)
        return;
    if (t == n_trj || start[t] >= n) {
        if (lane == 0)
            cstart[t] = off[nb];
        return;
    }
    const int64_t b = start[t] / MSM_WG;
    int before = 0;
    for (int64_t i = b * MSM_WG + lane; i < start[t]; i += EK_WAVE)
        before += (a[i] != -1) ? 1 : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        before += __shfl_xor(before, o, EK_WAVE);
    if (lane == 0)
        cstart[t] = off[b] + before;
}

// one atomic add per transition (:310-321); *bad is set if a state lies
// outside [0, n_states).  Round 6: the trajectory of a position -- which decides
// whether position + lag is still inside it -- is no longer searched by every thread
// (ten dependent loads each, beside two for the states): the workgroup's first
// position is searched once, the next MSM_HTRJ starts are staged in LDS and a thread
// counts how many of them lie at or before its position (a workgroup's 4096
// consecutive positions rarely span more; a position beyond them searches as before).
#define MSM_HTRJ 16
#define MSM_HPER 16     // positions per thread: 4096 per workgroup
// (Round 6 measured FOUR forms of this kernel on the bench's 10^7 transitions -- a
// trajectory search per thread; the search once per workgroup of 256 positions; this
// one, 4096 positions per workgroup with every state asked for up front; the additions
// gathered in LDS first, msm_hist_lds_kernel -- at 290, 291, 292 and 296 us: 3.4e10
// atomic adds per second, 1.7 x what MI355X_MICROARCH.md gives for 64 lanes in 64
// different rows.  The kernel is bound by the atomics themselves; a wave's 64
// consecutive positions of a walk fall on a dozen neighbouring cells, which is why it
// is above the scattered figure, and why gathering in a HASHED table does not pay: it
// halves the additions and flushes them in hash order, every lane on a cell of its
// own.  profiles/r06/kernel_summary_msm_*.csv)
__global__ void __launch_bounds__(EK_BLOCK)
msm_hist_kernel(const int32_t *__restrict__ c, const int64_t *__restrict__ cstart,
                int64_t n_trj, int32_t lag, int sliding, int32_t n_states,
                int32_t *__restrict__ table, int32_t *__restrict__ bad)
{
    __shared__ long long s_cs[MSM_HTRJ + 1];
    __shared__ long long s_t0;
    const int tid = threadIdx.x;
    const int64_t m = cstart[n_trj];
    const int64_t base = (int64_t)blockIdx.x * (EK_BLOCK * MSM_HPER);
    if (base >= m)
        return;
    int32_t from[MSM_HPER], to[MSM_HPER];
#pragma unroll
    for (int k = 0; k < MSM_HPER; ++k) {
        const int64_t p = base + tid + (int64_t)k * EK_BLOCK;
        from[k] = p < m ? c[p] : 0;
        to[k] = p + lag < m ? c[p + lag] : 0;
    }
    if (tid == 0) {
        // trajectory of position `base`: last t with cstart[t] <= base
        int64_t lo = 0, hi = n_trj - 1;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (cstart[mid] <= base)
                lo = mid;
            else
                hi = mid - 1;
        }
        s_t0 = lo;
    }
    __syncthreads();
    const int64_t t0 = s_t0;
    if (tid <= MSM_HTRJ)
        s_cs[tid] = t0 + tid <= n_trj ? cstart[t0 + tid] : m;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MSM_HPER; ++k) {
        const int64_t p = base + tid + (int64_t)k * EK_BLOCK;
        if (p >= m)
            continue;
        if (from[k] < 0 || from[k] >= n_states) {
            *bad = 1;
            continue;
        }
        int64_t ts, te;
        if (p < s_cs[MSM_HTRJ]) {
            int j = 0;
#pragma unroll
            for (int u = 1; u < MSM_HTRJ; ++u)
                j += (s_cs[u] <= p) ? 1 : 0;
            ts = s_cs[j];
            te = s_cs[j + 1];
        } else {
            int64_t lo = t0, hi = n_trj - 1;
            while (lo < hi) {
                const int64_t mid = (lo + hi + 1) >> 1;
                if (cstart[mid] <= p)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            ts = cstart[lo];
            te = cstart[lo + 1];
        }
        bool ok = (p + lag < te);
        if (ok && !sliding)
            ok = ((p - ts) % lag) == 0;
        if (!ok)
            continue;
        if (to[k] < 0 || to[k] >= n_states) {       // (p + lag < te <= m: it was loaded)
            *bad = 1;
            continue;
        }
        atomicAdd(&table[(size_t)from[k] * n_states + to[k]], 1);
    }
}

// The same counts with the additions of a stretch of the walk gathered in LDS first
// (round 6).  A workgroup takes MSM_HCH consecutive transitions; a trajectory stays in
// a neighbourhood of states for a while, so they hit far fewer different cells than
// there are of them (the bench's walk: 3650 cells per 8192 transitions) -- an
// open-addressed table in LDS (key = cell + 1, linear probing, atomicCAS + atomicAdd)
// takes the additions, and what leaves the workgroup is ONE global atomic add per
// different cell, with its count.  A transition that finds no slot within
// MSM_HPROBE probes (transitions all over the table: nothing to gather) goes to the
// global table directly, as in msm_hist_kernel.  The trajectory of a position: the
// workgroup's first by one search, the next sixteen starts from LDS.  Integer adds
// commute: the table is msm_hist_kernel's.
// MEASURED (profiles/r06/kernel_summary_msm_*.csv), and not the default: 296 us against
// 290 for one atomic per transition on the bench's 10^7 transitions, 8 % slower on
// transitions all over the table: half the global additions, but flushed in hash
// order -- 64 lanes on 64 cells of their own, the slowest shape an atomic has --
// where a wave of msm_hist_kernel falls on a dozen neighbouring cells.
// EK_MSM_HIST_LDS=1 runs this form.
#define MSM_HCH 8192
#define MSM_HSLOTS 8192
#define MSM_HPROBE 6
__global__ void __launch_bounds__(EK_BLOCK)
msm_hist_lds_kernel(const int32_t *__restrict__ c, const int64_t *__restrict__ cstart,
                    int64_t n_trj, int32_t lag, int sliding, int32_t n_states,
                    int32_t *__restrict__ table, int32_t *__restrict__ bad)
{
    __shared__ unsigned int keys[MSM_HSLOTS];
    __shared__ unsigned int vals[MSM_HSLOTS];
    __shared__ long long s_cs[MSM_HTRJ + 1];
    __shared__ long long s_t0;
    const int tid = threadIdx.x;
    const int64_t m = cstart[n_trj];
    const int64_t base = (int64_t)blockIdx.x * MSM_HCH;
    if (base >= m)
        return;
    for (int q = tid; q < MSM_HSLOTS; q += EK_BLOCK) {
        keys[q] = 0u;
        vals[q] = 0u;
    }
    if (tid == 0) {
        // trajectory of position `base`: last t with cstart[t] <= base
        int64_t lo = 0, hi = n_trj - 1;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (cstart[mid] <= base)
                lo = mid;
            else
                hi = mid - 1;
        }
        s_t0 = lo;
    }
    __syncthreads();
    const int64_t t0 = s_t0;
    if (tid <= MSM_HTRJ)
        s_cs[tid] = t0 + tid <= n_trj ? cstart[t0 + tid] : m;
    __syncthreads();
    for (int k = 0; k < MSM_HCH / EK_BLOCK; ++k) {
        const int64_t p = base + tid + (int64_t)k * EK_BLOCK;
        if (p >= m)
            break;
        const int32_t from = c[p];
        if (from < 0 || from >= n_states) {
            *bad = 1;
            continue;
        }
        // start and end of p's trajectory: among the staged starts, else by search
        int64_t ts, te;
        if (p < s_cs[MSM_HTRJ]) {
            int j = 0;
#pragma unroll
            for (int u = 1; u < MSM_HTRJ; ++u)
                j += (s_cs[u] <= p) ? 1 : 0;
            ts = s_cs[j];
            te = s_cs[j + 1];
        } else {
            int64_t lo = t0, hi = n_trj - 1;
            while (lo < hi) {
                const int64_t mid = (lo + hi + 1) >> 1;
                if (cstart[mid] <= p)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            ts = cstart[lo];
            te = cstart[lo + 1];
        }
        bool ok = (p + lag < te);
        if (ok && !sliding)
            ok = ((p - ts) % lag) == 0;
        if (!ok)
            continue;
        const int32_t to = c[p + lag];
        if (to < 0 || to >= n_states) {
            *bad = 1;
            continue;
        }
        const unsigned int cell = (unsigned int)from * (unsigned int)n_states + (unsigned int)to;
        unsigned int h = (cell * 2654435761u) >> (32 - 13);
        bool placed = false;
#pragma unroll 1
        for (int pr = 0; pr < MSM_HPROBE; ++pr) {
            const unsigned int old = atomicCAS(&keys[h], 0u, cell + 1u);
            if (old == 0u || old == cell + 1u) {
                atomicAdd(&vals[h], 1u);
                placed = true;
                break;
            }
            h = (h + 1u) & (MSM_HSLOTS - 1);
        }
        if (!placed)
            atomicAdd(&table[cell], 1);
    }
    __syncthreads();
    for (int q = tid; q < MSM_HSLOTS; q += EK_BLOCK)
        if (keys[q])
            atomicAdd(&table[keys[q] - 1u], (int32_t)vals[q]);
}
static_assert(MSM_HSLOTS == (1 << 13), "the hash keeps 13 bits");

// the cells that are not zero, in index order
__global__ void __launch_bounds__(MSM_WG)
msm_cells_kernel(const int32_t *__restrict__ table, int64_t n_cells,
                 int32_t n_states, const int64_t *__restrict__ off,
                 int64_t capacity, int32_t *__restrict__ rows,
                 int32_t *__restrict__ cols, int64_t *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * MSM_WG + threadIdx.x;
    const int32_t v = i < n_cells ? table[i] : 0;
    const int r = msm_wg_rank(v != 0);
    if (v != 0) {
        const int64_t o = off[blockIdx.x] + r;
        if (o < capacity) {
            rows[o] = (int32_t)(i / n_states);
            cols[o] = (int32_t)(i % n_states);
            vals[o] = v;
        }
    }
}

// scratch of a count, kept between calls by a context (hipMalloc costs more than
// the kernels)
struct EkMsmScratch {
    int32_t *c = nullptr;       // squeezed labels [n + lag]
    int32_t *cnt = nullptr;     // per-workgroup counts
    int64_t *off = nullptr;     // their prefix sums
    int64_t *start = nullptr, *cstart = nullptr;    // [n_trj + 1]
    int32_t *table = nullptr;   // [n_states^2]
    int32_t *bad = nullptr;
    int32_t *rows = nullptr, *cols = nullptr;
    int64_t *vals = nullptr;
    int64_t cap_c = 0, cap_blocks = 0, cap_trj = 0, cap_cells = 0, cap_out = 0;
};

static void msm_scratch_release(EkMsmScratch *w)
{
    if (!w)
        return;
    (void)hipFree(w->c);
    (void)hipFree(w->cnt);
    (void)hipFree(w->off);
    (void)hipFree(w->start);
    (void)hipFree(w->cstart);
    (void)hipFree(w->table);
    (void)hipFree(w->bad);
    (void)hipFree(w->rows);
    (void)hipFree(w->cols);
    (void)hipFree(w->vals);
    *w = EkMsmScratch();
}

void ek_msm_scratch_free(void *w)
{
    if (w) {
        msm_scratch_release((EkMsmScratch *)w);
        delete (EkMsmScratch *)w;
    }
}

template <typename P>
static hipError_t msm_grow(P *&ptr, int64_t &cap, int64_t need)
{
    if (need <= cap)
        return hipSuccess;
    (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    const hipError_t e = hipMalloc((void **)&ptr, (size_t)need * sizeof(*ptr));
    if (e == hipSuccess)
        cap = need;
    return e;
}

// d_a: the labels on the device, one per frame; h_start[t]: first frame of
// trajectory t (h_start[n_trj] = n)
static int msm_counts_device(hipStream_t s, EkMsmScratch &w, const int32_t *d_a,
                             const int64_t *h_start, int64_t n, int64_t n_trj,
                             int32_t lag_time, int32_t sliding_window,
                             int32_t n_states, int64_t capacity, int32_t *rows_out,
                             int32_t *cols_out, int64_t *counts_out,
                             int64_t *nnz_out)
{
    int rc = EK_OK;
    const int64_t n_cells = (int64_t)n_states * n_states;
    const int64_t nb = (n + MSM_WG - 1) / MSM_WG;
    const int64_t ncb = (n_cells + MSM_WG - 1) / MSM_WG;
    const int64_t cap = std::max<int64_t>(
        1, std::min<int64_t>(capacity, std::min<int64_t>(n, n_cells)));
    int32_t bad = 0;
    int64_t nnz = 0;
    {
        int64_t cap_rows = w.cap_out, cap_cols = w.cap_out, cap_vals = w.cap_out;
        int64_t cap_cnt = w.cap_blocks, cap_off = w.cap_blocks ? w.cap_blocks + 1 : 0;
        int64_t cap_start = w.cap_trj, cap_cstart = w.cap_trj, cap_bad = w.bad ? 1 : 0;
        hipError_t e = msm_grow(w.c, w.cap_c, n + lag_time);
        if (e == hipSuccess)
            e = msm_grow(w.cnt, cap_cnt, std::max(nb, ncb));
        if (e == hipSuccess)
            e = msm_grow(w.off, cap_off, std::max(nb, ncb) + 1);
        if (e == hipSuccess)
            e = msm_grow(w.start, cap_start, n_trj + 1);
        if (e == hipSuccess)
            e = msm_grow(w.cstart, cap_cstart, n_trj + 1);
        if (e == hipSuccess)
            e = msm_grow(w.table, w.cap_cells, n_cells);
        if (e == hipSuccess)
            e = msm_grow(w.bad, cap_bad, 1);
        if (e == hipSuccess)
            e = msm_grow(w.rows, cap_rows, cap);
        if (e == hipSuccess)
            e = msm_grow(w.cols, cap_cols, cap);
        if (e == hipSuccess)
            e = msm_grow(w.vals, cap_vals, cap);
        w.cap_blocks = std::min(cap_cnt, cap_off ? cap_off - 1 : 0);
        w.cap_trj = std::min(cap_start, cap_cstart);
        w.cap_out = std::min(cap_rows, std::min(cap_cols, cap_vals));
        if (e != hipSuccess)
            return ek_set_error(e == hipErrorOutOfMemory ? EK_ENOMEM : EK_EHIP,
                                "ek_msm_counts: %s (the count table of %d states "
                                "is %.1f GB)", hipGetErrorString(e), n_states,
                                (double)n_cells * 4e-9);
    }
    MSM_HIP(hipMemcpyAsync(w.start, h_start, (size_t)(n_trj + 1) * sizeof(int64_t),
                           hipMemcpyHostToDevice, s));
    MSM_HIP(hipMemsetAsync(w.table, 0, (size_t)n_cells * sizeof(int32_t), s));
    MSM_HIP(hipMemsetAsync(w.bad, 0, sizeof(int32_t), s));
    // 1. drop the -1 frames
    hipLaunchKernelGGL(msm_count_kernel<0>, dim3((unsigned)nb), dim3(MSM_WG), 0, s,
                       d_a, n, w.cnt);
    hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(MSM_WG), 0, s, w.cnt, nb,
                       w.off);
    hipLaunchKernelGGL(msm_squeeze_kernel, dim3((unsigned)nb), dim3(MSM_WG), 0, s,
                       d_a, n, w.off, w.c);
    hipLaunchKernelGGL(msm_cstart2_kernel,
                       dim3(msm_blocks((n_trj + 1) * EK_WAVE)), dim3(EK_BLOCK), 0,
                       s, d_a, n, w.start, w.off, nb, n_trj, w.cstart);
    // 2. the histogram (a launch over all frames; the survivors' count stays on
    //    the device)
    //    (EK_MSM_HIST_LDS=1: the additions gathered per stretch of the walk in LDS
    //    first -- built in round 6 and measured not to pay, see msm_hist_lds_kernel)
    {
        static const int use_lds = [] {
            const char *e = getenv("EK_MSM_HIST_LDS");
            return (e && e[0] == '1') ? 1 : 0;
        }();
        if (use_lds && n_cells < 0xffffffffll)
            hipLaunchKernelGGL(msm_hist_lds_kernel,
                               dim3((unsigned)((n + MSM_HCH - 1) / MSM_HCH)), dim3(EK_BLOCK),
                               0, s, w.c, w.cstart, n_trj, lag_time, sliding_window,
                               n_states, w.table, w.bad);
        else
            hipLaunchKernelGGL(msm_hist_kernel,
                               dim3((unsigned)((n + EK_BLOCK * MSM_HPER - 1) /
                                               (EK_BLOCK * MSM_HPER))),
                               dim3(EK_BLOCK), 0, s, w.c, w.cstart, n_trj, lag_time,
                               sliding_window, n_states, w.table, w.bad);
    }
    // 3. the cells that are not zero, in (row, col) order
    hipLaunchKernelGGL(msm_count_kernel<1>, dim3((unsigned)ncb), dim3(MSM_WG), 0, s,
                       w.table, n_cells, w.cnt);
    hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(MSM_WG), 0, s, w.cnt, ncb,
                       w.off);
    hipLaunchKernelGGL(msm_cells_kernel, dim3((unsigned)ncb), dim3(MSM_WG), 0, s,
                       w.table, n_cells, n_states, w.off, cap, w.rows, w.cols,
                       w.vals);
    MSM_HIP(hipGetLastError());
    MSM_HIP(hipMemcpyAsync(&nnz, w.off + ncb, sizeof(int64_t), hipMemcpyDeviceToHost,
                           s));
    MSM_HIP(hipMemcpyAsync(&bad, w.bad, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MSM_HIP(hipStreamSynchronize(s));
    if (bad)
        return ek_set_error(EK_EARG, "ek_msm_counts: a state lies outside [0, %d)",
                            n_states);
    if (nnz > capacity)
        return ek_set_error(EK_EARG, "ek_msm_counts: %lld entries exceed the "
                                     "capacity %lld", (long long)nnz,
                            (long long)capacity);
    if (nnz > 0) {
        MSM_HIP(hipMemcpyAsync(rows_out, w.rows, (size_t)nnz * sizeof(int32_t),
                               hipMemcpyDeviceToHost, s));
        MSM_HIP(hipMemcpyAsync(cols_out, w.cols, (size_t)nnz * sizeof(int32_t),
                               hipMemcpyDeviceToHost, s));
        MSM_HIP(hipMemcpyAsync(counts_out, w.vals, (size_t)nnz * sizeof(int64_t),
                               hipMemcpyDeviceToHost, s));
        MSM_HIP(hipStreamSynchronize(s));
    }
    *nnz_out = nnz;
done:
    return rc;
}

// -> starts[n_trj + 1] (malloc'd), *n_out = frames in all; nullptr on a bad length
static int64_t *msm_starts(const int64_t *lengths, int64_t n_trj, int64_t *n_out)
{
    int64_t *starts = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_trj + 1));
    if (!starts)
        return nullptr;
    starts[0] = 0;
    for (int64_t t = 0; t < n_trj; ++t) {
        if (lengths[t] < 0) {
            free(starts);
            return nullptr;
        }
        starts[t + 1] = starts[t] + lengths[t];
    }
    *n_out = starts[n_trj];
    return starts;
}

extern "C" int ek_msm_counts(int device, const int32_t *assigns,
                             const int64_t *lengths, int64_t n_trj,
                             int32_t lag_time, int32_t sliding_window,
                             int32_t n_states, int64_t capacity,
                             int32_t *rows_out, int32_t *cols_out,
                             int64_t *counts_out, int64_t *nnz_out)
{
    int rc = EK_OK;
    if (!lengths || !nnz_out || n_trj < 0 || lag_time < 1 || n_states < 1)
        return ek_set_error(EK_EARG, "ek_msm_counts: bad argument");
    int64_t n = 0;
    int64_t *starts = msm_starts(lengths, n_trj, &n);
    if (!starts)
        return ek_set_error(EK_EARG, "ek_msm_counts: negative length (or out of "
                                     "host memory)");
    *nnz_out = 0;
    if (n == 0 || n_trj == 0) {
        free(starts);
        return EK_OK;
    }
    EkMsmScratch w;
    int32_t *d_a = nullptr;
    hipStream_t s = nullptr;
    if (!assigns) {
        rc = ek_set_error(EK_EARG, "ek_msm_counts: assigns is NULL");
        goto done;
    }
    if (n >= ((int64_t)1 << 31)) {
        rc = ek_set_error(EK_EARG, "ek_msm_counts: %lld frames: a cell of the int32 "
                                   "table could overflow", (long long)n);
        goto done;
    }
    MSM_HIP(hipSetDevice(device));
    MSM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    MSM_HIP(hipMalloc((void **)&d_a, (size_t)n * sizeof(int32_t)));
    MSM_HIP(hipMemcpyAsync(d_a, assigns, (size_t)n * sizeof(int32_t),
                           hipMemcpyHostToDevice, s));
    rc = msm_counts_device(s, w, d_a, starts, n, n_trj, lag_time, sliding_window,
                           n_states, capacity, rows_out, cols_out, counts_out,
                           nnz_out);
done:
    if (s)
        (void)hipStreamSynchronize(s);
    (void)hipFree(d_a);
    msm_scratch_release(&w);
    free(starts);
    if (s)
        (void)hipStreamDestroy(s);
    return rc;
}

// the labels a fit left in the context's HBM (dist / assign of its n frames, in
// frame order), `lengths` saying how the frames split into trajectories
extern "C" int ek_msm_counts_ctx(ek_ctx *ctx, const int64_t *lengths, int64_t n_trj,
                                 int32_t lag_time, int32_t sliding_window,
                                 int32_t n_states, int64_t capacity,
                                 int32_t *rows_out, int32_t *cols_out,
                                 int64_t *counts_out, int64_t *nnz_out)
{
    if (!ctx || !lengths || !nnz_out || n_trj < 0 || lag_time < 1 || n_states < 1)
        return ek_set_error(EK_EARG, "ek_msm_counts_ctx: bad argument");
    int device = 0;
    int64_t n_ctx = 0;
    const int32_t *d_a = nullptr;
    hipStream_t s = nullptr;
    void **slot = nullptr;
    ek_ctx_msm_view(ctx, &device, &n_ctx, &d_a, &s, &slot);
    int64_t n = 0;
    int64_t *starts = msm_starts(lengths, n_trj, &n);
    if (!starts)
        return ek_set_error(EK_EARG, "ek_msm_counts_ctx: negative length (or out of "
                                     "host memory)");
    int rc = EK_OK;
    *nnz_out = 0;
    if (n != n_ctx) {
        rc = ek_set_error(EK_EARG, "ek_msm_counts_ctx: the lengths add up to %lld "
                                   "frames, the context holds %lld", (long long)n,
                          (long long)n_ctx);
    } else if (n > 0 && n_trj > 0) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) {
            rc = ek_set_error(EK_EHIP, "hipSetDevice(%d): %s", device,
                              hipGetErrorString(e));
        } else {
            if (!*slot)
                *slot = new (std::nothrow) EkMsmScratch();
            if (!*slot)
                rc = ek_set_error(EK_ENOMEM, "ek_msm_counts_ctx: out of host memory");
            else
                rc = msm_counts_device(s, *(EkMsmScratch *)*slot, d_a, starts, n,
                                       n_trj, lag_time, sliding_window, n_states,
                                       capacity, rows_out, cols_out, counts_out,
                                       nnz_out);
        }
    }
    free(starts);
    return rc;
}

// ---- row normalisation of a CSR matrix ------------------------------------------
// one wave per row; the row sum is accumulated sequentially in storage order
// by lane 0 (the order scipy's csr row sum uses), then inv = 1/sum once and
// T = inv * C element-wise (builders.py:190-195: diag(inv_weights).dot(C))
__global__ void __launch_bounds__(EK_BLOCK)
msm_rownorm_kernel(const int64_t *__restrict__ indptr,
                   const double *__restrict__ data, int64_t n_rows,
                   double *__restrict__ out, double *__restrict__ rowsum)
{
    const int64_t row = (int64_t)blockIdx.x * (EK_BLOCK / EK_WAVE) +
                        threadIdx.x / EK_WAVE;
    const int lane = threadIdx.x & (EK_WAVE - 1);
    if (row >= n_rows)
        return;
    const int64_t lo = indptr[row], hi = indptr[row + 1];
    double w = 0.0;
    if (lane == 0)
        for (int64_t j = lo; j < hi; ++j)
            w = w + data[j];
    w = __shfl(w, 0, 64);
    const double inv = (w > 0.0) ? 1.0 / w : 0.0;
    for (int64_t j = lo + lane; j < hi; j += EK_WAVE)
        out[j] = inv * data[j];
    if (lane == 0 && rowsum)
        rowsum[row] = w;
}

extern "C" int ek_msm_row_normalize(int device, const int64_t *indptr,
                                    const double *data, int64_t n_rows,
                                    double *probs_out, double *rowsum_out)
{
    int rc = EK_OK;
    if (!indptr || n_rows < 0 || (!probs_out))
        return ek_set_error(EK_EARG, "ek_msm_row_normalize: bad argument");
    if (n_rows == 0)
        return EK_OK;
    const int64_t nnz = indptr[n_rows];
    int64_t *d_ip = nullptr;
    double *d_in = nullptr, *d_out = nullptr, *d_rs = nullptr;
    hipStream_t s = nullptr;
    {
        hipError_t e0 = hipSetDevice(device);
        if (e0 != hipSuccess)
            return ek_set_error(EK_EHIP, "hipSetDevice(%d): %s", device,
                                hipGetErrorString(e0));
    }
    MSM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    MSM_HIP(hipMalloc((void **)&d_ip, (size_t)(n_rows + 1) * sizeof(int64_t)));
    MSM_HIP(hipMalloc((void **)&d_in, (size_t)std::max<int64_t>(nnz, 1) * sizeof(double)));
    MSM_HIP(hipMalloc((void **)&d_out, (size_t)std::max<int64_t>(nnz, 1) * sizeof(double)));
    MSM_HIP(hipMalloc((void **)&d_rs, (size_t)n_rows * sizeof(double)));
    MSM_HIP(hipMemcpyAsync(d_ip, indptr, (size_t)(n_rows + 1) * sizeof(int64_t),
                           hipMemcpyHostToDevice, s));
    if (nnz > 0)
        MSM_HIP(hipMemcpyAsync(d_in, data, (size_t)nnz * sizeof(double),
                               hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(msm_rownorm_kernel,
                       dim3((unsigned)((n_rows + 3) / 4)), dim3(EK_BLOCK), 0, s,
                       d_ip, d_in, n_rows, d_out, d_rs);
    if (nnz > 0)
        MSM_HIP(hipMemcpyAsync(probs_out, d_out, (size_t)nnz * sizeof(double),
                               hipMemcpyDeviceToHost, s));
    if (rowsum_out)
        MSM_HIP(hipMemcpyAsync(rowsum_out, d_rs, (size_t)n_rows * sizeof(double),
                               hipMemcpyDeviceToHost, s));
    MSM_HIP(hipStreamSynchronize(s));
done:
    if (s)
        (void)hipStreamSynchronize(s);
    (void)hipFree(d_ip);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    (void)hipFree(d_rs);
    if (s)
        (void)hipStreamDestroy(s);
    return rc;
}
