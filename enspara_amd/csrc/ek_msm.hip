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
// Integer work.  While the dense n_states x n_states table fits (<= 2^28 cells:
// 16 384 states, 1 GiB of int32) the counts are a histogram: the -1 frames are
// squeezed out by a two-level prefix sum (per-workgroup counts, one workgroup
// scans them, every workgroup places its survivors), every transition is one
// atomic add on its cell, and the cells that are not zero leave the table in
// index order -- the same two-level compaction -- which is COO sorted by
// (row, col) = CSR order.  Beyond that size: one 64-bit key
// (row * n_states + col) per transition, radix sort + run-length encode
// (rocPRIM through hipCUB).
#include "ek_common.h"

#include <hipcub/hipcub.hpp>

#include <stdarg.h>
#include <stdio.h>

#include <algorithm>

extern int ek_set_error(int code, const char *fmt, ...);

#define MSM_HIP(call)                                                          \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                \
            rc = ek_set_error(EK_EHIP, "%s failed: %s at %s:%d", #call,        \
                              hipGetErrorString(e_), __FILE__, __LINE__);      \
            goto done;                                                         \
        }                                                                      \
    } while (0)

__global__ void __launch_bounds__(EK_BLOCK)
msm_valid_kernel(const int32_t *__restrict__ a, int64_t n,
                 int32_t *__restrict__ valid)
{
    const int64_t i = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (i < n)
        valid[i] = (a[i] != -1) ? 1 : 0;
}

__global__ void __launch_bounds__(EK_BLOCK)
msm_compact_kernel(const int32_t *__restrict__ a,
                   const int64_t *__restrict__ pos, int64_t n,
                   int32_t *__restrict__ c)
{
    const int64_t i = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (i < n && a[i] != -1)
        c[pos[i]] = a[i];
}

// compacted start of every trajectory: cstart[t] = pos[start[t]]
__global__ void __launch_bounds__(EK_BLOCK)
msm_cstart_kernel(const int64_t *__restrict__ start,
                  const int64_t *__restrict__ pos, int64_t n_trj, int64_t n,
                  int64_t total_valid, int64_t *__restrict__ cstart)
{
    const int64_t t = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (t < n_trj)
        cstart[t] = (start[t] < n) ? pos[start[t]] : total_valid;
    if (t == n_trj)
        cstart[t] = total_valid;
}

__global__ void __launch_bounds__(EK_BLOCK)
msm_keys_kernel(const int32_t *__restrict__ c,
                const int64_t *__restrict__ cstart, int64_t n_trj,
                int64_t m, int32_t lag, int sliding, int64_t n_states,
                unsigned long long *__restrict__ keys)
{
    const int64_t p = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (p >= m)
        return;
    // trajectory of compacted position p: last t with cstart[t] <= p
    int64_t lo = 0, hi = n_trj - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (cstart[mid] <= p)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int64_t end = cstart[lo + 1];
    const int64_t local = p - cstart[lo];
    bool ok = (p + lag < end);
    if (ok && !sliding)
        ok = (local % lag) == 0;
    keys[p] = ok ? (unsigned long long)c[p] * (unsigned long long)n_states +
                       (unsigned long long)c[p + lag]
                 : ~0ull;
}

__global__ void __launch_bounds__(EK_BLOCK)
msm_split_kernel(const unsigned long long *__restrict__ ukeys,
                 const int64_t *__restrict__ ucnt, int64_t runs,
                 int64_t n_states, int32_t *__restrict__ rows,
                 int32_t *__restrict__ cols, int64_t *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (i >= runs)
        return;
    const unsigned long long k = ukeys[i];
    if (k == ~0ull) {
        rows[i] = -1;
        cols[i] = -1;
        vals[i] = 0;
        return;
    }
    rows[i] = (int32_t)(k / (unsigned long long)n_states);
    cols[i] = (int32_t)(k % (unsigned long long)n_states);
    vals[i] = ucnt[i];
}


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
    for (int64_t b = lo; b < hi; ++b)
        s += cnt[b];
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
    for (int64_t b = lo; b < hi; ++b) {
        off[b] = run;
        run += cnt[b];
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
// outside [0, n_states)
__global__ void __launch_bounds__(EK_BLOCK)
msm_hist_kernel(const int32_t *__restrict__ c, const int64_t *__restrict__ cstart,
                int64_t n_trj, int32_t lag, int sliding, int32_t n_states,
                int32_t *__restrict__ table, int32_t *__restrict__ bad)
{
    const int64_t m = cstart[n_trj];
    const int64_t p = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (p >= m)
        return;
    const int32_t from = c[p];
    if (from < 0 || from >= n_states) {
        *bad = 1;
        return;
    }
    // trajectory of compacted position p: last t with cstart[t] <= p
    int64_t lo = 0, hi = n_trj - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (cstart[mid] <= p)
            lo = mid;
        else
            hi = mid - 1;
    }
    bool ok = (p + lag < cstart[lo + 1]);
    if (ok && !sliding)
        ok = ((p - cstart[lo]) % lag) == 0;
    if (!ok)
        return;
    const int32_t to = c[p + lag];
    if (to < 0 || to >= n_states) {
        *bad = 1;
        return;
    }
    atomicAdd(&table[(size_t)from * n_states + to], 1);
}

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

#define MSM_DENSE_CELLS ((int64_t)1 << 28)

static int msm_counts_dense(int device, const int32_t *assigns,
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
    const int64_t cap = std::min<int64_t>(capacity, std::min<int64_t>(n, n_cells));
    int32_t *d_a = nullptr, *d_c = nullptr, *d_cnt = nullptr, *d_table = nullptr;
    int32_t *d_rows = nullptr, *d_cols = nullptr, *d_bad = nullptr;
    int64_t *d_off = nullptr, *d_start = nullptr, *d_cstart = nullptr,
            *d_vals = nullptr;
    hipStream_t s = nullptr;
    int32_t bad = 0;
    int64_t nnz = 0;
    MSM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    MSM_HIP(hipMalloc((void **)&d_a, (size_t)n * sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_c, (size_t)(n + lag_time) * sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_cnt,
                      (size_t)std::max(nb, ncb) * sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_off,
                      (size_t)(std::max(nb, ncb) + 1) * sizeof(int64_t)));
    MSM_HIP(hipMalloc((void **)&d_start, (size_t)(n_trj + 1) * sizeof(int64_t)));
    MSM_HIP(hipMalloc((void **)&d_cstart, (size_t)(n_trj + 1) * sizeof(int64_t)));
    MSM_HIP(hipMalloc((void **)&d_table, (size_t)n_cells * sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_bad, sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_rows, (size_t)std::max<int64_t>(cap, 1) * sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_cols, (size_t)std::max<int64_t>(cap, 1) * sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_vals, (size_t)std::max<int64_t>(cap, 1) * sizeof(int64_t)));
    MSM_HIP(hipMemcpyAsync(d_a, assigns, (size_t)n * sizeof(int32_t),
                           hipMemcpyHostToDevice, s));
    MSM_HIP(hipMemcpyAsync(d_start, h_start, (size_t)(n_trj + 1) * sizeof(int64_t),
                           hipMemcpyHostToDevice, s));
    MSM_HIP(hipMemsetAsync(d_table, 0, (size_t)n_cells * sizeof(int32_t), s));
    MSM_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), s));
    // 1. drop the -1 frames
    hipLaunchKernelGGL(msm_count_kernel<0>, dim3((unsigned)nb), dim3(MSM_WG), 0, s,
                       d_a, n, d_cnt);
    hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(MSM_WG), 0, s, d_cnt, nb,
                       d_off);
    hipLaunchKernelGGL(msm_squeeze_kernel, dim3((unsigned)nb), dim3(MSM_WG), 0, s,
                       d_a, n, d_off, d_c);
    hipLaunchKernelGGL(msm_cstart2_kernel,
                       dim3(msm_blocks((n_trj + 1) * EK_WAVE)), dim3(EK_BLOCK), 0,
                       s, d_a, n, d_start, d_off, nb, n_trj, d_cstart);
    // 2. the histogram (a launch over all frames; the survivors' count stays on
    //    the device)
    hipLaunchKernelGGL(msm_hist_kernel, dim3(msm_blocks(n)), dim3(EK_BLOCK), 0, s,
                       d_c, d_cstart, n_trj, lag_time, sliding_window, n_states,
                       d_table, d_bad);
    // 3. the cells that are not zero, in (row, col) order
    hipLaunchKernelGGL(msm_count_kernel<1>, dim3((unsigned)ncb), dim3(MSM_WG), 0, s,
                       d_table, n_cells, d_cnt);
    hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(MSM_WG), 0, s, d_cnt, ncb,
                       d_off);
    hipLaunchKernelGGL(msm_cells_kernel, dim3((unsigned)ncb), dim3(MSM_WG), 0, s,
                       d_table, n_cells, n_states, d_off, cap, d_rows, d_cols,
                       d_vals);
    MSM_HIP(hipGetLastError());
    MSM_HIP(hipMemcpyAsync(&nnz, d_off + ncb, sizeof(int64_t), hipMemcpyDeviceToHost,
                           s));
    MSM_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MSM_HIP(hipStreamSynchronize(s));
    if (bad) {
        rc = ek_set_error(EK_EARG, "ek_msm_counts: a state lies outside [0, %d)",
                          n_states);
        goto done;
    }
    if (nnz > capacity) {
        rc = ek_set_error(EK_EARG, "ek_msm_counts: %lld entries exceed the "
                                   "capacity %lld", (long long)nnz,
                          (long long)capacity);
        goto done;
    }
    if (nnz > 0) {
        MSM_HIP(hipMemcpyAsync(rows_out, d_rows, (size_t)nnz * sizeof(int32_t),
                               hipMemcpyDeviceToHost, s));
        MSM_HIP(hipMemcpyAsync(cols_out, d_cols, (size_t)nnz * sizeof(int32_t),
                               hipMemcpyDeviceToHost, s));
        MSM_HIP(hipMemcpyAsync(counts_out, d_vals, (size_t)nnz * sizeof(int64_t),
                               hipMemcpyDeviceToHost, s));
        MSM_HIP(hipStreamSynchronize(s));
    }
    *nnz_out = nnz;
done:
    if (s)
        (void)hipStreamSynchronize(s);
    (void)hipFree(d_a);
    (void)hipFree(d_c);
    (void)hipFree(d_cnt);
    (void)hipFree(d_off);
    (void)hipFree(d_start);
    (void)hipFree(d_cstart);
    (void)hipFree(d_table);
    (void)hipFree(d_bad);
    (void)hipFree(d_rows);
    (void)hipFree(d_cols);
    (void)hipFree(d_vals);
    if (s)
        (void)hipStreamDestroy(s);
    return rc;
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
    for (int64_t t = 0; t < n_trj; ++t) {
        if (lengths[t] < 0)
            return ek_set_error(EK_EARG, "ek_msm_counts: negative length");
        n += lengths[t];
    }
    *nnz_out = 0;
    if (n == 0 || n_trj == 0)
        return EK_OK;
    if (!assigns)
        return ek_set_error(EK_EARG, "ek_msm_counts: assigns is NULL");
    if ((int64_t)n_states * n_states <= MSM_DENSE_CELLS && n < ((int64_t)1 << 31)) {
        // the histogram form (int32 cells: a count cannot exceed the frames)
        hipError_t e0 = hipSetDevice(device);
        if (e0 != hipSuccess)
            return ek_set_error(EK_EHIP, "hipSetDevice(%d): %s", device,
                                hipGetErrorString(e0));
        int64_t *starts = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_trj + 1));
        if (!starts)
            return ek_set_error(EK_ENOMEM, "ek_msm_counts: out of host memory");
        starts[0] = 0;
        for (int64_t t = 0; t < n_trj; ++t)
            starts[t + 1] = starts[t] + lengths[t];
        rc = msm_counts_dense(device, assigns, starts, n, n_trj, lag_time,
                              sliding_window, n_states, capacity, rows_out,
                              cols_out, counts_out, nnz_out);
        free(starts);
        return rc;
    }

    int32_t *d_a = nullptr, *d_valid = nullptr, *d_c = nullptr;
    int64_t *d_pos = nullptr, *d_start = nullptr, *d_cstart = nullptr;
    unsigned long long *d_keys = nullptr, *d_keys2 = nullptr, *d_ukeys = nullptr;
    int64_t *d_ucnt = nullptr, *d_runs = nullptr, *d_vals = nullptr;
    int32_t *d_rows = nullptr, *d_cols = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0, need = 0;
    int64_t *h_start = nullptr;
    hipStream_t s = nullptr;
    int64_t m = 0, runs = 0;
    int32_t last_valid = 0;
    int64_t last_pos = 0;
    int end_bit = 64;

    {
        hipError_t e0 = hipSetDevice(device);
        if (e0 != hipSuccess)
            return ek_set_error(EK_EHIP, "hipSetDevice(%d): %s", device,
                                hipGetErrorString(e0));
    }
    MSM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    h_start = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_trj + 1));
    if (!h_start) {
        rc = ek_set_error(EK_ENOMEM, "ek_msm_counts: out of host memory");
        goto done;
    }
    h_start[0] = 0;
    for (int64_t t = 0; t < n_trj; ++t)
        h_start[t + 1] = h_start[t] + lengths[t];

    MSM_HIP(hipMalloc((void **)&d_a, (size_t)n * sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_valid, (size_t)n * sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_pos, (size_t)n * sizeof(int64_t)));
    MSM_HIP(hipMalloc((void **)&d_c, (size_t)(n + lag_time) * sizeof(int32_t)));
    MSM_HIP(hipMalloc((void **)&d_start, (size_t)(n_trj + 1) * sizeof(int64_t)));
    MSM_HIP(hipMalloc((void **)&d_cstart, (size_t)(n_trj + 1) * sizeof(int64_t)));
    MSM_HIP(hipMemcpyAsync(d_a, assigns, (size_t)n * sizeof(int32_t),
                           hipMemcpyHostToDevice, s));
    MSM_HIP(hipMemcpyAsync(d_start, h_start,
                           (size_t)(n_trj + 1) * sizeof(int64_t),
                           hipMemcpyHostToDevice, s));

    // 1. drop the -1 frames (transition_matrices.py:156): flags -> scan -> scatter
    hipLaunchKernelGGL(msm_valid_kernel, dim3(msm_blocks(n)), dim3(EK_BLOCK), 0,
                       s, d_a, n, d_valid);
    MSM_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_valid, d_pos, n, s));
    tmp_bytes = need;
    MSM_HIP(hipMalloc(&d_tmp, tmp_bytes));
    MSM_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, need, d_valid, d_pos, n, s));
    MSM_HIP(hipMemcpyAsync(&last_valid, d_valid + (n - 1), sizeof(int32_t),
                           hipMemcpyDeviceToHost, s));
    MSM_HIP(hipMemcpyAsync(&last_pos, d_pos + (n - 1), sizeof(int64_t),
                           hipMemcpyDeviceToHost, s));
    MSM_HIP(hipStreamSynchronize(s));
    m = last_pos + last_valid;                       // frames that survive
    if (m <= lag_time)
        goto done;
    hipLaunchKernelGGL(msm_compact_kernel, dim3(msm_blocks(n)), dim3(EK_BLOCK),
                       0, s, d_a, d_pos, n, d_c);
    hipLaunchKernelGGL(msm_cstart_kernel, dim3(msm_blocks(n_trj + 1)),
                       dim3(EK_BLOCK), 0, s, d_start, d_pos, n_trj, n, m,
                       d_cstart);

    // 2. one key per (start, end) pair (:310-321), sentinel where none
    MSM_HIP(hipMalloc((void **)&d_keys, (size_t)m * sizeof(unsigned long long)));
    MSM_HIP(hipMalloc((void **)&d_keys2, (size_t)m * sizeof(unsigned long long)));
    hipLaunchKernelGGL(msm_keys_kernel, dim3(msm_blocks(m)), dim3(EK_BLOCK), 0,
                       s, d_c, d_cstart, n_trj, m, lag_time, sliding_window,
                       (int64_t)n_states, d_keys);

    // 3. sort + run-length encode = the duplicate-summing of the COO->CSR
    //    conversion (:167-169)
    need = 0;
    MSM_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, need, d_keys, d_keys2, m,
                                              0, end_bit, s));
    if (need > tmp_bytes) {
        MSM_HIP(hipFree(d_tmp));
        d_tmp = nullptr;
        tmp_bytes = need;
        MSM_HIP(hipMalloc(&d_tmp, tmp_bytes));
    }
    MSM_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp, need, d_keys, d_keys2, m, 0,
                                              end_bit, s));
    MSM_HIP(hipMalloc((void **)&d_ukeys, (size_t)m * sizeof(unsigned long long)));
    MSM_HIP(hipMalloc((void **)&d_ucnt, (size_t)m * sizeof(int64_t)));
    MSM_HIP(hipMalloc((void **)&d_runs, sizeof(int64_t)));
    need = 0;
    MSM_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, need, d_keys2, d_ukeys,
                                                  d_ucnt, d_runs, m, s));
    if (need > tmp_bytes) {
        MSM_HIP(hipFree(d_tmp));
        d_tmp = nullptr;
        tmp_bytes = need;
        MSM_HIP(hipMalloc(&d_tmp, tmp_bytes));
    }
    MSM_HIP(hipcub::DeviceRunLengthEncode::Encode(d_tmp, need, d_keys2, d_ukeys,
                                                  d_ucnt, d_runs, m, s));
    MSM_HIP(hipMemcpyAsync(&runs, d_runs, sizeof(int64_t), hipMemcpyDeviceToHost,
                           s));
    MSM_HIP(hipStreamSynchronize(s));
    if (runs > 0) {
        MSM_HIP(hipMalloc((void **)&d_rows, (size_t)runs * sizeof(int32_t)));
        MSM_HIP(hipMalloc((void **)&d_cols, (size_t)runs * sizeof(int32_t)));
        MSM_HIP(hipMalloc((void **)&d_vals, (size_t)runs * sizeof(int64_t)));
        hipLaunchKernelGGL(msm_split_kernel, dim3(msm_blocks(runs)),
                           dim3(EK_BLOCK), 0, s, d_ukeys, d_ucnt, runs,
                           (int64_t)n_states, d_rows, d_cols, d_vals);
        // the sentinel run, if any, sorts last
        unsigned long long lastkey = 0;
        MSM_HIP(hipMemcpyAsync(&lastkey, d_ukeys + (runs - 1), sizeof(lastkey),
                               hipMemcpyDeviceToHost, s));
        MSM_HIP(hipStreamSynchronize(s));
        const int64_t nnz = (lastkey == ~0ull) ? runs - 1 : runs;
        if (nnz > capacity) {
            rc = ek_set_error(EK_EARG, "ek_msm_counts: %lld entries exceed the "
                                       "capacity %lld", (long long)nnz,
                              (long long)capacity);
            goto done;
        }
        if (nnz > 0) {
            MSM_HIP(hipMemcpyAsync(rows_out, d_rows, (size_t)nnz * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, s));
            MSM_HIP(hipMemcpyAsync(cols_out, d_cols, (size_t)nnz * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, s));
            MSM_HIP(hipMemcpyAsync(counts_out, d_vals,
                                   (size_t)nnz * sizeof(int64_t),
                                   hipMemcpyDeviceToHost, s));
            MSM_HIP(hipStreamSynchronize(s));
        }
        *nnz_out = nnz;
    }

done:
    if (s)
        (void)hipStreamSynchronize(s);
    (void)hipFree(d_a);
    (void)hipFree(d_valid);
    (void)hipFree(d_pos);
    (void)hipFree(d_c);
    (void)hipFree(d_start);
    (void)hipFree(d_cstart);
    (void)hipFree(d_keys);
    (void)hipFree(d_keys2);
    (void)hipFree(d_ukeys);
    (void)hipFree(d_ucnt);
    (void)hipFree(d_runs);
    (void)hipFree(d_rows);
    (void)hipFree(d_cols);
    (void)hipFree(d_vals);
    (void)hipFree(d_tmp);
    free(h_start);
    if (s)
        (void)hipStreamDestroy(s);
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
