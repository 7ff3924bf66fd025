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
// Integer work, HBM/atomic-light: compaction by prefix sum, one 64-bit key
// (row * n_states + col) per transition, radix sort + run-length encode
// (rocPRIM through hipCUB) -> COO sorted by (row, col) = CSR order.
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
