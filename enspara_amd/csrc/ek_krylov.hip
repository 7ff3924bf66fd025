// ek_krylov.hip -- device primitives of an Arnoldi / Krylov-Schur eigensolver
// for the leading eigenpairs of a sparse transition matrix.
//
// Replaces the O(nnz) and O(n*m) work of
//   eigenspectrum   enspara/msm/transition_matrices.py:173-233
//   eq_probs        enspara/msm/transition_matrices.py:304-307
// (the reference calls ARPACK through scipy.sparse.linalg.eigs, or LAPACK for
// fewer than 1000 states).  The basis and the operator live in HBM; the host
// (enspara_amd/msm/transition_matrices.py) keeps only the small projected
// m x m problem.  All float64; every reduction has a fixed order, so results
// are reproducible run to run.
#include "ek_common.h"

#include <stdlib.h>

#include <algorithm>
#include <new>

extern int ek_set_error(int code, const char *fmt, ...);

struct ek_krylov {
    int device = 0;
    int64_t n = 0, nnz = 0;
    int32_t m_max = 0;
    hipStream_t s = nullptr;
    int64_t *indptr = nullptr;
    int32_t *indices = nullptr;
    double *data = nullptr;
    double *V = nullptr;      // [m_max + 1][n]
    double *w = nullptr;      // [n]
    double *w2 = nullptr;     // [n] the other half of the ping-pong in ek_krylov_expand
    double *part = nullptr;   // [blocks(n)] per-block sums of w^2 (its norm, in block order)
    double *tmp = nullptr;    // [m_max + 1][n] scratch for basis rotations
    double *h = nullptr;      // [m_max + 2] coefficients of one pass
    double *q = nullptr;      // [(m_max+1) * (m_max+1)] rotation matrix
    double *hcols = nullptr;  // [m_max][m_max + 2] Hessenberg columns (expand)
    // the operator of the steps: A itself (fdeg 0) or the Chebyshev polynomial of
    // degree fdeg of (A - fc) / fe (ek_krylov_set_filter)
    int32_t fdeg = 0;
    double fc = 0.0, fe = 1.0;
    double *t[2] = {nullptr, nullptr};  // [n] each: the recurrence's other vectors
};

#define KR_HIP(call)                                                           \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess)                                                  \
            return ek_set_error(EK_EHIP, "%s failed: %s at %s:%d", #call,      \
                                hipGetErrorString(e_), __FILE__, __LINE__);    \
    } while (0)

// w = A v, one wave per row, fixed-order shuffle reduction
__global__ void __launch_bounds__(EK_BLOCK)
kr_spmv_kernel(const int64_t *__restrict__ indptr,
               const int32_t *__restrict__ indices,
               const double *__restrict__ data, const double *__restrict__ v,
               int64_t n, double *__restrict__ w)
{
    const int64_t row = (int64_t)blockIdx.x * (EK_BLOCK / EK_WAVE) +
                        threadIdx.x / EK_WAVE;
    const int lane = threadIdx.x & (EK_WAVE - 1);
    if (row >= n)
        return;
    double acc = 0.0;
    for (int64_t j = indptr[row] + lane; j < indptr[row + 1]; j += EK_WAVE)
        acc = __builtin_fma(data[j], v[indices[j]], acc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        acc = acc + __shfl_xor(acc, off, 64);
    if (lane == 0)
        w[row] = acc;
}

// h[i] = V[i] . w  for i = blockIdx.x (one workgroup per basis vector)
__global__ void __launch_bounds__(EK_BLOCK)
kr_dots_kernel(const double *__restrict__ V, const double *__restrict__ w,
               int64_t n, double *__restrict__ h)
{
    __shared__ double sh[EK_BLOCK];
    const double *v = V + (size_t)blockIdx.x * n;
    double acc = 0.0;
    for (int64_t e = threadIdx.x; e < n; e += EK_BLOCK)
        acc = __builtin_fma(v[e], w[e], acc);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = EK_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        h[blockIdx.x] = sh[0];
}

// w -= sum_{i < cnt} h[i] V[i]
__global__ void __launch_bounds__(EK_BLOCK)
kr_axpy_kernel(const double *__restrict__ V, const double *__restrict__ h,
               int cnt, int64_t n, double *__restrict__ w)
{
    const int64_t e = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (e >= n)
        return;
    double x = w[e];
    for (int i = 0; i < cnt; ++i)
        x = __builtin_fma(-h[i], V[(size_t)i * n + e], x);
    w[e] = x;
}

__global__ void __launch_bounds__(EK_BLOCK)
kr_scale_kernel(const double *__restrict__ w, double alpha, int64_t n,
                double *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (e < n)
        out[e] = alpha * w[e];
}

// out[c][e] = sum_{r < m} V[r][e] * Q[r + c*m]   (Q column-major m x kk)
__global__ void __launch_bounds__(EK_BLOCK)
kr_rotate_kernel(const double *__restrict__ V, const double *__restrict__ Q,
                 int m, int kk, int64_t n, double *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    const int c = blockIdx.y;
    if (e >= n || c >= kk)
        return;
    double x = 0.0;
    for (int r = 0; r < m; ++r)
        x = __builtin_fma(V[(size_t)r * n + e], Q[(size_t)c * m + r], x);
    out[(size_t)c * n + e] = x;
}

// out = alpha (A x - c x) + beta z, x = x_raw / ||x_raw|| where `part` is given (the
// norm's partial sums of the step before, as in kr_spmv_norm_kernel: then V_out
// receives x and *col_last the norm).  z may be `out` itself (a row reads its own
// element before it writes it).  One wave per row.
__global__ void __launch_bounds__(EK_BLOCK)
kr_spmv_cheb_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                    const double *__restrict__ data, const double *x_raw,
                    const double *__restrict__ part, int nb, int64_t n, double c,
                    double alpha, const double *z, double beta, double *out,
                    double *V_out, double *col_last);

static double *kr_apply_filter(ek_krylov *k, const double *x_raw, double *Vj, double *col_last,
                               int nb, double *avoid);

static inline unsigned kr_blocks(int64_t n)
{
    return (unsigned)std::max<int64_t>(1, (n + EK_BLOCK - 1) / EK_BLOCK);
}

extern "C" int ek_krylov_destroy(ek_krylov *k)
{
    if (!k)
        return EK_OK;
    (void)hipSetDevice(k->device);
    if (k->s)
        (void)hipStreamSynchronize(k->s);
    (void)hipFree(k->indptr);
    (void)hipFree(k->indices);
    (void)hipFree(k->data);
    (void)hipFree(k->V);
    (void)hipFree(k->w);
    (void)hipFree(k->w2);
    (void)hipFree(k->part);
    (void)hipFree(k->tmp);
    (void)hipFree(k->h);
    (void)hipFree(k->q);
    (void)hipFree(k->hcols);
    (void)hipFree(k->t[0]);
    (void)hipFree(k->t[1]);
    if (k->s)
        (void)hipStreamDestroy(k->s);
    delete k;
    return EK_OK;
}

extern "C" int ek_krylov_create(int device, int64_t n, const int64_t *indptr,
                                const int32_t *indices, const double *data,
                                int32_t m_max, ek_krylov **out)
{
    if (!out || !indptr || n < 1 || m_max < 1)
        return ek_set_error(EK_EARG, "ek_krylov_create: bad argument");
    *out = nullptr;
    KR_HIP(hipSetDevice(device));
    ek_krylov *k = new (std::nothrow) ek_krylov();
    if (!k)
        return ek_set_error(EK_ENOMEM, "ek_krylov_create: out of memory");
    k->device = device;
    k->n = n;
    k->nnz = indptr[n];
    k->m_max = m_max;
    const size_t nz = (size_t)std::max<int64_t>(k->nnz, 1);
    hipError_t e = hipStreamCreateWithFlags(&k->s, hipStreamNonBlocking);
#define KA(ptr, bytes)                                                         \
    if (e == hipSuccess)                                                       \
        e = hipMalloc((void **)&(ptr), (bytes));
    KA(k->indptr, (size_t)(n + 1) * sizeof(int64_t));
    KA(k->indices, nz * sizeof(int32_t));
    KA(k->data, nz * sizeof(double));
    KA(k->V, (size_t)(m_max + 1) * n * sizeof(double));
    KA(k->tmp, (size_t)(m_max + 1) * n * sizeof(double));
    KA(k->w, (size_t)n * sizeof(double));
    KA(k->w2, (size_t)n * sizeof(double));
    KA(k->part, (size_t)((n + EK_BLOCK - 1) / EK_BLOCK) * sizeof(double));
    KA(k->h, (size_t)(m_max + 2) * sizeof(double));
    KA(k->q, (size_t)(m_max + 1) * (m_max + 1) * sizeof(double));
    KA(k->hcols, (size_t)m_max * (m_max + 2) * sizeof(double));
    KA(k->t[0], (size_t)n * sizeof(double));
    KA(k->t[1], (size_t)n * sizeof(double));
#undef KA
    if (e == hipSuccess)        // (ek_krylov_expand copies whole columns out)
        e = hipMemsetAsync(k->hcols, 0, (size_t)m_max * (m_max + 2) * sizeof(double), k->s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(k->indptr, indptr, (size_t)(n + 1) * sizeof(int64_t),
                           hipMemcpyHostToDevice, k->s);
    if (e == hipSuccess && k->nnz > 0)
        e = hipMemcpyAsync(k->indices, indices, (size_t)k->nnz * sizeof(int32_t),
                           hipMemcpyHostToDevice, k->s);
    if (e == hipSuccess && k->nnz > 0)
        e = hipMemcpyAsync(k->data, data, (size_t)k->nnz * sizeof(double),
                           hipMemcpyHostToDevice, k->s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(k->s);
    if (e != hipSuccess) {
        ek_krylov_destroy(k);
        return ek_set_error(e == hipErrorOutOfMemory ? EK_ENOMEM : EK_EHIP,
                            "ek_krylov_create: %s", hipGetErrorString(e));
    }
    *out = k;
    return EK_OK;
}

extern "C" int ek_krylov_set_vector(ek_krylov *k, int32_t j, const double *vec)
{
    if (!k || !vec || j < 0 || j > k->m_max)
        return ek_set_error(EK_EARG, "ek_krylov_set_vector: bad argument");
    KR_HIP(hipSetDevice(k->device));
    KR_HIP(hipMemcpyAsync(k->V + (size_t)j * k->n, vec,
                          (size_t)k->n * sizeof(double), hipMemcpyHostToDevice,
                          k->s));
    KR_HIP(hipStreamSynchronize(k->s));
    return EK_OK;
}

extern "C" int ek_krylov_get_vector(ek_krylov *k, int32_t j, double *vec)
{
    if (!k || !vec || j < 0 || j > k->m_max)
        return ek_set_error(EK_EARG, "ek_krylov_get_vector: bad argument");
    KR_HIP(hipSetDevice(k->device));
    KR_HIP(hipMemcpyAsync(vec, k->V + (size_t)j * k->n,
                          (size_t)k->n * sizeof(double), hipMemcpyDeviceToHost,
                          k->s));
    KR_HIP(hipStreamSynchronize(k->s));
    return EK_OK;
}

// One Arnoldi step.  apply != 0: w = A V[j]; apply == 0: w = V[j+1] as it
// stands (used to orthogonalise a fresh start vector after a breakdown).
// w is orthogonalised against V[0..j] by classical Gram-Schmidt done twice;
// h_col[0..j] receives the summed coefficients, h_col[j+1] = ||w||, and
// V[j+1] = w / ||w|| (left untouched when the norm is 0).
extern "C" int ek_krylov_step(ek_krylov *k, int32_t j, int32_t apply,
                              double *h_col)
{
    if (!k || !h_col || j < 0 || j >= k->m_max)
        return ek_set_error(EK_EARG, "ek_krylov_step: bad argument");
    KR_HIP(hipSetDevice(k->device));
    const int64_t n = k->n;
    const int cnt = j + 1;
    if (apply && k->fdeg >= 2) {
        double *r = kr_apply_filter(k, nullptr, k->V + (size_t)j * n, nullptr, 0, k->w);
        KR_HIP(hipMemcpyAsync(k->w, r, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice,
                              k->s));
    } else if (apply) {
        hipLaunchKernelGGL(kr_spmv_kernel, dim3((unsigned)((n + 3) / 4)),
                           dim3(EK_BLOCK), 0, k->s, k->indptr, k->indices,
                           k->data, k->V + (size_t)j * n, n, k->w);
    } else {
        KR_HIP(hipMemcpyAsync(k->w, k->V + (size_t)(j + 1) * n,
                              (size_t)n * sizeof(double),
                              hipMemcpyDeviceToDevice, k->s));
    }
    double *h1 = (double *)malloc(sizeof(double) * (size_t)(cnt + 1) * 2);
    if (!h1)
        return ek_set_error(EK_ENOMEM, "ek_krylov_step: out of memory");
    double *h2 = h1 + cnt + 1;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(kr_dots_kernel, dim3(cnt), dim3(EK_BLOCK), 0, k->s,
                           k->V, k->w, n, k->h);
        hipLaunchKernelGGL(kr_axpy_kernel, dim3(kr_blocks(n)), dim3(EK_BLOCK), 0,
                           k->s, k->V, k->h, cnt, n, k->w);
        hipError_t e = hipMemcpyAsync(pass ? h2 : h1, k->h,
                                      (size_t)cnt * sizeof(double),
                                      hipMemcpyDeviceToHost, k->s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(k->s);
        if (e != hipSuccess) {
            free(h1);
            return ek_set_error(EK_EHIP, "ek_krylov_step: %s",
                                hipGetErrorString(e));
        }
    }
    for (int i = 0; i < cnt; ++i)
        h_col[i] = h1[i] + h2[i];
    free(h1);
    // norm of what is left
    hipLaunchKernelGGL(kr_dots_kernel, dim3(1), dim3(EK_BLOCK), 0, k->s, k->w,
                       k->w, n, k->h);
    double nn = 0.0;
    KR_HIP(hipMemcpyAsync(&nn, k->h, sizeof(double), hipMemcpyDeviceToHost,
                          k->s));
    KR_HIP(hipStreamSynchronize(k->s));
    const double nrm = sqrt(nn);
    h_col[cnt] = nrm;
    if (nrm > 0.0) {
        hipLaunchKernelGGL(kr_scale_kernel, dim3(kr_blocks(n)), dim3(EK_BLOCK),
                           0, k->s, k->w, 1.0 / nrm, n,
                           k->V + (size_t)(j + 1) * n);
        KR_HIP(hipGetLastError());
    }
    return EK_OK;
}

// V[0..kk) <- V[0..m) * Q (Q column-major m x kk, host); if move_last != 0
// also V[kk] <- V[m] (the residual direction of a Krylov-Schur restart).
extern "C" int ek_krylov_rotate(ek_krylov *k, int32_t m, int32_t kk,
                                const double *Q, int32_t move_last)
{
    if (!k || !Q || m < 1 || m > k->m_max + 1 || kk < 1 || kk > m)
        return ek_set_error(EK_EARG, "ek_krylov_rotate: bad argument");
    KR_HIP(hipSetDevice(k->device));
    const int64_t n = k->n;
    KR_HIP(hipMemcpyAsync(k->q, Q, (size_t)m * kk * sizeof(double),
                          hipMemcpyHostToDevice, k->s));
    hipLaunchKernelGGL(kr_rotate_kernel, dim3(kr_blocks(n), kk), dim3(EK_BLOCK),
                       0, k->s, k->V, k->q, m, kk, n, k->tmp);
    KR_HIP(hipGetLastError());
    if (move_last)
        KR_HIP(hipMemcpyAsync(k->tmp + (size_t)kk * n, k->V + (size_t)m * n,
                              (size_t)n * sizeof(double),
                              hipMemcpyDeviceToDevice, k->s));
    KR_HIP(hipMemcpyAsync(k->V, k->tmp,
                          (size_t)(kk + (move_last ? 1 : 0)) * n * sizeof(double),
                          hipMemcpyDeviceToDevice, k->s));
    KR_HIP(hipStreamSynchronize(k->s));
    return EK_OK;
}

// out[c] = V[0..m) * Q[:, c] for c < kk, copied to host memory ([kk][n]);
// the basis is left untouched (final Ritz vectors).
extern "C" int ek_krylov_combine(ek_krylov *k, int32_t m, int32_t kk,
                                 const double *Q, double *out_host)
{
    if (!k || !Q || !out_host || m < 1 || m > k->m_max + 1 || kk < 1 ||
        kk > k->m_max + 1)
        return ek_set_error(EK_EARG, "ek_krylov_combine: bad argument");
    KR_HIP(hipSetDevice(k->device));
    const int64_t n = k->n;
    KR_HIP(hipMemcpyAsync(k->q, Q, (size_t)m * kk * sizeof(double),
                          hipMemcpyHostToDevice, k->s));
    hipLaunchKernelGGL(kr_rotate_kernel, dim3(kr_blocks(n), kk), dim3(EK_BLOCK),
                       0, k->s, k->V, k->q, m, kk, n, k->tmp);
    KR_HIP(hipGetLastError());
    KR_HIP(hipMemcpyAsync(out_host, k->tmp, (size_t)kk * n * sizeof(double),
                          hipMemcpyDeviceToHost, k->s));
    KR_HIP(hipStreamSynchronize(k->s));
    return EK_OK;
}

// ---- a whole run of Arnoldi steps without host round trips ---------------------
// ---- fused forms for ek_krylov_expand: five launches per step instead of nine ------
// ||w||^2 from the per-block partial sums, in a fixed order: every wave adds the
// partials lane-strided and reduces over the lanes (all lanes get the total)
__device__ __forceinline__ double kr_total(const double *__restrict__ part, int nb)
{
    const int lane = threadIdx.x & (EK_WAVE - 1);
    double t = 0.0;
    for (int i = lane; i < nb; i += EK_WAVE)
        t = t + part[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        t = t + __shfl_xor(t, off, 64);
    return t;
}

__global__ void __launch_bounds__(EK_BLOCK)
kr_spmv_cheb_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                    const double *__restrict__ data, const double *x_raw,
                    const double *__restrict__ part, int nb, int64_t n, double c,
                    double alpha, const double *z, double beta, double *out,
                    double *V_out, double *col_last)
{
    const int64_t row = (int64_t)blockIdx.x * (EK_BLOCK / EK_WAVE) +
                        threadIdx.x / EK_WAVE;
    const int lane = threadIdx.x & (EK_WAVE - 1);
    if (row >= n)
        return;
    const double nrm = part ? sqrt(kr_total(part, nb)) : 1.0;
    double acc = 0.0;
    for (int64_t j = indptr[row] + lane; j < indptr[row + 1]; j += EK_WAVE) {
        const double v = part ? ((nrm > 0.0) ? x_raw[indices[j]] / nrm : 0.0)
                              : x_raw[indices[j]];
        acc = __builtin_fma(data[j], v, acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        acc = acc + __shfl_xor(acc, off, 64);
    if (lane == 0) {
        const double xr = part ? ((nrm > 0.0) ? x_raw[row] / nrm : 0.0) : x_raw[row];
        if (V_out) {
            V_out[row] = xr;
            if (row == 0)
                *col_last = nrm;
        }
        double r = alpha * (acc - c * xr);
        if (z)
            r = r + beta * z[row];
        out[row] = r;
    }
}

// w = T_d((A - c) / e) x by the three-term recurrence, d launches: y_1 = (A - c) y_0 / e,
// y_{k+1} = 2 (A - c) y_k / e - y_{k-1}.  x is V[j] (x_raw = nullptr) or x_raw / its
// norm (then V[j] and the column's last entry are written on the way).  -> where the
// result is (one of k->w, k->w2, k->t[0], k->t[1], never x_raw)
static double *kr_apply_filter(ek_krylov *k, const double *x_raw, double *Vj, double *col_last,
                               int nb, double *avoid)
{
    const int64_t n = k->n;
    double *bufs[4] = {k->w, k->w2, k->t[0], k->t[1]};
    double *free2[2];
    int nf = 0;
    for (int i = 0; i < 4 && nf < 2; ++i)
        if (bufs[i] != x_raw && bufs[i] != avoid)
            free2[nf++] = bufs[i];
    double *cur = free2[0], *other = free2[1];
    const dim3 grid((unsigned)((n + 3) / 4));
    // y_1 -> cur
    hipLaunchKernelGGL(kr_spmv_cheb_kernel, grid, dim3(EK_BLOCK), 0, k->s, k->indptr,
                       k->indices, k->data, x_raw ? x_raw : (const double *)Vj,
                       x_raw ? k->part : (const double *)nullptr, nb, n, k->fc, 1.0 / k->fe,
                       (const double *)nullptr, 0.0, cur, x_raw ? Vj : (double *)nullptr,
                       col_last);
    for (int d = 2; d <= k->fdeg; ++d) {
        // y_d = 2 (A - c) y_{d-1} / e - y_{d-2}: y_{d-1} is in cur, y_{d-2} in Vj (d == 2)
        // or in other, which the result overwrites
        hipLaunchKernelGGL(kr_spmv_cheb_kernel, grid, dim3(EK_BLOCK), 0, k->s, k->indptr,
                           k->indices, k->data, (const double *)cur,
                           (const double *)nullptr, nb, n, k->fc, 2.0 / k->fe,
                           d == 2 ? (const double *)Vj : (const double *)other, -1.0, other,
                           (double *)nullptr, (double *)nullptr);
        std::swap(cur, other);
    }
    return cur;
}

extern "C" int ek_krylov_set_filter(ek_krylov *k, int32_t degree, double a, double b)
{
    if (!k || degree < 0 || degree == 1 || degree > 4096 || (degree > 0 && !(b > a)))
        return ek_set_error(EK_EARG, "ek_krylov_set_filter: degree 0 (none) or 2 .. 4096 "
                                     "on an interval a < b");
    k->fdeg = degree;
    k->fc = degree ? 0.5 * (a + b) : 0.0;
    k->fe = degree ? 0.5 * (b - a) : 1.0;
    return EK_OK;
}

// w_out = A (w_prev / ||w_prev||), and on the way V_out = w_prev / ||w_prev||,
// *col_last = ||w_prev||: the normalisation of the step before folded into the
// sparse product of this one (the product reads w_prev / nrm, the very values
// V_out receives).  One wave per row.
__global__ void __launch_bounds__(EK_BLOCK)
kr_spmv_norm_kernel(const int64_t *__restrict__ indptr,
                    const int32_t *__restrict__ indices,
                    const double *__restrict__ data, const double *__restrict__ w_prev,
                    const double *__restrict__ part, int nb, int64_t n,
                    double *__restrict__ w_out, double *__restrict__ V_out,
                    double *__restrict__ col_last)
{
    const int64_t row = (int64_t)blockIdx.x * (EK_BLOCK / EK_WAVE) +
                        threadIdx.x / EK_WAVE;
    const int lane = threadIdx.x & (EK_WAVE - 1);
    if (row >= n)
        return;
    const double nrm = sqrt(kr_total(part, nb));
    if (lane == 0) {
        V_out[row] = (nrm > 0.0) ? w_prev[row] / nrm : 0.0;
        if (row == 0)
            *col_last = nrm;
    }
    double acc = 0.0;
    for (int64_t j = indptr[row] + lane; j < indptr[row + 1]; j += EK_WAVE) {
        const double v = (nrm > 0.0) ? w_prev[indices[j]] / nrm : 0.0;
        acc = __builtin_fma(data[j], v, acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        acc = acc + __shfl_xor(acc, off, 64);
    if (lane == 0)
        w_out[row] = acc;
}

// w -= sum_{i < cnt} h[i] V[i]; the column of coefficients (= on the first pass,
// += on the second) by workgroup 0; and, where asked, the workgroup's share of
// ||w||^2 (a fixed tree over its 256 elements)
__global__ void __launch_bounds__(EK_BLOCK)
kr_axpy_hsum_kernel(const double *__restrict__ V, const double *__restrict__ h,
                    int cnt, int64_t n, double *__restrict__ w,
                    double *__restrict__ col, int first, double *__restrict__ part)
{
    __shared__ double sh[EK_BLOCK];
    const int64_t e = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    double x = 0.0;
    if (e < n) {
        x = w[e];
        for (int i = 0; i < cnt; ++i)
            x = __builtin_fma(-h[i], V[(size_t)i * n + e], x);
        w[e] = x;
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < cnt; i += EK_BLOCK)
            col[i] = first ? h[i] : col[i] + h[i];
    if (part) {
        sh[threadIdx.x] = x * x;
        __syncthreads();
        for (int s = EK_BLOCK / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s)
                sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0)
            part[blockIdx.x] = sh[0];
    }
}

// the last step's normalisation: V_out = w / ||w||, *col_last = ||w||
__global__ void __launch_bounds__(EK_BLOCK)
kr_finish_kernel(const double *__restrict__ w, const double *__restrict__ part, int nb,
                 int64_t n, double *__restrict__ V_out, double *__restrict__ col_last)
{
    const int64_t e = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    const double nrm = sqrt(kr_total(part, nb));
    if (e == 0)
        *col_last = nrm;
    if (e < n)
        V_out[e] = (nrm > 0.0) ? w[e] / nrm : 0.0;
}

// Steps j0 .. m-1 back to back.  H_out is column-major with leading dimension
// ldh >= m + 1: column j receives h[0..j+1].  A (near-)zero sub-diagonal entry
// signals a breakdown at that step; the caller then redoes the tail with
// ek_krylov_step, which can insert a fresh direction.
// Five launches per step (round 4; nine before): the sparse product with the
// normalisation of the step before folded in, and twice [coefficients, update +
// their column sum (+ the norm's partial sums on the second pass)].
extern "C" int ek_krylov_expand(ek_krylov *k, int32_t j0, int32_t m,
                                double *H_out, int32_t ldh)
{
    if (!k || !H_out || j0 < 0 || m > k->m_max || j0 > m || ldh < m + 1)
        return ek_set_error(EK_EARG, "ek_krylov_expand: bad argument");
    KR_HIP(hipSetDevice(k->device));
    const int64_t n = k->n;
    const int ld = k->m_max + 2;
    const int nb = (int)kr_blocks(n);
    double *w = k->w, *w_prev = k->w2;
    for (int j = j0; j < m; ++j) {
        const int cnt = j + 1;
        double *col = k->hcols + (size_t)j * ld;
        if (k->fdeg >= 2) {
            // (the result lands in one of four vectors; the orthogonalisation below
            // works on it in place, the next step reads it as w_prev)
            w = (j == j0)
                    ? kr_apply_filter(k, nullptr, k->V + (size_t)j * n, nullptr, nb, nullptr)
                    : kr_apply_filter(k, w_prev, k->V + (size_t)j * n,
                                      k->hcols + (size_t)(j - 1) * ld + j, nb, nullptr);
        } else if (j == j0)
            hipLaunchKernelGGL(kr_spmv_kernel, dim3((unsigned)((n + 3) / 4)),
                               dim3(EK_BLOCK), 0, k->s, k->indptr, k->indices,
                               k->data, k->V + (size_t)j * n, n, w);
        else        // (w_prev: what step j - 1 left; its column's last entry)
            hipLaunchKernelGGL(kr_spmv_norm_kernel, dim3((unsigned)((n + 3) / 4)),
                               dim3(EK_BLOCK), 0, k->s, k->indptr, k->indices,
                               k->data, w_prev, k->part, nb, n, w,
                               k->V + (size_t)j * n,
                               k->hcols + (size_t)(j - 1) * ld + j);
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(kr_dots_kernel, dim3(cnt), dim3(EK_BLOCK), 0,
                               k->s, k->V, w, n, k->h);
            hipLaunchKernelGGL(kr_axpy_hsum_kernel, dim3(kr_blocks(n)),
                               dim3(EK_BLOCK), 0, k->s, k->V, k->h, cnt, n, w, col,
                               pass == 0, pass == 1 ? k->part : (double *)nullptr);
        }
        std::swap(w, w_prev);
    }
    if (m > j0)     // the last step's vector and sub-diagonal entry
        hipLaunchKernelGGL(kr_finish_kernel, dim3(kr_blocks(n)), dim3(EK_BLOCK), 0,
                           k->s, w_prev, k->part, nb, n, k->V + (size_t)m * n,
                           k->hcols + (size_t)(m - 1) * ld + m);
    KR_HIP(hipGetLastError());
    // all columns j0 .. m - 1 in ONE copy (forty small ones cost 0.2-0.3 ms per
    // restart): m + 1 entries each -- column j has j + 2, the rest of a column is
    // zero since the allocation and never written.  (The ~200 launches of a
    // restart replayed from a captured hipGraph took the same 0.73 ms: the time
    // is the dependent kernels' own turn-around, not the host's launches.)
    if (m > j0)
        KR_HIP(hipMemcpy2DAsync(H_out + (size_t)j0 * ldh, (size_t)ldh * sizeof(double),
                                k->hcols + (size_t)j0 * ld, (size_t)ld * sizeof(double),
                                (size_t)(m + 1) * sizeof(double), (size_t)(m - j0),
                                hipMemcpyDeviceToHost, k->s));
    KR_HIP(hipStreamSynchronize(k->s));
    return EK_OK;
}
