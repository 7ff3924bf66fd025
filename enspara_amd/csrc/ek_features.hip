// ek_features.hip -- point-vs-set distances in feature space.
//
// Replaces the reference's only native distance code, the Cython/OpenMP
// kernels of enspara/geometry/libdist.pyx (bound as metrics 'euclidean' and
// 'manhattan'/'cityblock' at enspara/cluster/util.py:292-295):
//   _euclidean :122-145   out[i] = sqrt( sum_j (X[i,j] - y[j])**2 )
//   _manhattan :100-117   out[i] = sum_j fabs(X[i,j] - y[j])
//   _hamming   :77-95     out[i] = (number of j with X[i,j] != y[j]) / n_features
// Output is float64 in all three.  Arithmetic contract (what the generated C
// of the reference does, so results are bit-identical):
//   float32 input: the difference and its square are float32 operations, the
//   running sum is float64, terms added in feature order; manhattan widens the
//   float32 difference to float64 before fabs;
//   float64 input: everything in float64;
//   hamming: exact integer comparison.
//
// Same mapping as the RMSD kernels: samples are stored feature-major in tiles
// of 256 ("frame-minor"), one lane owns one sample and walks the features in
// order (no cross-lane reduction, order fixed), the target point is staged in
// LDS in chunks and read as wave-wide broadcasts.  HBM-bound: 4 or 8 bytes per
// (sample, feature) and ~2 flops.
#include "ek_common.h"
#include "ek_reduce.h"

#include <stdlib.h>

#include <string.h>
#include <algorithm>
#include <vector>
#include <new>

extern int ek_set_error(int code, const char *fmt, ...);

#define FT_CHUNK 32           // features per staged transposition chunk
#define FY_CHUNK 2048         // target-point features staged in LDS at a time

struct FeatPam;           // working set of ek_feat_pam_sweep (below)

struct ek_feat {
    int device = 0;
    int64_t n = 0;
    int32_t F = 0;
    int32_t kind = 0;         // 0 float32, 1 float64, 2 int64
    int32_t esize = 4;
    int64_t n_tiles = 0;
    hipStream_t s = nullptr;
    void *tiles = nullptr;    // [n_tiles][F][EK_TILE] elements
    void *stage = nullptr;
    int64_t stage_rows = 0;
    void *y = nullptr;        // [F] elements
    double *out = nullptr;    // [n]
    bool loaded = false;
    // device-resident k-centers state (ek_feat_kcenters)
    double *kdist = nullptr;  // [n] float64, as the reference keeps it
    int32_t *kassign = nullptr;
    struct FeatBlockMax *bm = nullptr;
    struct FeatCtl *ctl = nullptr;
    int64_t *hist = nullptr;
    int32_t hist_cap = 0;
    FeatPam *pam = nullptr;
};

// per-workgroup partial of the arg-max over float64 distances
struct FeatBlockMax {
    double val;
    int64_t idx;
};
struct FeatCtl {
    int64_t next;       // sample that becomes the next center
    int32_t n_done;     // centers applied by this run
    int32_t stopped;    // distances.max() <= cutoff (kcenters.py:217)
    double last_max;
};

#define FE_HIP(call)                                                           \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess)                                                  \
            return ek_set_error(EK_EHIP, "%s failed: %s at %s:%d", #call,      \
                                hipGetErrorString(e_), __FILE__, __LINE__);    \
    } while (0)

// ---- row-major [count][F] -> tiles, through LDS -------------------------------
template <typename T>
__global__ void __launch_bounds__(EK_BLOCK)
feat_transpose_kernel(const T *__restrict__ src, int64_t count, int F,
                      T *__restrict__ tiles, int64_t first)
{
    __shared__ T stage[EK_BLOCK * (FT_CHUNK + 1)];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * EK_BLOCK;
    const int64_t rows = (count - r0 < EK_BLOCK) ? (count - r0) : EK_BLOCK;
    const int64_t g = first + r0 + t;
    T *obase = tiles + (size_t)(g / EK_TILE) * (size_t)F * EK_TILE + (g % EK_TILE);
    for (int j0 = 0; j0 < F; j0 += FT_CHUNK) {
        const int w = (F - j0 < FT_CHUNK) ? (F - j0) : FT_CHUNK;
        const int64_t total = rows * w;
        for (int64_t i = t; i < total; i += EK_BLOCK) {
            const int r = (int)(i / w), j = (int)(i % w);
            stage[r * (FT_CHUNK + 1) + j] = src[(size_t)(r0 + r) * F + j0 + j];
        }
        __syncthreads();
        if (t < rows)
            for (int j = 0; j < w; ++j)
                obase[(size_t)(j0 + j) * EK_TILE] = stage[t * (FT_CHUNK + 1) + j];
        __syncthreads();
    }
}

// ---- distances -----------------------------------------------------------------
template <typename T, int METRIC> struct FeatAcc;
// euclidean
template <> struct FeatAcc<float, 0> {
    static __device__ __forceinline__ void add(double &acc, float x, float y)
    {
        const float d = x - y;           // float32 subtraction
        const float q = d * d;           // float32 product (powf(d, 2) == d*d)
        acc = acc + (double)q;
    }
};
template <> struct FeatAcc<double, 0> {
    static __device__ __forceinline__ void add(double &acc, double x, double y)
    {
        const double d = x - y;
        acc = acc + d * d;
    }
};
// manhattan
template <> struct FeatAcc<float, 1> {
    static __device__ __forceinline__ void add(double &acc, float x, float y)
    {
        const float d = x - y;
        acc = acc + __builtin_fabs((double)d);
    }
};
template <> struct FeatAcc<double, 1> {
    static __device__ __forceinline__ void add(double &acc, double x, double y)
    {
        acc = acc + __builtin_fabs(x - y);
    }
};
// hamming
template <> struct FeatAcc<long long, 2> {
    static __device__ __forceinline__ void add(double &acc, long long x,
                                               long long y)
    {
        if (x != y)
            acc = acc + 1.0;
    }
};

// what libdist.pyx does with a row's sum: sqrt (:143), nothing (:119), / n_features (:93)
template <int METRIC> __device__ __forceinline__ double feat_finish(double acc, int F)
{
    if (METRIC == 0)
        return __builtin_sqrt(acc);
    if (METRIC == 2)
        return acc / (double)F;
    return acc;
}

template <typename T, int METRIC>
__global__ void __launch_bounds__(EK_BLOCK)
feat_distance_kernel(const T *__restrict__ tiles, const T *__restrict__ y,
                     int64_t n, int F, double *__restrict__ out)
{
    __shared__ T ys[FY_CHUNK];
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    const T *p = tiles + (size_t)(f / EK_TILE) * (size_t)F * EK_TILE + (f % EK_TILE);
    double acc = 0.0;
    for (int j0 = 0; j0 < F; j0 += FY_CHUNK) {
        const int w = (F - j0 < FY_CHUNK) ? (F - j0) : FY_CHUNK;
        __syncthreads();
        for (int j = threadIdx.x; j < w; j += EK_BLOCK)
            ys[j] = y[j0 + j];
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < w; ++j)
            FeatAcc<T, METRIC>::add(acc, __builtin_nontemporal_load(
                                             p + (size_t)(j0 + j) * EK_TILE),
                                    ys[j]);
    }
    if (f < n) {
        if (METRIC == 0)
            acc = __builtin_sqrt(acc);
        else if (METRIC == 2)
            acc = acc / (double)F;
        out[f] = acc;
    }
}

// ---- C ABI ----------------------------------------------------------------------
extern "C" void ek_feat_pam_release(ek_feat *k);

extern "C" int ek_feat_destroy(ek_feat *k)
{
    if (!k)
        return EK_OK;
    ek_feat_pam_release(k);
    (void)hipSetDevice(k->device);
    if (k->s)
        (void)hipStreamSynchronize(k->s);
    (void)hipFree(k->tiles);
    (void)hipFree(k->stage);
    (void)hipFree(k->y);
    (void)hipFree(k->out);
    (void)hipFree(k->kdist);
    (void)hipFree(k->kassign);
    (void)hipFree(k->bm);
    (void)hipFree(k->ctl);
    (void)hipFree(k->hist);
    if (k->s)
        (void)hipStreamDestroy(k->s);
    delete k;
    return EK_OK;
}

extern "C" int ek_feat_create(int device, int64_t n_samples, int32_t n_features,
                              int32_t elem_kind, ek_feat **out)
{
    if (!out || n_samples < 0 || n_features < 1 || elem_kind < 0 ||
        elem_kind > 2)
        return ek_set_error(EK_EARG, "ek_feat_create: bad argument");
    *out = nullptr;
    FE_HIP(hipSetDevice(device));
    ek_feat *k = new (std::nothrow) ek_feat();
    if (!k)
        return ek_set_error(EK_ENOMEM, "ek_feat_create: out of memory");
    k->device = device;
    k->n = n_samples;
    k->F = n_features;
    k->kind = elem_kind;
    k->esize = elem_kind == 0 ? 4 : 8;
    k->n_tiles = (n_samples + EK_TILE - 1) / EK_TILE;
    const size_t tb = (size_t)std::max<int64_t>(k->n_tiles, 1) * n_features *
                      EK_TILE * k->esize;
    hipError_t e = hipStreamCreateWithFlags(&k->s, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipMalloc(&k->tiles, tb);
    if (e == hipSuccess)
        e = hipMemsetAsync(k->tiles, 0, tb, k->s);
    if (e == hipSuccess)
        e = hipMalloc(&k->y, (size_t)n_features * k->esize);
    if (e == hipSuccess)
        e = hipMalloc((void **)&k->out,
                      (size_t)std::max<int64_t>(n_samples, 1) * sizeof(double));
    if (e != hipSuccess) {
        ek_feat_destroy(k);
        return ek_set_error(e == hipErrorOutOfMemory ? EK_ENOMEM : EK_EHIP,
                            "ek_feat_create: %s", hipGetErrorString(e));
    }
    *out = k;
    return EK_OK;
}

extern "C" int ek_feat_load(ek_feat *k, const void *X, int64_t first,
                            int64_t count)
{
    if (!k || (!X && count > 0) || first < 0 || count < 0 ||
        first + count > k->n || first % EK_TILE)
        return ek_set_error(EK_EARG, "ek_feat_load: bad argument");
    FE_HIP(hipSetDevice(k->device));
    const size_t row = (size_t)k->F * k->esize;
    int64_t chunk = (int64_t)((128u << 20) / row);
    chunk = std::max<int64_t>(EK_TILE, chunk / EK_TILE * EK_TILE);
    chunk = std::min<int64_t>(chunk, (count + EK_TILE - 1) / EK_TILE * EK_TILE);
    if (chunk > k->stage_rows) {
        FE_HIP(hipStreamSynchronize(k->s));
        (void)hipFree(k->stage);
        k->stage = nullptr;
        k->stage_rows = 0;
        FE_HIP(hipMalloc(&k->stage, (size_t)chunk * row));
        k->stage_rows = chunk;
    }
    for (int64_t done = 0; done < count; done += chunk) {
        const int64_t cnt = std::min(chunk, count - done);
        FE_HIP(hipMemcpyAsync(k->stage, (const char *)X + (size_t)done * row,
                              (size_t)cnt * row, hipMemcpyHostToDevice, k->s));
        const unsigned blocks = (unsigned)((cnt + EK_BLOCK - 1) / EK_BLOCK);
        if (k->kind == 0)
            hipLaunchKernelGGL(feat_transpose_kernel<float>, dim3(blocks),
                               dim3(EK_BLOCK), 0, k->s, (const float *)k->stage,
                               cnt, k->F, (float *)k->tiles, first + done);
        else
            hipLaunchKernelGGL(feat_transpose_kernel<double>, dim3(blocks),
                               dim3(EK_BLOCK), 0, k->s, (const double *)k->stage,
                               cnt, k->F, (double *)k->tiles, first + done);
        FE_HIP(hipGetLastError());
        FE_HIP(hipStreamSynchronize(k->s));
    }
    k->loaded = true;
    return EK_OK;
}

extern "C" int ek_feat_distance(ek_feat *k, int32_t metric, const void *y,
                                double *out_host)
{
    if (!k || !y || !out_host || metric < 0 || metric > 2)
        return ek_set_error(EK_EARG, "ek_feat_distance: bad argument");
    if (!k->loaded)
        return ek_set_error(EK_ESTATE, "ek_feat_distance: no samples loaded");
    if ((metric == 2) != (k->kind == 2))
        return ek_set_error(EK_EARG, "ek_feat_distance: hamming needs integer "
                                     "samples, the other metrics floating point");
    if (k->n == 0)
        return EK_OK;
    FE_HIP(hipSetDevice(k->device));
    FE_HIP(hipMemcpyAsync(k->y, y, (size_t)k->F * k->esize,
                          hipMemcpyHostToDevice, k->s));
    const unsigned blocks = (unsigned)((k->n + EK_BLOCK - 1) / EK_BLOCK);
#define FE_GO(T, M)                                                            \
    hipLaunchKernelGGL((feat_distance_kernel<T, M>), dim3(blocks),             \
                       dim3(EK_BLOCK), 0, k->s, (const T *)k->tiles,           \
                       (const T *)k->y, k->n, k->F, k->out)
    if (metric == 2)
        FE_GO(long long, 2);
    else if (k->kind == 0) {
        if (metric == 0)
            FE_GO(float, 0);
        else
            FE_GO(float, 1);
    } else {
        if (metric == 0)
            FE_GO(double, 0);
        else
            FE_GO(double, 1);
    }
#undef FE_GO
    FE_HIP(hipGetLastError());
    FE_HIP(hipMemcpyAsync(out_host, k->out, (size_t)k->n * sizeof(double),
                          hipMemcpyDeviceToHost, k->s));
    FE_HIP(hipStreamSynchronize(k->s));
    return EK_OK;
}

// ===========================================================================
// k-centers in feature space, resident on the device
// ===========================================================================
// Reference: the loop of enspara/cluster/kcenters.py:217-231 with the serial
// iteration :243-311 for metrics 'euclidean' / 'manhattan' (libdist.pyx) --
//   new_index = argmax(distances); dist = metric(X, X[new_index]);
//   closer = dist < distances; distances[closer] = dist[closer]; assignments[closer] = k;
//   maxdist = distances.max()
// -- which costs a metric call plus six numpy passes over n and an arg-max on the
// host per center when only the metric runs on the device.  Here the float64
// distances and the labels stay in HBM: one launch computes the new center's
// distances (the arithmetic of feat_distance_kernel, bit for bit), applies the
// strict-< update and leaves per-workgroup (max, first index) partials; a
// single-workgroup launch reduces them, applies the stop rule, and copies the
// next center's features out of the tiles.  No host round trip per center.
__device__ __forceinline__ bool feat_better(double v, int64_t i, double bv, int64_t bi)
{
    return (v > bv) || (v == bv && i < bi);
}

__device__ __forceinline__ void feat_wave_argmax(double &v, int64_t &i)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const int64_t oi = __shfl_xor(i, off, 64);
        if (feat_better(ov, oi, v, i)) {
            v = ov;
            i = oi;
        }
    }
}

// block partial of (value, index) pairs held one per thread -> bm[blockIdx.x]
__device__ __forceinline__ void feat_block_partial(double v, int64_t i,
                                                   FeatBlockMax *bm)
{
    __shared__ double rv[EK_BLOCK / EK_WAVE];
    __shared__ int64_t ri[EK_BLOCK / EK_WAVE];
    feat_wave_argmax(v, i);
    if ((threadIdx.x & (EK_WAVE - 1)) == 0) {
        rv[threadIdx.x / EK_WAVE] = v;
        ri[threadIdx.x / EK_WAVE] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
            if (feat_better(rv[w], ri[w], v, i)) {
                v = rv[w];
                i = ri[w];
            }
        bm[blockIdx.x].val = v;
        bm[blockIdx.x].idx = i;
    }
}

template <typename T, int METRIC>
__global__ void __launch_bounds__(EK_BLOCK)
feat_step_kernel(const T *__restrict__ tiles, const T *__restrict__ y, int64_t n,
                 int F, int32_t label, double *__restrict__ dist,
                 int32_t *__restrict__ assign, FeatBlockMax *__restrict__ bm,
                 FeatCtl *__restrict__ ctl, int64_t *__restrict__ hist)
{
    __shared__ T ys[FY_CHUNK];
    if (ctl->stopped)
        return;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    const T *p = tiles + (size_t)(f / EK_TILE) * (size_t)F * EK_TILE + (f % EK_TILE);
    double acc = 0.0;
    for (int j0 = 0; j0 < F; j0 += FY_CHUNK) {
        const int w = (F - j0 < FY_CHUNK) ? (F - j0) : FY_CHUNK;
        __syncthreads();
        for (int j = threadIdx.x; j < w; j += EK_BLOCK)
            ys[j] = y[j0 + j];
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < w; ++j)
            FeatAcc<T, METRIC>::add(acc, __builtin_nontemporal_load(
                                             p + (size_t)(j0 + j) * EK_TILE),
                                    ys[j]);
    }
    double v = -__builtin_inf();
    int64_t i = 0x7fffffffffffffffLL;
    if (f < n) {
        if (METRIC == 0)
            acc = __builtin_sqrt(acc);
        else if (METRIC == 2)
            acc = acc / (double)F;
        double cur = dist[f];
        if (acc < cur) {                    // kcenters.py:304: strict <
            cur = acc;
            dist[f] = acc;
            assign[f] = label;
        }
        v = cur;
        i = f;
    }
    feat_block_partial(v, i, bm);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hist[label] = ctl->next;            // the sample this center is
        ctl->n_done = ctl->n_done + 1;
    }
}

// per-workgroup partials of the state as it stands (before the first step)
__global__ void __launch_bounds__(EK_BLOCK)
feat_blockmax_kernel(const double *__restrict__ dist, int64_t n,
                     FeatBlockMax *__restrict__ bm)
{
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    double v = -__builtin_inf();
    int64_t i = 0x7fffffffffffffffLL;
    if (f < n) {
        v = dist[f];
        i = f;
    }
    feat_block_partial(v, i, bm);
}

// np.argmax / distances.max() (kcenters.py:282, :226), the stop rule (:217) and
// the next center's features, contiguous in y
template <typename T>
__global__ void __launch_bounds__(1024)
feat_pick_kernel(const FeatBlockMax *__restrict__ bm, int nb,
                 const T *__restrict__ tiles, int F, double cutoff,
                 T *__restrict__ y, FeatCtl *__restrict__ ctl)
{
    __shared__ double rv[1024 / EK_WAVE];
    __shared__ int64_t ri[1024 / EK_WAVE];
    __shared__ int64_t win;
    if (ctl->stopped)
        return;
    const int tid = threadIdx.x;
    double v = -__builtin_inf();
    int64_t i = 0x7fffffffffffffffLL;
    for (int b = tid; b < nb; b += 1024) {
        const FeatBlockMax m = bm[b];
        if (feat_better(m.val, m.idx, v, i)) {
            v = m.val;
            i = m.idx;
        }
    }
    feat_wave_argmax(v, i);
    if ((tid & (EK_WAVE - 1)) == 0) {
        rv[tid / EK_WAVE] = v;
        ri[tid / EK_WAVE] = i;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 1024 / EK_WAVE; ++w)
            if (feat_better(rv[w], ri[w], v, i)) {
                v = rv[w];
                i = ri[w];
            }
        ctl->last_max = v;
        if (!(v > cutoff))
            ctl->stopped = 1;
        ctl->next = i;
        win = (v > cutoff) ? i : -1;
    }
    __syncthreads();
    const int64_t c = win;
    if (c < 0)
        return;
    const T *p = tiles + (size_t)(c / EK_TILE) * (size_t)F * EK_TILE + (c % EK_TILE);
    for (int j = tid; j < F; j += 1024)
        y[j] = p[(size_t)j * EK_TILE];
}

// Runs up to max_new iterations from the state (dist_io, assign_io) the caller
// passes in (float64 distances, int32 labels; a fresh run passes +inf / -1) with
// labels first_label, first_label + 1, ..; writes the state back, the samples
// chosen as centers to centers_out[0..*n_added) and distances.max() after the
// last update to *final_max.
extern "C" int ek_feat_kcenters(ek_feat *k, int32_t metric, int32_t first_label,
                                int32_t max_new, double dist_cutoff,
                                double *dist_io, int32_t *assign_io,
                                int64_t *centers_out, int32_t *n_added,
                                double *final_max)
{
    if (!k || !dist_io || !assign_io || !n_added || metric < 0 || metric > 2 ||
        first_label < 0 || max_new < 0)
        return ek_set_error(EK_EARG, "ek_feat_kcenters: bad argument");
    if (!k->loaded)
        return ek_set_error(EK_ESTATE, "ek_feat_kcenters: no samples loaded");
    if ((metric == 2) != (k->kind == 2))
        return ek_set_error(EK_EARG, "ek_feat_kcenters: hamming needs integer "
                                     "samples, the other metrics floating point");
    *n_added = 0;
    if (k->n == 0)
        return EK_OK;
    FE_HIP(hipSetDevice(k->device));
    const int nb = (int)((k->n + EK_BLOCK - 1) / EK_BLOCK);
    if (!k->kdist) {
        FE_HIP(hipMalloc((void **)&k->kdist, (size_t)k->n * sizeof(double)));
        FE_HIP(hipMalloc((void **)&k->kassign, (size_t)k->n * sizeof(int32_t)));
        FE_HIP(hipMalloc((void **)&k->bm, (size_t)nb * sizeof(FeatBlockMax)));
        FE_HIP(hipMalloc((void **)&k->ctl, sizeof(FeatCtl)));
    }
    if (first_label + max_new + 1 > k->hist_cap) {
        FE_HIP(hipStreamSynchronize(k->s));
        (void)hipFree(k->hist);
        k->hist = nullptr;
        k->hist_cap = 0;
        FE_HIP(hipMalloc((void **)&k->hist,
                         (size_t)(first_label + max_new + 1) * sizeof(int64_t)));
        k->hist_cap = first_label + max_new + 1;
    }
    FeatCtl c0;
    c0.next = 0;
    c0.n_done = 0;
    c0.stopped = 0;
    c0.last_max = 0.0;
    FE_HIP(hipMemcpyAsync(k->ctl, &c0, sizeof(c0), hipMemcpyHostToDevice, k->s));
    FE_HIP(hipMemcpyAsync(k->kdist, dist_io, (size_t)k->n * sizeof(double),
                          hipMemcpyHostToDevice, k->s));
    FE_HIP(hipMemcpyAsync(k->kassign, assign_io, (size_t)k->n * sizeof(int32_t),
                          hipMemcpyHostToDevice, k->s));
    const unsigned blocks = (unsigned)nb;
#define FK_PICK(T)                                                             \
    hipLaunchKernelGGL((feat_pick_kernel<T>), dim3(1), dim3(1024), 0, k->s,    \
                       k->bm, nb, (const T *)k->tiles, k->F, dist_cutoff,      \
                       (T *)k->y, k->ctl)
#define FK_STEP(T, M, LABEL)                                                   \
    hipLaunchKernelGGL((feat_step_kernel<T, M>), dim3(blocks), dim3(EK_BLOCK), \
                       0, k->s, (const T *)k->tiles, (const T *)k->y, k->n,    \
                       k->F, (LABEL), k->kdist, k->kassign, k->bm, k->ctl,     \
                       k->hist)
    hipLaunchKernelGGL(feat_blockmax_kernel, dim3(blocks), dim3(EK_BLOCK), 0, k->s,
                       k->kdist, k->n, k->bm);
    if (metric == 2)
        FK_PICK(long long);
    else if (k->kind == 0)
        FK_PICK(float);
    else
        FK_PICK(double);
    // with no cut-off the trip count is known: everything is enqueued at once;
    // with one, in batches, looking at the stop flag in between (steps enqueued
    // past the stopping point return at once)
    const bool open_loop = !(dist_cutoff > 0.0);
    const int32_t batch = open_loop ? max_new : 32;
    int32_t issued = 0;
    FeatCtl cr = c0;
    while (issued < max_new) {
        const int32_t todo = std::min(batch, max_new - issued);
        for (int32_t t = 0; t < todo; ++t) {
            const int32_t label = first_label + issued + t;
            if (metric == 2) {
                FK_STEP(long long, 2, label);
                FK_PICK(long long);
            } else if (k->kind == 0) {
                if (metric == 0)
                    FK_STEP(float, 0, label);
                else
                    FK_STEP(float, 1, label);
                FK_PICK(float);
            } else {
                if (metric == 0)
                    FK_STEP(double, 0, label);
                else
                    FK_STEP(double, 1, label);
                FK_PICK(double);
            }
        }
        FE_HIP(hipGetLastError());
        issued += todo;
        if (!open_loop) {
            FE_HIP(hipMemcpyAsync(&cr, k->ctl, sizeof(cr), hipMemcpyDeviceToHost,
                                  k->s));
            FE_HIP(hipStreamSynchronize(k->s));
            if (cr.stopped)
                break;
        }
    }
#undef FK_PICK
#undef FK_STEP
    FE_HIP(hipMemcpyAsync(&cr, k->ctl, sizeof(cr), hipMemcpyDeviceToHost, k->s));
    FE_HIP(hipMemcpyAsync(dist_io, k->kdist, (size_t)k->n * sizeof(double),
                          hipMemcpyDeviceToHost, k->s));
    FE_HIP(hipMemcpyAsync(assign_io, k->kassign, (size_t)k->n * sizeof(int32_t),
                          hipMemcpyDeviceToHost, k->s));
    FE_HIP(hipStreamSynchronize(k->s));
    *n_added = cr.n_done;
    if (final_max)
        *final_max = cr.last_max;
    if (centers_out && cr.n_done > 0) {
        FE_HIP(hipMemcpyAsync(centers_out, k->hist + first_label,
                              (size_t)cr.n_done * sizeof(int64_t),
                              hipMemcpyDeviceToHost, k->s));
        FE_HIP(hipStreamSynchronize(k->s));
    }
    return EK_OK;
}

// ===========================================================================
// PAM (k-medoids) sweep in feature space, resident on the device
// ===========================================================================
// Reference: enspara/cluster/kmedoids.py:575-699 (_kmedoids_pam_update, serial
// branch) for metrics 'euclidean' / 'manhattan' (libdist.pyx): per cluster
//   state_inds = where(assignments == cid); prop = choice(state_inds)      :611, :514
//   nd = metric(X, X[prop])                                                :637
//   distances > nd            -> (nd, cid)                                 :644
//   else assignments != cid   -> unchanged                                 :651
//   else                      -> assign_to_nearest_center(X[those], medoids
//                                with the proposal in place of medoid cid)  :658-666
//   accept iff mean(new**2) < mean(old**2), float64, numpy's summation     :478, :683
// -- a metric call, the read-back of n float64 and a dozen numpy passes over n
// per proposal when only the metric runs on the device.  Here the float64
// distances, the labels and the medoids' features stay in HBM; the host keeps
// the random stream (numpy's draws on raw outputs, ek_np_choice_draws) and the
// accept / reject decision: two waits per proposal.  Distances are computed with
// the arithmetic of feat_distance_kernel (FeatAcc, features in order) whatever
// the pairing of sample and medoid, so every number is the one the reference's
// loop -- metric(X[subset], center) per center, strict < in ascending center
// order (util.py:199-203) -- produces.
#include "ek_pw.h"

extern "C" int64_t ek_np_choice_draws(const uint32_t *raw, int64_t n_raw, int64_t *pos,
                                      const int64_t *m, int64_t count, int64_t *out);
// (ek_pam.hip: the scan of per-workgroup member counts and the chunk sums of the
// pairwise cost tree, each without the step that follows it there)
void ek_launch_scan_counts(const int32_t *blockcnt, int64_t n, int64_t *scan,
                           int64_t *total, hipStream_t s);
void ek_launch_pw_chunks(double *part, const EkPwShape *shapes, int n_full,
                         int n_leaves_total, int n_chunks, hipStream_t s);

struct FeatPam {
    int32_t K = 0, Kcap = 0;
    void *MT = nullptr;         // medoids' features, transposed: [F][Kcap] elements
    void *col = nullptr;        // [F] the column a proposal displaced
    int64_t *med = nullptr;     // [Kcap] the medoids' samples (for the table)
    int64_t *idx = nullptr;     // [1] the proposed sample (device)
    double *ndist = nullptr;    // trial state
    int32_t *nassign = nullptr;
    uint32_t *amb = nullptr;    // ambiguous members
    double *best_d = nullptr;
    int32_t *best_c = nullptr;
    unsigned int *counters = nullptr;   // [0] ambiguous members
    int32_t *blockcnt = nullptr;
    int64_t *scan = nullptr, *total = nullptr;
    double *part = nullptr;     // leaf sums + chunk sums (both columns)
    double *out2 = nullptr;
    EkPwShape *shapes = nullptr;
    int n_full = 0, n_leaves = 0, n_chunks = 0;
    // the sweep without a host round trip per proposal (round 4)
    struct FeatPamCtl *ctl = nullptr;   // device: stream position, status, last verdict
    uint32_t *raw_dev = nullptr;        // the caller's raw random outputs
    int64_t raw_cap = 0;
    int64_t *jdev = nullptr;            // [1] the member drawn
    int64_t *props_dev = nullptr;       // [Kcap] explicit proposals
    int32_t *accept_dev = nullptr;      // [Kcap]
    int32_t Kcap_async = 0;
    // windows of proposals (round 5): one pass over the samples for a window's distances
    struct FeatWin *win = nullptr;      // device: the window's draws and proposals
    void *Y = nullptr;                  // [FEAT_WIN][F] the proposals' features
    double *vecs = nullptr;             // [FEAT_WIN][n] every sample's distance to each
    int32_t *blockcntW = nullptr;       // [FEAT_WIN][workgroups] member counts
    int64_t *scanW = nullptr, *totalW = nullptr;
    int64_t n_windows = 0, n_stale = 0; // (since the context was made: a diagnostic)
    int win_width = 8;                  // slots of the next window of drawn proposals
    int plain_left = 0;                 // proposals to go one at a time before the next window
    // the ambiguous members' search, tiled (round 5)
    double *near_d = nullptr;           // [n][chunks of 256 medoids] a chunk's nearest
    int32_t *near_c = nullptr;
    unsigned int *near_tick = nullptr;  // [n / FN_MB + 1] arrivals per batch of members
    int32_t near_kc = 0;
};

#define FEAT_WIN 32     // proposals per window
#define FEAT_MD_CH 32   // features per LDS slice of the window's distance kernel
// a window's draws (numpy's choice on the raw outputs, one cluster after the other,
// from the member counts the window opens with) and proposals
struct FeatWin {
    long long pos_before[FEAT_WIN + 1]; // stream position before slot j's draw
    int64_t want[FEAT_WIN];             // the member drawn (-1: none)
    int64_t prop[FEAT_WIN];             // the proposed samples
    int32_t slot_status[FEAT_WIN];      // 0 drawn; 1 the raw outputs ran out; 2 empty
                                        // cluster; 3 not drawn (a slot before failed)
};

// device-side state of an asynchronous sweep
struct FeatPamCtl {
    long long pos;      // next raw output to use
    int32_t status;     // 0 ok; 1 the raw outputs ran out; 2 an empty cluster; 3 (windows)
                        // the window's draw for cluster win_stop no longer holds
    int32_t fail_cid;   // the cluster at which status was set
    int32_t acc;        // the last proposal was accepted
    uint32_t moved;     // (windows) clusters of the window whose member lists changed
    int32_t win_stop;
    int32_t pad;
};

extern "C" void ek_feat_pam_release(ek_feat *k)
{
    if (!k || !k->pam)
        return;
    FeatPam &p = *k->pam;
    (void)hipFree(p.MT);
    (void)hipFree(p.col);
    (void)hipFree(p.med);
    (void)hipFree(p.idx);
    (void)hipFree(p.ndist);
    (void)hipFree(p.nassign);
    (void)hipFree(p.amb);
    (void)hipFree(p.best_d);
    (void)hipFree(p.best_c);
    (void)hipFree(p.counters);
    (void)hipFree(p.blockcnt);
    (void)hipFree(p.scan);
    (void)hipFree(p.total);
    (void)hipFree(p.part);
    (void)hipFree(p.out2);
    (void)hipFree(p.shapes);
    (void)hipFree(p.ctl);
    (void)hipFree(p.raw_dev);
    (void)hipFree(p.jdev);
    (void)hipFree(p.props_dev);
    (void)hipFree(p.accept_dev);
    (void)hipFree(p.win);
    (void)hipFree(p.Y);
    (void)hipFree(p.vecs);
    (void)hipFree(p.blockcntW);
    (void)hipFree(p.scanW);
    (void)hipFree(p.totalW);
    (void)hipFree(p.near_d);
    (void)hipFree(p.near_c);
    (void)hipFree(p.near_tick);
    delete k->pam;
    k->pam = nullptr;
}

// MT[j][c] = feature j of sample med[c]
template <typename T>
__global__ void __launch_bounds__(EK_BLOCK)
feat_medoid_table_kernel(const T *__restrict__ tiles, int F,
                         const int64_t *__restrict__ med, int K, int Kcap,
                         T *__restrict__ MT)
{
    const int c = blockIdx.x;
    const int64_t f = med[c];
    const T *p = tiles + (size_t)(f / EK_TILE) * (size_t)F * EK_TILE + (f % EK_TILE);
    for (int j = threadIdx.x; j < F; j += EK_BLOCK)
        MT[(size_t)j * Kcap + c] = p[(size_t)j * EK_TILE];
}

// y = features of sample *idx; column cid of MT is saved in `col` and replaced by y
template <typename T>
__global__ void __launch_bounds__(EK_BLOCK)
feat_propose_kernel(const T *__restrict__ tiles, int F, const int64_t *__restrict__ idx,
                    int cid, int Kcap, T *__restrict__ MT, T *__restrict__ col,
                    T *__restrict__ y, unsigned int *__restrict__ counters)
{
    const int64_t f = idx[0];
    const T *p = tiles + (size_t)(f / EK_TILE) * (size_t)F * EK_TILE + (f % EK_TILE);
    for (int j = threadIdx.x; j < F; j += EK_BLOCK) {
        const T v = p[(size_t)j * EK_TILE];
        col[j] = MT[(size_t)j * Kcap + cid];
        MT[(size_t)j * Kcap + cid] = v;
        y[j] = v;
    }
    if (threadIdx.x == 0)
        counters[0] = 0;
}

template <typename T>
__global__ void __launch_bounds__(EK_BLOCK)
feat_restore_kernel(int F, int cid, int Kcap, T *__restrict__ MT,
                    const T *__restrict__ col)
{
    for (int j = threadIdx.x; j < F; j += EK_BLOCK)
        MT[(size_t)j * Kcap + cid] = col[j];
}

// kmedoids.py:644-658 on float64 distances
__global__ void __launch_bounds__(EK_BLOCK)
feat_pam_classify_kernel(const double *__restrict__ dist,
                         const int32_t *__restrict__ assign,
                         const double *__restrict__ nd, int64_t n, int32_t cid,
                         double *__restrict__ ndist, int32_t *__restrict__ nassign,
                         uint32_t *__restrict__ amb, unsigned int *__restrict__ counters,
                         const int32_t *__restrict__ halt = nullptr)
{
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (f >= n || (halt && *halt))
        return;
    const double d = dist[f], x = nd[f];
    const int32_t a = assign[f];
    if (d > x) {
        ndist[f] = x;
        nassign[f] = cid;
    } else if (a != cid) {
        ndist[f] = d;
        nassign[f] = a;
    } else {
        amb[atomicAdd(&counters[0], 1u)] = (uint32_t)f;
    }
}

// One workgroup per ambiguous member: threads stride the medoids in ascending
// order, every (member, medoid) distance is one thread's FeatAcc chain over the
// features in order; the workgroup keeps the smallest distance, the lowest medoid
// index among equal ones -- util.py:199-203's strict-< scan from +inf.
template <typename T, int METRIC>
__global__ void __launch_bounds__(EK_BLOCK)
feat_pam_nearest_kernel(const T *__restrict__ tiles, int F,
                        const uint32_t *__restrict__ amb,
                        const unsigned int *__restrict__ counters,
                        const T *__restrict__ MT, int K, int Kcap,
                        double *__restrict__ ndist, int32_t *__restrict__ nassign,
                        const int32_t *__restrict__ halt = nullptr)
{
    __shared__ T xs[FY_CHUNK];
    __shared__ double rv[EK_BLOCK / EK_WAVE];
    __shared__ int32_t rc[EK_BLOCK / EK_WAVE];
    // (the asynchronous sweep: the stream of draws ran out, or a cluster was
    // empty, earlier in this batch of proposals -- nothing of the batch's rest
    // is kept, so nothing of it is computed either)
    if (halt && *halt)
        return;
    // (any grid: workgroup b takes members b, b + gridDim.x, ..)
    for (unsigned int mem = blockIdx.x; mem < counters[0]; mem += gridDim.x) {
    __syncthreads();        // (rv / rc of the member before are read by then)
    const uint32_t f = amb[mem];
    const T *p = tiles + (size_t)(f / EK_TILE) * (size_t)F * EK_TILE + (f % EK_TILE);
    // (label 0 where no distance is below +inf -- overflowed squares --: what
    // util.py:186-203's zeros + strict < leave)
    double best = __builtin_inf();
    int32_t bc = 0;
    for (int c0 = 0; c0 < K; c0 += EK_BLOCK) {
        const int c = c0 + threadIdx.x;
        double acc = 0.0;
        for (int j0 = 0; j0 < F; j0 += FY_CHUNK) {
            const int w = (F - j0 < FY_CHUNK) ? (F - j0) : FY_CHUNK;
            __syncthreads();
            for (int j = threadIdx.x; j < w; j += EK_BLOCK)
                xs[j] = p[(size_t)(j0 + j) * EK_TILE];
            __syncthreads();
            if (c < K)
                for (int j = 0; j < w; ++j)
                    FeatAcc<T, METRIC>::add(acc, xs[j], MT[(size_t)(j0 + j) * Kcap + c]);
        }
        if (c < K) {
            acc = feat_finish<METRIC>(acc, F);
            if (acc < best) {               // ascending c per thread: strict <
                best = acc;
                bc = c;
            }
        }
    }
    // the smallest distance, the lowest index among equal ones
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(best, off, 64);
        const int32_t oc = __shfl_xor(bc, off, 64);
        if (ov < best || (ov == best && oc < bc)) {
            best = ov;
            bc = oc;
        }
    }
    if ((threadIdx.x & (EK_WAVE - 1)) == 0) {
        rv[threadIdx.x / EK_WAVE] = best;
        rc[threadIdx.x / EK_WAVE] = bc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < EK_BLOCK / EK_WAVE; ++w)
            if (rv[w] < best || (rv[w] == best && rc[w] < bc)) {
                best = rv[w];
                bc = rc[w];
            }
        ndist[f] = best;
        nassign[f] = bc;
    }
    }
}

// The same search tiled (round 5): one workgroup per member read the whole medoid
// table again -- a quarter of a gigabyte through the L2 per proposal at 1000 members
// x 1000 medoids x 64 features, 67 us.  Here a workgroup takes FN_MB members and 256
// medoids (thread = medoid, the members' features in LDS, FN_MB chains per thread,
// each still FeatAcc's chain over the features in order), a table column is read
// once per FN_MB members; the chunks' nearest go through memory to the workgroup
// that arrives last for the batch (ek_arrive_last), which takes the smallest
// distance, the lowest medoid index among equal ones.
#define FN_MB 8
#define FN_FC 128
template <typename T, int METRIC>
__global__ void __launch_bounds__(EK_BLOCK)
feat_pam_nearest_tiled_kernel(const T *__restrict__ tiles, int F,
                              const uint32_t *__restrict__ amb,
                              const unsigned int *__restrict__ counters,
                              const T *__restrict__ MT, int K, int Kcap,
                              double *__restrict__ ndist, int32_t *__restrict__ nassign,
                              const int32_t *__restrict__ halt, double *__restrict__ part_d,
                              int32_t *__restrict__ part_c, unsigned int *__restrict__ ticks)
{
    __shared__ T xs[FN_MB][FN_FC];
    __shared__ double rv[FN_MB][EK_BLOCK / EK_WAVE];
    __shared__ int32_t rc[FN_MB][EK_BLOCK / EK_WAVE];
    if (halt && *halt)
        return;
    const unsigned int n_amb = counters[0];
    const int KC = gridDim.y, kc = blockIdx.y;
    const int c = kc * EK_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & (EK_WAVE - 1), wv = threadIdx.x / EK_WAVE;
    for (unsigned int b = blockIdx.x; (size_t)b * FN_MB < n_amb; b += gridDim.x) {
        const unsigned int m0 = b * FN_MB;
        double acc[FN_MB];
#pragma unroll
        for (int m = 0; m < FN_MB; ++m)
            acc[m] = 0.0;
        for (int j0 = 0; j0 < F; j0 += FN_FC) {
            const int w = (F - j0 < FN_FC) ? (F - j0) : FN_FC;
            __syncthreads();
            for (int e = threadIdx.x; e < FN_MB * FN_FC; e += EK_BLOCK) {
                const int m = e / FN_FC, j = e % FN_FC;
                T v = (T)0;
                if (m0 + m < n_amb && j < w) {
                    const uint32_t f = amb[m0 + m];
                    v = tiles[(size_t)(f / EK_TILE) * (size_t)F * EK_TILE +
                              (size_t)(j0 + j) * EK_TILE + (f % EK_TILE)];
                }
                xs[m][j] = v;
            }
            __syncthreads();
            if (c < K) {
#pragma unroll 16
                for (int j = 0; j < w; ++j) {
                    const T y = MT[(size_t)(j0 + j) * Kcap + c];
#pragma unroll
                    for (int m = 0; m < FN_MB; ++m)
                        FeatAcc<T, METRIC>::add(acc[m], xs[m][j], y);
                }
            }
        }
        // this chunk's nearest medoid per member (label 0 where no distance is below
        // +inf -- overflowed squares --: what util.py:186-203's zeros + strict < leave)
#pragma unroll
        for (int m = 0; m < FN_MB; ++m) {
            double best = __builtin_inf();
            int32_t bc = 0;
            if (c < K) {
                const double a = feat_finish<METRIC>(acc[m], F);
                if (a < best) {
                    best = a;
                    bc = c;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(best, off, 64);
                const int32_t oc = __shfl_xor(bc, off, 64);
                if (ov < best || (ov == best && oc < bc)) {
                    best = ov;
                    bc = oc;
                }
            }
            if (lane == 0) {
                rv[m][wv] = best;
                rc[m][wv] = bc;
            }
        }
        __syncthreads();
        if (threadIdx.x < FN_MB && m0 + threadIdx.x < n_amb) {
            const int m = threadIdx.x;
            double best = rv[m][0];
            int32_t bc = rc[m][0];
            for (int q = 1; q < EK_BLOCK / EK_WAVE; ++q)
                if (rv[m][q] < best || (rv[m][q] == best && rc[m][q] < bc)) {
                    best = rv[m][q];
                    bc = rc[m][q];
                }
            if (KC == 1) {
                const uint32_t f = amb[m0 + m];
                ndist[f] = best;
                nassign[f] = bc;
            } else {
                __hip_atomic_store(&part_d[(size_t)(m0 + m) * KC + kc], best, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                ek_coh_store(&part_c[(size_t)(m0 + m) * KC + kc], bc);
            }
        }
        if (KC > 1 && ek_arrive_last(&ticks[b], (unsigned int)KC)) {
            if (threadIdx.x < FN_MB && m0 + threadIdx.x < n_amb) {
                const int m = threadIdx.x;
                double best = __builtin_inf();
                int32_t bc = 0;
                for (int q = 0; q < KC; ++q) {
                    const double ov = __hip_atomic_load(&part_d[(size_t)(m0 + m) * KC + q],
                                                        __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
                    const int32_t oc = ek_coh_load(&part_c[(size_t)(m0 + m) * KC + q]);
                    if (ov < best || (ov == best && oc < bc)) {
                        best = ov;
                        bc = oc;
                    }
                }
                const uint32_t f = amb[m0 + m];
                ndist[f] = best;
                nassign[f] = bc;
            }
            if (threadIdx.x == 0)
                __hip_atomic_store(&ticks[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- the sweep without a host round trip per proposal (round 4) -------------------
// The draw is numpy's RandomState.choice(m) on the raw 32-bit outputs
// (kmedoids.py:514; the host form is ek_np_choice_draws): mask to the bits of
// m - 1, reject above it; m == 1 consumes nothing.  An empty cluster or a stream
// that runs out stops the sweep: every later kernel of it returns at once.
// an accepted trial state becomes the state (kmedoids.py:684-690) and, in the same
// sweep over the labels, the members of cluster `cid` are counted per workgroup
// (cid < 0: the commit alone, after the last proposal)
__global__ void __launch_bounds__(EK_BLOCK)
feat_commit_count_kernel(const FeatPamCtl *__restrict__ ctl, long long n,
                         const double *__restrict__ ndist,
                         const int32_t *__restrict__ nassign, double *__restrict__ dist,
                         int32_t *__restrict__ assign, int do_commit, int cid,
                         int32_t *__restrict__ blockcnt)
{
    __shared__ int cnt;
    if (threadIdx.x == 0)
        cnt = 0;
    __syncthreads();
    const long long f = (long long)blockIdx.x * EK_BLOCK + threadIdx.x;
    if (f < n) {
        int32_t a;
        if (do_commit && !ctl->status && ctl->acc) {
            dist[f] = ndist[f];
            a = nassign[f];
            assign[f] = a;
        } else {
            a = assign[f];
        }
        if (cid >= 0 && a == cid)
            atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (cid >= 0 && threadIdx.x == 0)
        blockcnt[blockIdx.x] = cnt;
}

// one workgroup: the draw, the member it names
// (ek_select_member_multi_kernel's search) and the proposal's features into y and
// into the medoid table (feat_propose_kernel)
template <typename T>
__global__ void __launch_bounds__(EK_BLOCK)
feat_pick_kernel(FeatPamCtl *__restrict__ ctl, int cid, const int64_t *__restrict__ total,
                 const uint32_t *__restrict__ raw, long long n_raw,
                 const int64_t *__restrict__ props, const int32_t *__restrict__ assign,
                 long long n, const int64_t *__restrict__ scan, int nblocks,
                 const T *__restrict__ tiles, int F, int Kcap, T *__restrict__ MT,
                 T *__restrict__ col, T *__restrict__ y, int64_t *__restrict__ idx,
                 unsigned int *__restrict__ counters)
{
    __shared__ long long s_want, s_f;
    __shared__ int s_go, s_lo;
    __shared__ int wcnt[EK_BLOCK / EK_WAVE];
    if (threadIdx.x == 0) {
        s_go = 0;
        s_want = -1;
        s_f = -1;
        counters[0] = 0;
        if (!ctl->status) {
            if (props) {
                s_f = props[cid];
                s_go = 1;
            } else {
                const long long m = total[0];
                if (m <= 0) {
                    ctl->status = 2;
                    ctl->fail_cid = cid;
                } else {
                    const unsigned long long rng = (unsigned long long)(m - 1);
                    if (rng == 0) {
                        s_want = 0;
                        s_go = 1;
                    } else {
                        unsigned long long mask = rng;
                        mask |= mask >> 1;
                        mask |= mask >> 2;
                        mask |= mask >> 4;
                        mask |= mask >> 8;
                        mask |= mask >> 16;
                        long long p = ctl->pos;
                        for (;;) {
                            if (p >= n_raw) {
                                ctl->status = 1;
                                ctl->fail_cid = cid;
                                break;
                            }
                            const unsigned long long v = raw[p++] & mask;
                            if (v <= rng) {
                                s_want = (long long)v;
                                ctl->pos = p;
                                s_go = 1;
                                break;
                            }
                        }
                    }
                }
            }
        }
        if (s_go && s_want >= 0) {      // last workgroup whose scan <= want
            int lo = 0, hi = nblocks - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) / 2;
                if (scan[mid] <= s_want)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            s_lo = lo;
        }
    }
    __syncthreads();
    if (!s_go)
        return;
    if (s_want >= 0) {
        const int lo = s_lo;
        const long long rank = s_want - scan[lo];
        const long long f = (long long)lo * EK_BLOCK + threadIdx.x;
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
            s_f = f;
        __syncthreads();
    }
    const long long f = s_f;
    if (f < 0)
        return;             // (cannot happen: the count said the member exists)
    if (threadIdx.x == 0)
        idx[0] = f;
    const T *p = tiles + (size_t)(f / EK_TILE) * (size_t)F * EK_TILE + (f % EK_TILE);
    for (int j = threadIdx.x; j < F; j += EK_BLOCK) {
        const T v = p[(size_t)j * EK_TILE];
        col[j] = MT[(size_t)j * Kcap + cid];
        MT[(size_t)j * Kcap + cid] = v;
        y[j] = v;
    }
}

// distance of every sample to the proposal (feat_distance_kernel's chain) and its
// classification (kmedoids.py:644-658) in one sweep
template <typename T, int METRIC>
__global__ void __launch_bounds__(EK_BLOCK)
feat_dist_classify_kernel(const T *__restrict__ tiles, const T *__restrict__ y,
                          int64_t n, int F, const double *__restrict__ dist,
                          const int32_t *__restrict__ assign, int32_t cid,
                          double *__restrict__ ndist, int32_t *__restrict__ nassign,
                          uint32_t *__restrict__ amb, unsigned int *__restrict__ counters,
                          const int32_t *__restrict__ halt)
{
    __shared__ T ys[FY_CHUNK];
    if (halt && *halt)          // (see feat_pam_nearest_kernel: up to 127 passes over
        return;                 // all samples for nothing otherwise)
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    const T *p = tiles + (size_t)(f / EK_TILE) * (size_t)F * EK_TILE + (f % EK_TILE);
    double acc = 0.0;
    for (int j0 = 0; j0 < F; j0 += FY_CHUNK) {
        const int w = (F - j0 < FY_CHUNK) ? (F - j0) : FY_CHUNK;
        __syncthreads();
        for (int j = threadIdx.x; j < w; j += EK_BLOCK)
            ys[j] = y[j0 + j];
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < w; ++j)
            FeatAcc<T, METRIC>::add(acc, __builtin_nontemporal_load(
                                             p + (size_t)(j0 + j) * EK_TILE),
                                    ys[j]);
    }
    if (f >= n)
        return;
    acc = feat_finish<METRIC>(acc, F);
    const double d = dist[f], x = acc;
    const int32_t a = assign[f];
    if (d > x) {
        ndist[f] = x;
        nassign[f] = cid;
    } else if (a != cid) {
        ndist[f] = d;
        nassign[f] = a;
    } else {
        amb[atomicAdd(&counters[0], 1u)] = (uint32_t)f;
    }
}

// the two cost sums (the chunk sums added left to right: ek_pw_total_kernel) and the
// verdict (kmedoids.py:478-479, :683: np.square(x).mean() of either state, strictly
// lower wins; the table's column back if not)
template <typename T>
__global__ void __launch_bounds__(EK_BLOCK)
feat_total_decide_kernel(FeatPamCtl *__restrict__ ctl, const double *__restrict__ chunksum,
                         int n_chunks, long long n, int cid,
                         const int64_t *__restrict__ idx, int32_t *__restrict__ accept,
                         int64_t *__restrict__ med, int F, int Kcap, T *__restrict__ MT,
                         const T *__restrict__ col)
{
    __shared__ double sums[2];
    __shared__ double cs[2 * EK_BLOCK];
    if (ctl->status)
        return;
    // (the chunk sums through LDS, EK_BLOCK chunks at a time: one after the other from
    // memory, a trip each, this was 12 us at 123 chunks)
    double run = 0.0;
    for (int c0 = 0; c0 < n_chunks; c0 += EK_BLOCK) {
        const int w = (n_chunks - c0 < EK_BLOCK) ? (n_chunks - c0) : EK_BLOCK;
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * w; e += EK_BLOCK)
            cs[e] = chunksum[2 * (size_t)c0 + e];
        __syncthreads();
        if (threadIdx.x < 2)
            for (int c = 0; c < w; ++c)
                run = run + cs[2 * c + threadIdx.x];
    }
    if (threadIdx.x < 2)
        sums[threadIdx.x] = run;
    __syncthreads();
    const double old_cost = sums[0] / (double)n, new_cost = sums[1] / (double)n;
    const bool acc = new_cost < old_cost;
    if (!acc)
        for (int j = threadIdx.x; j < F; j += EK_BLOCK)
            MT[(size_t)j * Kcap + cid] = col[j];
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl->acc = acc ? 1 : 0;
        accept[cid] = acc ? 1 : 0;
        if (acc)
            med[cid] = idx[0];
    }
}

// numpy's leaf (ek_pam.hip, "cost sums in numpy's order") over the squares of
// float64 values: np.square(x) rounds each square, then the pairwise sum
__global__ void __launch_bounds__(EK_BLOCK)
feat_pw_leaf_kernel(const double *__restrict__ a, const double *__restrict__ b,
                    const EkPwShape *__restrict__ shapes, int n_full,
                    int n_leaves_total, double *__restrict__ leafsum,
                    const int32_t *__restrict__ halt = nullptr)
{
    if (halt && *halt)
        return;
    const int g = blockIdx.x * (EK_BLOCK / 8) + threadIdx.x / 8;
    const int l8 = threadIdx.x & 7;
    if (g >= n_leaves_total)
        return;                     // (whole groups of eight lanes)
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
    for (int i = 0; i < body; i += 8) {
        const double va = a[off + i + l8], vb = b[off + i + l8];
        if (i == 0) {
            ra = va * va;
            rb = vb * vb;
        } else {
            ra = ra + va * va;
            rb = rb + vb * vb;
        }
    }
    if (body > 0) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {       // (r0+r1)+(r2+r3) ...
            ra = ra + __shfl_xor(ra, o, 8);
            rb = rb + __shfl_xor(rb, o, 8);
        }
    }
    if (l8 == 0) {
        for (int i = body; i < len; ++i) {      // sequential tail
            const double va = a[off + i], vb = b[off + i];
            ra = ra + va * va;
            rb = rb + vb * vb;
        }
        leafsum[2 * (size_t)g + 0] = ra;
        leafsum[2 * (size_t)g + 1] = rb;
    }
}

// ---- windows of proposals (round 5) ------------------------------------------------------------
// A proposal's pass over all samples for its distances was three quarters of its
// time; a window's proposals are known when it opens -- drawn from the member
// counts of its clusters as they stand then -- so ONE pass gives every sample's
// distance to each of them (FeatAcc's chain per pair, as before).  A draw stops
// holding when an accepted earlier proposal of the window moved a sample into or
// out of its cluster (kmedoids.py:611-614 draws from the member list of the
// moment): the commit keeps a mask of such clusters, the slot's first kernel
// stops the window there (status 3) and the host opens the next one at that cluster.

// one thread: the window's draws, in cluster order, on the counts it opens with
__global__ void feat_window_draw_kernel(FeatPamCtl *__restrict__ ctl, FeatWin *__restrict__ win,
                                        int cid0, int cnt, const int64_t *__restrict__ total,
                                        const uint32_t *__restrict__ raw, long long n_raw,
                                        const int64_t *__restrict__ props)
{
    if (threadIdx.x != 0 || ctl->status)
        return;
    ctl->moved = 0;
    long long pos = ctl->pos;
    bool failed = false;
    for (int j = 0; j < cnt; ++j) {
        win->pos_before[j] = pos;
        win->want[j] = -1;
        win->slot_status[j] = failed ? 3 : 0;
        if (failed)
            continue;
        if (props) {
            win->prop[j] = props[cid0 + j];
            continue;
        }
        const long long m = total[j];
        if (m <= 0) {
            win->slot_status[j] = 2;
            failed = true;
            continue;
        }
        const unsigned long long rng = (unsigned long long)(m - 1);
        if (rng == 0) {
            win->want[j] = 0;
            continue;
        }
        unsigned long long mask = rng;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        for (;;) {
            if (pos >= n_raw) {
                win->slot_status[j] = 1;
                failed = true;
                break;
            }
            const unsigned long long v = raw[pos++] & mask;
            if (v <= rng) {
                win->want[j] = (long long)v;
                break;
            }
        }
        if (failed)
            pos = win->pos_before[j];
    }
    win->pos_before[cnt] = pos;
}

// Y[j][:] = the features of slot j's proposal
template <typename T>
__global__ void __launch_bounds__(EK_BLOCK)
feat_window_gather_kernel(const T *__restrict__ tiles, int F, const FeatWin *__restrict__ win,
                          T *__restrict__ Y, const int32_t *__restrict__ halt)
{
    if (*halt)
        return;
    const int j = blockIdx.x;
    const int64_t f = win->prop[j];
    const bool ok = win->slot_status[j] == 0 && f >= 0;
    const T *p = tiles + (size_t)((ok ? f : 0) / EK_TILE) * (size_t)F * EK_TILE +
                 ((ok ? f : 0) % EK_TILE);
    for (int q = threadIdx.x; q < F; q += EK_BLOCK)
        Y[(size_t)j * F + q] = ok ? p[(size_t)q * EK_TILE] : (T)0;
}

// vecs[g][f] = distance of sample f to proposal g: one read of the samples for the
// whole window, every pair one thread's FeatAcc chain over the features in order
template <typename T, int METRIC, int W>
__global__ void __launch_bounds__(EK_BLOCK)
feat_multi_distance_kernel(const T *__restrict__ tiles, const T *__restrict__ Y, int64_t n,
                           int F, int cnt, double *__restrict__ vecs,
                           const int32_t *__restrict__ halt)
{
    __shared__ T ys[FEAT_MD_CH][W];
    if (*halt)
        return;
    const int64_t f = (int64_t)blockIdx.x * EK_BLOCK + threadIdx.x;
    const T *p = tiles + (size_t)(f / EK_TILE) * (size_t)F * EK_TILE + (f % EK_TILE);
    double acc[W];
#pragma unroll
    for (int g = 0; g < W; ++g)
        acc[g] = 0.0;
    for (int j0 = 0; j0 < F; j0 += FEAT_MD_CH) {
        const int w = (F - j0 < FEAT_MD_CH) ? (F - j0) : FEAT_MD_CH;
        __syncthreads();
        for (int e = threadIdx.x; e < FEAT_MD_CH * W; e += EK_BLOCK) {
            const int g = e / FEAT_MD_CH, j = e % FEAT_MD_CH;
            ys[j][g] = (g < cnt && j < w) ? Y[(size_t)g * F + j0 + j] : (T)0;
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < w; ++j) {
            const T x = __builtin_nontemporal_load(p + (size_t)(j0 + j) * EK_TILE);
#pragma unroll
            for (int g = 0; g < W; ++g)
                FeatAcc<T, METRIC>::add(acc[g], x, ys[j][g]);
        }
    }
    if (f >= n)
        return;
#pragma unroll
    for (int g = 0; g < W; ++g)
        if (g < cnt)
            vecs[(size_t)g * n + f] = feat_finish<METRIC>(acc[g], F);
}

// a slot's first kernel (one workgroup): does its draw still hold, did it succeed;
// then the proposal into the medoid table (feat_propose_kernel)
template <typename T>
__global__ void __launch_bounds__(EK_BLOCK)
feat_slot_begin_kernel(FeatPamCtl *__restrict__ ctl, const FeatWin *__restrict__ win,
                       int cid0, int j, int check_stale, int F, int Kcap,
                       T *__restrict__ MT, T *__restrict__ col, const T *__restrict__ Y,
                       int64_t *__restrict__ idx, unsigned int *__restrict__ counters)
{
    __shared__ int s_go;
    if (threadIdx.x == 0) {
        s_go = 0;
        if (!ctl->status) {
            if (check_stale && ((ctl->moved >> j) & 1u)) {
                ctl->status = 3;
                ctl->win_stop = cid0 + j;
                ctl->pos = win->pos_before[j];
            } else if (win->slot_status[j] != 0) {
                ctl->status = win->slot_status[j];
                ctl->fail_cid = cid0 + j;
                ctl->pos = win->pos_before[j];
            } else {
                ctl->pos = win->pos_before[j + 1];
                idx[0] = win->prop[j];
                counters[0] = 0;
                s_go = 1;
            }
        }
    }
    __syncthreads();
    if (!s_go)
        return;
    const int cid = cid0 + j;
    for (int q = threadIdx.x; q < F; q += EK_BLOCK) {
        col[q] = MT[(size_t)q * Kcap + cid];
        MT[(size_t)q * Kcap + cid] = Y[(size_t)j * F + q];
    }
}

// an accepted trial state becomes the state (kmedoids.py:684-690); the clusters of the
// window that lose or gain a sample by it are marked
__global__ void __launch_bounds__(EK_BLOCK)
feat_commit_mask_kernel(FeatPamCtl *__restrict__ ctl, long long n,
                        const double *__restrict__ ndist, const int32_t *__restrict__ nassign,
                        double *__restrict__ dist, int32_t *__restrict__ assign, int cid0,
                        int cnt)
{
    __shared__ uint32_t s_m;
    if (ctl->status || !ctl->acc)
        return;
    if (threadIdx.x == 0)
        s_m = 0;
    __syncthreads();
    const long long f = (long long)blockIdx.x * EK_BLOCK + threadIdx.x;
    uint32_t m = 0;
    if (f < n) {
        const int32_t a = assign[f], na = nassign[f];
        dist[f] = ndist[f];
        if (a != na) {
            assign[f] = na;
            const int ia = a - cid0, ib = na - cid0;
            if (ia >= 0 && ia < cnt)
                m |= 1u << ia;
            if (ib >= 0 && ib < cnt)
                m |= 1u << ib;
        }
    }
    if (m)
        atomicOr(&s_m, m);
    __syncthreads();
    if (threadIdx.x == 0 && s_m)
        atomicOr(&ctl->moved, s_m);
}

static int feat_pam_alloc(ek_feat *k, FeatPam &p, int32_t K)
{
    const size_t n = (size_t)std::max<int64_t>(k->n, 1);
    const size_t nb = (n + EK_BLOCK - 1) / EK_BLOCK;
    if (!p.ndist) {
        FE_HIP(hipMalloc((void **)&p.ndist, n * sizeof(double)));
        FE_HIP(hipMalloc((void **)&p.nassign, n * sizeof(int32_t)));
        FE_HIP(hipMalloc((void **)&p.amb, n * sizeof(uint32_t)));
        FE_HIP(hipMalloc((void **)&p.counters, 4 * sizeof(unsigned int)));
        FE_HIP(hipMalloc((void **)&p.blockcnt, nb * sizeof(int32_t)));
        FE_HIP(hipMalloc((void **)&p.scan, nb * sizeof(int64_t)));
        FE_HIP(hipMalloc((void **)&p.total, sizeof(int64_t)));
        FE_HIP(hipMalloc((void **)&p.idx, sizeof(int64_t)));
        FE_HIP(hipMalloc((void **)&p.col, (size_t)k->F * k->esize));
        FE_HIP(hipMalloc((void **)&p.out2, 2 * sizeof(double)));
        EkPwShape hs[2];
        const int64_t n_full = k->n / EK_PW_CHUNK;
        const int last_len = (int)(k->n - n_full * EK_PW_CHUNK);
        ek_pw_build_shape(n_full > 0 ? EK_PW_CHUNK : 0, &hs[0]);
        ek_pw_build_shape(last_len, &hs[1]);
        p.n_full = (int)n_full;
        p.n_leaves = (int)n_full * EK_PW_FULL_LEAVES + hs[1].n_leaves;
        p.n_chunks = (int)n_full + (last_len > 0 ? 1 : 0);
        if (n_full > 0 && hs[0].n_leaves != EK_PW_FULL_LEAVES)
            return ek_set_error(EK_ESTATE, "ek_feat_pam_sweep: unexpected shape of a "
                                           "full chunk's pairwise sum");
        FE_HIP(hipMalloc((void **)&p.shapes, sizeof(hs)));
        FE_HIP(hipMemcpy(p.shapes, hs, sizeof(hs), hipMemcpyHostToDevice));
        FE_HIP(hipMalloc((void **)&p.part, (2 * (size_t)std::max(p.n_leaves, 1) +
                                            2 * (size_t)std::max(p.n_chunks, 1)) *
                                               sizeof(double)));
    }
    if (K > p.Kcap) {
        FE_HIP(hipStreamSynchronize(k->s));
        (void)hipFree(p.MT);
        (void)hipFree(p.med);
        p.MT = nullptr;
        p.med = nullptr;
        p.Kcap = 0;
        FE_HIP(hipMalloc((void **)&p.MT, (size_t)k->F * K * k->esize));
        FE_HIP(hipMalloc((void **)&p.med, (size_t)K * sizeof(int64_t)));
        p.Kcap = K;
    }
    p.K = K;
    // the tiled search's hand-over between the chunks of 256 medoids
    const int KC = (K + EK_BLOCK - 1) / EK_BLOCK;
    if (!p.near_tick) {
        const size_t nt = n / FN_MB + 2;
        FE_HIP(hipMalloc((void **)&p.near_tick, nt * sizeof(unsigned int)));
        FE_HIP(hipMemsetAsync(p.near_tick, 0, nt * sizeof(unsigned int), k->s));
    }
    if (KC > 1 && KC > p.near_kc) {
        FE_HIP(hipStreamSynchronize(k->s));
        (void)hipFree(p.near_d);
        (void)hipFree(p.near_c);
        p.near_d = nullptr;
        p.near_c = nullptr;
        p.near_kc = 0;
        FE_HIP(hipMalloc((void **)&p.near_d, n * (size_t)KC * sizeof(double)));
        FE_HIP(hipMalloc((void **)&p.near_c, n * (size_t)KC * sizeof(int32_t)));
        p.near_kc = KC;
    }
    return EK_OK;
}

// One sweep over clusters *cid .. n_medoids - 1 (kmedoids.py:575-699) from the
// state (dist_io float64, assign_io int32) -- uploaded when *cid == 0, written
// back when the sweep is through.  proposals == NULL: numpy's draws on `raw`
// (ek_np_choice_draws), *pos outputs consumed.  medoids[c] is replaced and
// accept[c] set where proposal c was accepted.
// *status: 0 done; 1 `raw` ran out at cluster *cid (call again with more: the
// state stays on the device); 2 cluster *cid has no member (choice raises).
// proposals c0 .. c1 - 1 in round 4's form (a pass over the samples each), the last
// one's verdict applied: eight launches per proposal -- [commit of the proposal before
// + member count], scan, [draw + member + proposal], [distances + classification], the
// ambiguous members' search, leaf sums, chunk sums, [totals + verdict]
template <typename T, int M>
static void feat_enqueue_plain(ek_feat *k, FeatPam &p, int32_t K, int32_t c0, int32_t c1,
                               bool have_props, int64_t raw_left, dim3 near_grid)
{
    const int nb = (int)((k->n + EK_BLOCK - 1) / EK_BLOCK);
    const unsigned blocks = (unsigned)nb;
    const int per = EK_BLOCK / 8;
    for (int32_t cid = c0; cid < c1; ++cid) {
        hipLaunchKernelGGL(feat_commit_count_kernel, dim3(blocks), dim3(EK_BLOCK), 0, k->s,
                           p.ctl, (long long)k->n, p.ndist, p.nassign, k->kdist, k->kassign,
                           cid > c0 ? 1 : 0, have_props ? -1 : cid, p.blockcnt);
        if (!have_props)
            ek_launch_scan_counts(p.blockcnt, k->n, p.scan, p.total, k->s);
        hipLaunchKernelGGL((feat_pick_kernel<T>), dim3(1), dim3(EK_BLOCK), 0, k->s, p.ctl, cid,
                           p.total, p.raw_dev, (long long)raw_left,
                           have_props ? p.props_dev : (const int64_t *)nullptr, k->kassign,
                           (long long)k->n, p.scan, nb, (const T *)k->tiles, k->F, p.Kcap,
                           (T *)p.MT, (T *)p.col, (T *)k->y, p.idx, p.counters);
        hipLaunchKernelGGL((feat_dist_classify_kernel<T, M>), dim3(blocks), dim3(EK_BLOCK), 0,
                           k->s, (const T *)k->tiles, (const T *)k->y, k->n, k->F, k->kdist,
                           k->kassign, cid, p.ndist, p.nassign, p.amb, p.counters,
                           &p.ctl->status);
        hipLaunchKernelGGL((feat_pam_nearest_tiled_kernel<T, M>), near_grid, dim3(EK_BLOCK), 0,
                           k->s, (const T *)k->tiles, k->F, p.amb, p.counters,
                           (const T *)p.MT, K, p.Kcap, p.ndist, p.nassign, &p.ctl->status,
                           p.near_d, p.near_c, p.near_tick);
        hipLaunchKernelGGL(feat_pw_leaf_kernel, dim3((p.n_leaves + per - 1) / per),
                           dim3(EK_BLOCK), 0, k->s, k->kdist, p.ndist, p.shapes, p.n_full,
                           p.n_leaves, p.part, &p.ctl->status);
        ek_launch_pw_chunks(p.part, p.shapes, p.n_full, p.n_leaves, p.n_chunks, k->s);
        hipLaunchKernelGGL((feat_total_decide_kernel<T>), dim3(1), dim3(EK_BLOCK), 0, k->s,
                           p.ctl, p.part + 2 * (size_t)p.n_leaves, p.n_chunks, (long long)k->n,
                           cid, p.idx, p.accept_dev, p.med, k->F, p.Kcap, (T *)p.MT,
                           (const T *)p.col);
    }
    // (the verdict on the batch's last proposal)
    hipLaunchKernelGGL(feat_commit_count_kernel, dim3(blocks), dim3(EK_BLOCK), 0, k->s, p.ctl,
                       (long long)k->n, p.ndist, p.nassign, k->kdist, k->kassign, 1, -1,
                       p.blockcnt);
}

static void feat_enqueue_plain_any(ek_feat *k, int32_t metric, FeatPam &p, int32_t K,
                                   int32_t c0, int32_t c1, bool have_props, int64_t raw_left,
                                   dim3 near_grid)
{
    if (k->kind == 2) {      // hamming on integer samples (libdist.pyx:77-95)
        feat_enqueue_plain<long long, 2>(k, p, K, c0, c1, have_props, raw_left, near_grid);
    } else if (k->kind == 0) {
        if (metric == 0)
            feat_enqueue_plain<float, 0>(k, p, K, c0, c1, have_props, raw_left, near_grid);
        else
            feat_enqueue_plain<float, 1>(k, p, K, c0, c1, have_props, raw_left, near_grid);
    } else {
        if (metric == 0)
            feat_enqueue_plain<double, 0>(k, p, K, c0, c1, have_props, raw_left, near_grid);
        else
            feat_enqueue_plain<double, 1>(k, p, K, c0, c1, have_props, raw_left, near_grid);
    }
}

extern "C" int ek_feat_pam_sweep(ek_feat *k, int32_t metric, int32_t n_medoids,
                                 int64_t *medoids, const int64_t *proposals,
                                 const uint32_t *raw, int64_t n_raw, int64_t *pos,
                                 double *dist_io, int32_t *assign_io,
                                 int32_t *accept, int32_t *cid_io, int32_t *status)
{
    if (!k || !medoids || !dist_io || !assign_io || !accept || !cid_io || !status ||
        !pos || n_medoids < 1 || metric < 0 || metric > 2)
        return ek_set_error(EK_EARG, "ek_feat_pam_sweep: bad argument (metrics: "
                                     "euclidean 0, manhattan 1, hamming 2)");
    if (!k->loaded)
        return ek_set_error(EK_ESTATE, "ek_feat_pam_sweep: samples have to be loaded");
    if ((metric == 2) != (k->kind == 2))
        return ek_set_error(EK_EARG, "ek_feat_pam_sweep: hamming needs integer "
                                     "samples, the other metrics floating point");
    if (k->n < 1 || k->n > 0xffffffffLL)
        return ek_set_error(EK_EARG, "ek_feat_pam_sweep: %lld samples",
                            (long long)k->n);
    const int32_t K = n_medoids;
    for (int32_t c = 0; c < K; ++c)
        if (medoids[c] < 0 || medoids[c] >= k->n ||
            (proposals && (proposals[c] < 0 || proposals[c] >= k->n)))
            return ek_set_error(EK_EARG, "ek_feat_pam_sweep: medoid or proposal %d "
                                         "out of range", c);
    FE_HIP(hipSetDevice(k->device));
    if (!k->pam) {
        k->pam = new (std::nothrow) FeatPam();
        if (!k->pam)
            return ek_set_error(EK_ENOMEM, "ek_feat_pam_sweep: out of host memory");
    }
    FeatPam &p = *k->pam;
    int rc = feat_pam_alloc(k, p, K);
    if (rc)
        return rc;
    const int nb = (int)((k->n + EK_BLOCK - 1) / EK_BLOCK);
    if (!k->kdist) {
        FE_HIP(hipMalloc((void **)&k->kdist, (size_t)k->n * sizeof(double)));
        FE_HIP(hipMalloc((void **)&k->kassign, (size_t)k->n * sizeof(int32_t)));
        FE_HIP(hipMalloc((void **)&k->bm, (size_t)nb * sizeof(FeatBlockMax)));
        FE_HIP(hipMalloc((void **)&k->ctl, sizeof(FeatCtl)));
    }
    *status = 0;
    int32_t cid = *cid_io;
    if (cid == 0) {
        FE_HIP(hipMemcpyAsync(k->kdist, dist_io, (size_t)k->n * sizeof(double),
                              hipMemcpyHostToDevice, k->s));
        FE_HIP(hipMemcpyAsync(k->kassign, assign_io, (size_t)k->n * sizeof(int32_t),
                              hipMemcpyHostToDevice, k->s));
        // the medoids' features
        FE_HIP(hipMemcpyAsync(p.med, medoids, (size_t)K * sizeof(int64_t),
                              hipMemcpyHostToDevice, k->s));
        if (k->kind == 2)
            hipLaunchKernelGGL(feat_medoid_table_kernel<long long>, dim3(K), dim3(EK_BLOCK),
                               0, k->s, (const long long *)k->tiles, k->F, p.med, K,
                               p.Kcap, (long long *)p.MT);
        else if (k->kind == 0)
            hipLaunchKernelGGL(feat_medoid_table_kernel<float>, dim3(K), dim3(EK_BLOCK),
                               0, k->s, (const float *)k->tiles, k->F, p.med, K,
                               p.Kcap, (float *)p.MT);
        else
            hipLaunchKernelGGL(feat_medoid_table_kernel<double>, dim3(K),
                               dim3(EK_BLOCK), 0, k->s, (const double *)k->tiles, k->F,
                               p.med, K, p.Kcap, (double *)p.MT);
        FE_HIP(hipGetLastError());
        FE_HIP(hipStreamSynchronize(k->s));
    }
    const unsigned blocks = (unsigned)nb;
    // ---- round 5: windows of FEAT_WIN proposals, their distances in one pass ------------
    // (EK_FEAT_PAM_WINDOWS=0: round 4's form below, a pass over the samples per proposal)
    // Given proposals: always.  Drawn ones: only on request (EK_FEAT_PAM_WINDOWS=1) --
    // on the data measured a draw stops holding every 5 to 8 proposals, and a window
    // that short costs more to open than its one pass over the samples saves.
    const char *fw_env = getenv("EK_FEAT_PAM_WINDOWS");
    const bool fw_forced = fw_env && fw_env[0] == '1';
    // (a window pays where the pass over the samples is most of a proposal: the samples
    // well beyond the caches.  Drawn proposals: while the draws hold -- a window that
    // ends within its first slots costs more to open than its one pass saves, the next
    // hundred proposals then go one at a time before a window is tried again)
    const bool fw_big = (int64_t)k->n * k->F * k->esize >= (64ll << 20);
    if (!getenv("EK_FEAT_PAM_SYNC") && !(fw_env && fw_env[0] == '0')) {
        if (!p.ctl) {
            FE_HIP(hipMalloc((void **)&p.ctl, sizeof(FeatPamCtl)));
            FE_HIP(hipMalloc((void **)&p.jdev, sizeof(int64_t)));
            FE_HIP(hipMemsetAsync(p.jdev, 0, sizeof(int64_t), k->s));
        }
        if (!p.win) {
            FE_HIP(hipMalloc((void **)&p.win, sizeof(FeatWin)));
            FE_HIP(hipMalloc((void **)&p.Y, (size_t)FEAT_WIN * k->F * k->esize));
            FE_HIP(hipMalloc((void **)&p.vecs, (size_t)FEAT_WIN * k->n * sizeof(double)));
            FE_HIP(hipMalloc((void **)&p.blockcntW, (size_t)FEAT_WIN * nb * sizeof(int32_t)));
            FE_HIP(hipMalloc((void **)&p.scanW, (size_t)FEAT_WIN * nb * sizeof(int64_t)));
            FE_HIP(hipMalloc((void **)&p.totalW, FEAT_WIN * sizeof(int64_t)));
        }
        if (!p.accept_dev || K > p.Kcap_async) {
            FE_HIP(hipStreamSynchronize(k->s));
            (void)hipFree(p.accept_dev);
            (void)hipFree(p.props_dev);
            p.accept_dev = nullptr;
            p.props_dev = nullptr;
            FE_HIP(hipMalloc((void **)&p.accept_dev, (size_t)K * sizeof(int32_t)));
            FE_HIP(hipMalloc((void **)&p.props_dev, (size_t)K * sizeof(int64_t)));
            p.Kcap_async = K;
        }
        const int64_t raw_left = proposals ? 0 : std::max<int64_t>(n_raw - *pos, 0);
        if (raw_left > p.raw_cap) {
            FE_HIP(hipStreamSynchronize(k->s));
            (void)hipFree(p.raw_dev);
            p.raw_dev = nullptr;
            p.raw_cap = 0;
            FE_HIP(hipMalloc((void **)&p.raw_dev, (size_t)raw_left * sizeof(uint32_t)));
            p.raw_cap = raw_left;
        }
        if (raw_left > 0)       // (positions on the device count from *pos)
            FE_HIP(hipMemcpyAsync(p.raw_dev, raw + *pos, (size_t)raw_left * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, k->s));
        if (proposals)
            FE_HIP(hipMemcpyAsync(p.props_dev, proposals, (size_t)K * sizeof(int64_t),
                                  hipMemcpyHostToDevice, k->s));
        FeatPamCtl hc;
        memset(&hc, 0, sizeof(hc));
        FE_HIP(hipMemcpyAsync(p.ctl, &hc, sizeof(hc), hipMemcpyHostToDevice, k->s));
        FE_HIP(hipMemsetAsync(p.accept_dev, 0, (size_t)K * sizeof(int32_t), k->s));
        const int32_t cid_start = cid;
        const dim3 near_grid((unsigned)std::min<int64_t>((k->n + FN_MB - 1) / FN_MB, 512),
                             (unsigned)((K + EK_BLOCK - 1) / EK_BLOCK));
        const int per = EK_BLOCK / 8;
        const int32_t *halt = &p.ctl->status;
        while (cid < K) {
            // given proposals: nothing can end a window early, four of them are enqueued
            // before the control block is read; drawn ones: one window, as wide as
            // the draws have lately held (an accepted proposal that takes samples
            // from or gives samples to a later cluster of the window ends it there)
            // (hamming: one proposal at a time -- the windows' kernels are built for the
            // floating-point metrics only)
            const bool use_win = metric != 2 &&
                                 (fw_forced || (fw_big && (proposals || p.plain_left <= 0)));
            if (!use_win) {
                const int32_t c1 = std::min<int32_t>(
                    K, cid + (p.plain_left > 0 ? std::min(128, p.plain_left) : 128));
                feat_enqueue_plain_any(k, metric, p, K, cid, c1, proposals != nullptr, raw_left,
                                       near_grid);
                FE_HIP(hipGetLastError());
                FE_HIP(hipMemcpyAsync(&hc, p.ctl, sizeof(hc), hipMemcpyDeviceToHost, k->s));
                FE_HIP(hipStreamSynchronize(k->s));
                if (hc.status)
                    break;
                p.plain_left -= c1 - cid;
                cid = c1;
                continue;
            }
            int32_t enq = cid;
            const int first_cid0 = cid;
            for (int w = 0; w < (proposals ? 4 : 1) && enq < K; ++w) {
                const int32_t cid0 = enq;
                const int cnt = std::min<int32_t>(proposals ? FEAT_WIN : p.win_width, K - cid0);
                ++p.n_windows;
                if (!proposals)
                    ek_launch_count_members_multi(k->kassign, k->n, cid0, cnt, p.blockcntW,
                                                  p.scanW, p.totalW, k->s);
                hipLaunchKernelGGL(feat_window_draw_kernel, dim3(1), dim3(EK_WAVE), 0, k->s,
                                   p.ctl, p.win, cid0, cnt, p.totalW, p.raw_dev,
                                   (long long)raw_left,
                                   proposals ? p.props_dev : (const int64_t *)nullptr);
                if (!proposals)
                    ek_launch_select_member_multi(k->kassign, k->n, cid0, cnt, p.scanW,
                                                  p.win->want, p.win->prop, k->s);
#define FW_OPEN(T, M)                                                          \
    do {                                                                       \
        hipLaunchKernelGGL((feat_window_gather_kernel<T>), dim3(cnt), dim3(EK_BLOCK), 0, \
                           k->s, (const T *)k->tiles, k->F, p.win, (T *)p.Y, halt); \
        if (cnt <= 4)                                                          \
            hipLaunchKernelGGL((feat_multi_distance_kernel<T, M, 4>), dim3(blocks), \
                               dim3(EK_BLOCK), 0, k->s, (const T *)k->tiles,   \
                               (const T *)p.Y, k->n, k->F, cnt, p.vecs, halt); \
        else if (cnt <= 8)                                                     \
            hipLaunchKernelGGL((feat_multi_distance_kernel<T, M, 8>), dim3(blocks), \
                               dim3(EK_BLOCK), 0, k->s, (const T *)k->tiles,   \
                               (const T *)p.Y, k->n, k->F, cnt, p.vecs, halt); \
        else if (cnt <= 16)                                                    \
            hipLaunchKernelGGL((feat_multi_distance_kernel<T, M, 16>), dim3(blocks), \
                               dim3(EK_BLOCK), 0, k->s, (const T *)k->tiles,   \
                               (const T *)p.Y, k->n, k->F, cnt, p.vecs, halt); \
        else                                                                   \
            hipLaunchKernelGGL((feat_multi_distance_kernel<T, M, FEAT_WIN>), dim3(blocks), \
                               dim3(EK_BLOCK), 0, k->s, (const T *)k->tiles,   \
                               (const T *)p.Y, k->n, k->F, cnt, p.vecs, halt); \
    } while (0)
#define FW_SLOT(T, M)                                                          \
    do {                                                                       \
        hipLaunchKernelGGL((feat_slot_begin_kernel<T>), dim3(1), dim3(EK_BLOCK), 0, k->s, \
                           p.ctl, p.win, cid0, j, proposals ? 0 : 1, k->F, p.Kcap, \
                           (T *)p.MT, (T *)p.col, (const T *)p.Y, p.idx, p.counters); \
        hipLaunchKernelGGL(feat_pam_classify_kernel, dim3(blocks), dim3(EK_BLOCK), 0, \
                           k->s, k->kdist, k->kassign, p.vecs + (size_t)j * k->n, k->n, \
                           cid0 + j, p.ndist, p.nassign, p.amb, p.counters, halt); \
        hipLaunchKernelGGL((feat_pam_nearest_tiled_kernel<T, M>), near_grid,   \
                           dim3(EK_BLOCK), 0, k->s, (const T *)k->tiles, k->F, \
                           p.amb, p.counters, (const T *)p.MT, K, p.Kcap,      \
                           p.ndist, p.nassign, halt, p.near_d, p.near_c, p.near_tick); \
        hipLaunchKernelGGL(feat_pw_leaf_kernel, dim3((p.n_leaves + per - 1) / per), \
                           dim3(EK_BLOCK), 0, k->s, k->kdist, p.ndist, p.shapes, \
                           p.n_full, p.n_leaves, p.part, halt);                \
        ek_launch_pw_chunks(p.part, p.shapes, p.n_full, p.n_leaves, p.n_chunks, k->s); \
        hipLaunchKernelGGL((feat_total_decide_kernel<T>), dim3(1), dim3(EK_BLOCK), 0, \
                           k->s, p.ctl, p.part + 2 * (size_t)p.n_leaves, p.n_chunks, \
                           (long long)k->n, cid0 + j, p.idx, p.accept_dev, p.med, k->F, \
                           p.Kcap, (T *)p.MT, (const T *)p.col);               \
        hipLaunchKernelGGL(feat_commit_mask_kernel, dim3(blocks), dim3(EK_BLOCK), 0, k->s, \
                           p.ctl, (long long)k->n, p.ndist, p.nassign, k->kdist, \
                           k->kassign, cid0, cnt);                             \
    } while (0)
#define FW_ALL(T, M)                                                           \
    do {                                                                       \
        FW_OPEN(T, M);                                                         \
        for (int j = 0; j < cnt; ++j)                                          \
            FW_SLOT(T, M);                                                     \
    } while (0)
                if (k->kind == 0) {
                    if (metric == 0)
                        FW_ALL(float, 0);
                    else
                        FW_ALL(float, 1);
                } else {
                    if (metric == 0)
                        FW_ALL(double, 0);
                    else
                        FW_ALL(double, 1);
                }
#undef FW_ALL
#undef FW_SLOT
#undef FW_OPEN
                enq += cnt;
            }
            FE_HIP(hipGetLastError());
            FE_HIP(hipMemcpyAsync(&hc, p.ctl, sizeof(hc), hipMemcpyDeviceToHost, k->s));
            FE_HIP(hipStreamSynchronize(k->s));
            if (!proposals && hc.status != 3)
                p.win_width = std::min(FEAT_WIN, 2 * p.win_width);
            if (hc.status == 3) {
                // a draw no longer held: the next window opens at that cluster
                ++p.n_stale;
                p.win_width = std::max(2, std::min(FEAT_WIN, hc.win_stop - first_cid0 + 1));
                if (!fw_forced && hc.win_stop - first_cid0 < 6) {
                    p.plain_left = 96;
                    p.win_width = 8;
                }
                cid = hc.win_stop;
                hc.status = 0;
                hc.moved = 0;
                FE_HIP(hipMemcpyAsync(p.ctl, &hc, sizeof(hc), hipMemcpyHostToDevice, k->s));
                continue;
            }
            if (hc.status)
                break;
            cid = enq;
        }
        if (getenv("EK_FEAT_PAM_VERBOSE"))
            fprintf(stderr, "ek_feat_pam_sweep: %lld windows so far, %lld ended where a draw "
                            "no longer held\n", (long long)p.n_windows, (long long)p.n_stale);
        // what was decided: clusters cid_start .. (the stop)
        const int32_t cid_end = hc.status ? hc.fail_cid : K;
        if (cid_end > cid_start) {
            std::vector<int64_t> hm((size_t)K);
            FE_HIP(hipMemcpyAsync(accept + cid_start, p.accept_dev + cid_start,
                                  (size_t)(cid_end - cid_start) * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, k->s));
            FE_HIP(hipMemcpyAsync(hm.data(), p.med, (size_t)K * sizeof(int64_t),
                                  hipMemcpyDeviceToHost, k->s));
            FE_HIP(hipStreamSynchronize(k->s));
            for (int32_t c = cid_start; c < cid_end; ++c)
                if (accept[c])
                    medoids[c] = hm[(size_t)c];
        }
        *pos += hc.pos;
        if (hc.status) {
            *cid_io = hc.fail_cid;
            *status = hc.status;
            return EK_OK;
        }
        FE_HIP(hipMemcpyAsync(dist_io, k->kdist, (size_t)k->n * sizeof(double),
                              hipMemcpyDeviceToHost, k->s));
        FE_HIP(hipMemcpyAsync(assign_io, k->kassign, (size_t)k->n * sizeof(int32_t),
                              hipMemcpyDeviceToHost, k->s));
        FE_HIP(hipStreamSynchronize(k->s));
        *cid_io = K;
        return EK_OK;
    }
    // ---- round 4: the whole sweep enqueued, no host round trip per proposal ----------
    // The draw (numpy's choice on the raw outputs), the choice of the member, the
    // verdict and the commit are kernels; the host reads the control block every
    // 128 proposals.  Same kernels for the arithmetic, same results; the loop below
    // (EK_FEAT_PAM_SYNC=1) is the form with two waits per proposal.
    if (!getenv("EK_FEAT_PAM_SYNC")) {
        if (!p.ctl) {
            FE_HIP(hipMalloc((void **)&p.ctl, sizeof(FeatPamCtl)));
            FE_HIP(hipMalloc((void **)&p.jdev, sizeof(int64_t)));
            FE_HIP(hipMemsetAsync(p.jdev, 0, sizeof(int64_t), k->s));
        }
        if (!p.accept_dev || K > p.Kcap_async) {
            FE_HIP(hipStreamSynchronize(k->s));
            (void)hipFree(p.accept_dev);
            (void)hipFree(p.props_dev);
            p.accept_dev = nullptr;
            p.props_dev = nullptr;
            FE_HIP(hipMalloc((void **)&p.accept_dev, (size_t)K * sizeof(int32_t)));
            FE_HIP(hipMalloc((void **)&p.props_dev, (size_t)K * sizeof(int64_t)));
            p.Kcap_async = K;
        }
        const int64_t raw_left = proposals ? 0 : std::max<int64_t>(n_raw - *pos, 0);
        if (raw_left > p.raw_cap) {
            FE_HIP(hipStreamSynchronize(k->s));
            (void)hipFree(p.raw_dev);
            p.raw_dev = nullptr;
            p.raw_cap = 0;
            FE_HIP(hipMalloc((void **)&p.raw_dev, (size_t)raw_left * sizeof(uint32_t)));
            p.raw_cap = raw_left;
        }
        if (raw_left > 0)       // (positions on the device count from *pos)
            FE_HIP(hipMemcpyAsync(p.raw_dev, raw + *pos, (size_t)raw_left * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, k->s));
        if (proposals)
            FE_HIP(hipMemcpyAsync(p.props_dev, proposals, (size_t)K * sizeof(int64_t),
                                  hipMemcpyHostToDevice, k->s));
        FeatPamCtl hc;
        memset(&hc, 0, sizeof(hc));
        FE_HIP(hipMemcpyAsync(p.ctl, &hc, sizeof(hc), hipMemcpyHostToDevice, k->s));
        FE_HIP(hipMemsetAsync(p.accept_dev, 0, (size_t)K * sizeof(int32_t), k->s));
        const int32_t cid_start = cid;
        const dim3 near_grid((unsigned)std::min<int64_t>((k->n + FN_MB - 1) / FN_MB, 512),
                             (unsigned)((K + EK_BLOCK - 1) / EK_BLOCK));
        while (cid < K) {
            const int32_t stop = std::min(K, cid + 128);
            feat_enqueue_plain_any(k, metric, p, K, cid, stop, proposals != nullptr, raw_left,
                                   near_grid);
            cid = stop;
            FE_HIP(hipGetLastError());
            FE_HIP(hipMemcpyAsync(&hc, p.ctl, sizeof(hc), hipMemcpyDeviceToHost, k->s));
            FE_HIP(hipStreamSynchronize(k->s));
            if (hc.status)
                break;
        }
        // what was decided: clusters cid_start .. (the stop)
        const int32_t cid_end = hc.status ? hc.fail_cid : K;
        if (cid_end > cid_start) {
            std::vector<int64_t> hm((size_t)K);
            FE_HIP(hipMemcpyAsync(accept + cid_start, p.accept_dev + cid_start,
                                  (size_t)(cid_end - cid_start) * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, k->s));
            FE_HIP(hipMemcpyAsync(hm.data(), p.med, (size_t)K * sizeof(int64_t),
                                  hipMemcpyDeviceToHost, k->s));
            FE_HIP(hipStreamSynchronize(k->s));
            for (int32_t c = cid_start; c < cid_end; ++c)
                if (accept[c])
                    medoids[c] = hm[(size_t)c];
        }
        *pos += hc.pos;
        if (hc.status) {
            *cid_io = hc.fail_cid;
            *status = hc.status;
            return EK_OK;
        }
        FE_HIP(hipMemcpyAsync(dist_io, k->kdist, (size_t)k->n * sizeof(double),
                              hipMemcpyDeviceToHost, k->s));
        FE_HIP(hipMemcpyAsync(assign_io, k->kassign, (size_t)k->n * sizeof(int32_t),
                              hipMemcpyDeviceToHost, k->s));
        FE_HIP(hipStreamSynchronize(k->s));
        *cid_io = K;
        return EK_OK;
    }
    for (; cid < K; ++cid) {
        // ---- the proposal: a member drawn like choice(state_inds), or given ----------
        ek_launch_count_members(k->kassign, k->n, cid, p.blockcnt, p.scan, p.total, k->s);
        int64_t m = 0;
        FE_HIP(hipMemcpyAsync(&m, p.total, sizeof(int64_t), hipMemcpyDeviceToHost, k->s));
        FE_HIP(hipStreamSynchronize(k->s));
        if (!proposals) {
            if (m <= 0) {
                *cid_io = cid;
                *status = 2;
                return EK_OK;
            }
            int64_t j = 0;
            if (ek_np_choice_draws(raw, n_raw, pos, &m, 1, &j) != 1) {
                *cid_io = cid;
                *status = 1;
                return EK_OK;
            }
            ek_launch_select_member(k->kassign, k->n, cid, p.scan, j, p.idx, k->s);
        } else {
            FE_HIP(hipMemcpyAsync(p.idx, &proposals[cid], sizeof(int64_t),
                                  hipMemcpyHostToDevice, k->s));
        }
#define FP_T(T, M)                                                             \
    do {                                                                       \
        hipLaunchKernelGGL((feat_propose_kernel<T>), dim3(1), dim3(EK_BLOCK), 0, k->s, \
                           (const T *)k->tiles, k->F, p.idx, cid, p.Kcap,      \
                           (T *)p.MT, (T *)p.col, (T *)k->y, p.counters);      \
        hipLaunchKernelGGL((feat_distance_kernel<T, M>), dim3(blocks),         \
                           dim3(EK_BLOCK), 0, k->s, (const T *)k->tiles,       \
                           (const T *)k->y, k->n, k->F, k->out);               \
        hipLaunchKernelGGL(feat_pam_classify_kernel, dim3(blocks), dim3(EK_BLOCK), 0, \
                           k->s, k->kdist, k->kassign, k->out, k->n, cid, p.ndist, \
                           p.nassign, p.amb, p.counters);                      \
        if (m > 0)                                                             \
            hipLaunchKernelGGL((feat_pam_nearest_kernel<T, M>), dim3((unsigned)m), \
                               dim3(EK_BLOCK), 0, k->s, (const T *)k->tiles, k->F, \
                               p.amb, p.counters, (const T *)p.MT, K, p.Kcap,  \
                               p.ndist, p.nassign);                            \
    } while (0)
        if (k->kind == 2) {
            FP_T(long long, 2);
        } else if (k->kind == 0) {
            if (metric == 0)
                FP_T(float, 0);
            else
                FP_T(float, 1);
        } else {
            if (metric == 0)
                FP_T(double, 0);
            else
                FP_T(double, 1);
        }
#undef FP_T
        // ---- cost of the state and of the trial state, numpy's order ------------------
        const int per = EK_BLOCK / 8;
        hipLaunchKernelGGL(feat_pw_leaf_kernel, dim3((p.n_leaves + per - 1) / per),
                           dim3(EK_BLOCK), 0, k->s, k->kdist, p.ndist, p.shapes, p.n_full,
                           p.n_leaves, p.part);
        ek_launch_pw_chunks_total(p.part, p.shapes, p.n_full, p.n_leaves, p.n_chunks,
                                  p.out2, k->s);
        FE_HIP(hipGetLastError());
        double sums[2] = {0.0, 0.0};
        int64_t prop = -1;
        FE_HIP(hipMemcpyAsync(sums, p.out2, sizeof(sums), hipMemcpyDeviceToHost, k->s));
        FE_HIP(hipMemcpyAsync(&prop, p.idx, sizeof(int64_t), hipMemcpyDeviceToHost, k->s));
        FE_HIP(hipStreamSynchronize(k->s));
        // np.square(x).mean(): the pairwise sum divided by n (kmedoids.py:478-479)
        const double old_cost = sums[0] / (double)k->n, new_cost = sums[1] / (double)k->n;
        const bool acc = new_cost < old_cost;               // :683
        accept[cid] = acc ? 1 : 0;
        if (acc) {
            std::swap(k->kdist, p.ndist);
            std::swap(k->kassign, p.nassign);
            medoids[cid] = prop;
        } else {
            if (k->kind == 2)
                hipLaunchKernelGGL(feat_restore_kernel<long long>, dim3(1), dim3(EK_BLOCK), 0,
                                   k->s, k->F, cid, p.Kcap, (long long *)p.MT,
                                   (const long long *)p.col);
            else if (k->kind == 0)
                hipLaunchKernelGGL(feat_restore_kernel<float>, dim3(1), dim3(EK_BLOCK), 0,
                                   k->s, k->F, cid, p.Kcap, (float *)p.MT,
                                   (const float *)p.col);
            else
                hipLaunchKernelGGL(feat_restore_kernel<double>, dim3(1), dim3(EK_BLOCK),
                                   0, k->s, k->F, cid, p.Kcap, (double *)p.MT,
                                   (const double *)p.col);
        }
    }
    FE_HIP(hipMemcpyAsync(dist_io, k->kdist, (size_t)k->n * sizeof(double),
                          hipMemcpyDeviceToHost, k->s));
    FE_HIP(hipMemcpyAsync(assign_io, k->kassign, (size_t)k->n * sizeof(int32_t),
                          hipMemcpyDeviceToHost, k->s));
    FE_HIP(hipStreamSynchronize(k->s));
    *cid_io = K;
    return EK_OK;
}
