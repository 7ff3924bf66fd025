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

#include <stdlib.h>

#include <algorithm>
#include <new>

extern int ek_set_error(int code, const char *fmt, ...);

#define FT_CHUNK 32           // features per staged transposition chunk
#define FY_CHUNK 2048         // target-point features staged in LDS at a time

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
extern "C" int ek_feat_destroy(ek_feat *k)
{
    if (!k)
        return EK_OK;
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
