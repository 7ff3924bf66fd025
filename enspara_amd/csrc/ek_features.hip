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
