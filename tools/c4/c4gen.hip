// c4gen.hip -- measurement infrastructure (tools/), not part of the product
// library: a synthetic-trajectory generator that runs ON the device, so that
// BASELINE.json configs[3] (10^7 frames x 500 atoms = 60 GB of coordinates)
// can be put into HBM in a second instead of the half hour numpy's
// generator (enspara_amd/synth.py) takes for it on the host.
//
// Every value is integer arithmetic on a counter-based hash up to the final
// int -> float32 conversion, so tools/c4/c4gen.py restates it in numpy BIT FOR
// BIT for any subset of the frames (the centers a run picked, a sample of a
// shard) -- what the oracle checks of tools/c4_one_gpu.py are made with.
//
//   frame f = template[hash(f) % T] + noise, rotated by a rational rotation
//   matrix (integer quaternion / its squared norm), translated; coordinates in
//   units of 1e-6 nm until the last step.
#include <hip/hip_runtime.h>
#include <stdint.h>

__host__ __device__ static inline uint64_t c4_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ static inline uint64_t c4_key(uint64_t seed, uint64_t f, uint64_t a,
                                                  uint64_t k)
{
    return c4_mix(c4_mix(seed * 0x100000001B3ull + f) + (a * 8ull + k));
}

__host__ __device__ static inline int64_t c4_floor_div(int64_t num, int64_t den)
{
    int64_t q = num / den;
    if ((num % den) != 0 && num < 0)
        --q;
    return q;
}

#define C4_FRAME 0xFFFFFFFFull   // the "atom" of a frame's own draws

// element e = (frame first + e / A, atom e % A): its three coordinates
__host__ __device__ static inline void c4_element(float *o3, const int32_t *tmpl, int64_t T,
                                                  int32_t A, int64_t first, int64_t e,
                                                  uint64_t seed)
{
    const int64_t fl = e / A;
    const int32_t a = (int32_t)(e - fl * A);
    const uint64_t f = (uint64_t)(first + fl);
    const uint64_t which = c4_key(seed, f, C4_FRAME, 0) % (uint64_t)T;
    const uint64_t hq = c4_key(seed, f, C4_FRAME, 1);
    int64_t w = (int64_t)(hq & 0xFFFF) - 32768, x = (int64_t)((hq >> 16) & 0xFFFF) - 32768,
            y = (int64_t)((hq >> 32) & 0xFFFF) - 32768, z = (int64_t)((hq >> 48) & 0xFFFF) - 32768;
    int64_t N = w * w + x * x + y * y + z * z;
    if (N < (1 << 24)) {
        w = 1; x = y = z = 0; N = 1;
    }
    const int64_t R[9] = {w * w + x * x - y * y - z * z, 2 * (x * y - z * w), 2 * (x * z + y * w),
                          2 * (x * y + z * w), w * w - x * x + y * y - z * z, 2 * (y * z - x * w),
                          2 * (x * z - y * w), 2 * (y * z + x * w), w * w - x * x - y * y + z * z};
    int64_t v[3];
    for (int k = 0; k < 3; ++k) {
        const uint64_t h = c4_key(seed, f, (uint64_t)a, (uint64_t)k);
        const int64_t s = (int64_t)(h & 0xFFFF) + (int64_t)((h >> 16) & 0xFFFF) +
                          (int64_t)((h >> 32) & 0xFFFF) + (int64_t)((h >> 48) & 0xFFFF) - 131070;
        // Irwin-Hall(4) scaled to sigma = 0.05 nm = 50 000 units
        const int64_t noise = c4_floor_div(s * 86603, 65536);
        v[k] = (int64_t)tmpl[((int64_t)which * A + a) * 3 + k] + noise;
    }
    for (int i = 0; i < 3; ++i) {
        const uint64_t ht = c4_key(seed, f, C4_FRAME, 2 + (uint64_t)i);
        const int64_t t = (int64_t)(((ht & 0xFFFFFF) * 10000000ull) >> 24) - 5000000;
        const int64_t r = c4_floor_div(R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2], N);
        o3[i] = (float)(r + t) * 1e-6f;
    }
}

__global__ void c4gen_kernel(float *__restrict__ out, const int32_t *__restrict__ tmpl,
                             int64_t T, int32_t A, int64_t first, int64_t count,
                             uint64_t seed)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count * A)
        return;
    float o[3];
    c4_element(o, tmpl, T, A, first, e, seed);
    out[e * 3] = o[0];
    out[e * 3 + 1] = o[1];
    out[e * 3 + 2] = o[2];
}

// the same function compiled for the host (tests: the numpy restatement of
// tools/c4/c4gen.py against the source the device runs, without a GPU)
extern "C" void c4gen_frames_host(float *out, const int32_t *tmpl, int64_t T, int32_t A,
                                  int64_t first, int64_t count, uint64_t seed)
{
    for (int64_t e = 0; e < count * (int64_t)A; ++e)
        c4_element(out + e * 3, tmpl, T, A, first, e, seed);
}

// out: device float32 [count][A][3]; tmpl: device int32 [T][A][3] (1e-6 nm)
extern "C" int c4gen_frames(float *out, const int32_t *tmpl, int64_t T, int32_t A,
                            int64_t first, int64_t count, uint64_t seed, void *stream)
{
    if (count <= 0)
        return 0;
    const int64_t total = count * (int64_t)A;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7FFFFFFFll)
        return -2;
    hipLaunchKernelGGL(c4gen_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       out, tmpl, T, A, first, count, seed);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
