// Probe (measurement only, round 6): is memory from hipExtMallocWithFlags(hipDeviceMallocUncached)
// coherent between the XCDs of ONE MI355X over its whole size -- what the mailboxes of
// csrc/ek_mshard.hip rely on -- and does it come free of an earlier owner's cache lines?
//
//   1. "previous life": a plain hipMalloc buffer is written by workgroups on every XCD and
//      read back (lines resident in every L2), then freed; the buffer under test is allocated
//      right after (the allocator usually hands the same range out again) and zeroed.
//   2. readers: one workgroup per XCD reads the whole buffer with system-scope loads (what
//      ek_msg_load does), reports what it saw that was not zero, says "ready" and polls a flag.
//   3. a writer on another stream stores a pattern with system-scope stores, waits for them,
//      raises the flag.
//   4. the readers read again and count the words that are not the pattern (stale lines).
//
// build: hipcc --offload-arch=gfx950 -O2 -o uc_probe uc_probe.hip;  usage: uc_probe [KB ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__device__ __forceinline__ unsigned sys_load(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(unsigned *p, unsigned v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void pollute(unsigned *buf, size_t words, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < words; i += blockDim.x) {
        buf[i] = 0x5a5a0000u + (unsigned)blockIdx.x;       // (every workgroup: lines in every L2)
        acc += buf[i];
    }
    if (acc == 1u)
        *sink = acc;
}

// out[8 * wg + ..]: XCC id, non-zero words on the first read, stale words on the second,
// first stale word, its value
__global__ void reader(const unsigned *buf, size_t words, unsigned *ctl, unsigned *out,
                       unsigned want_base)
{
    __shared__ unsigned s_nz, s_stale, s_first, s_val;
    if (threadIdx.x == 0) {
        s_nz = s_stale = 0;
        s_first = 0xffffffffu;
        s_val = 0;
    }
    __syncthreads();
    unsigned nz = 0;
    for (size_t i = threadIdx.x; i < words; i += blockDim.x)
        nz += sys_load(buf + i) != 0u;
    atomicAdd(&s_nz, nz);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctl + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (sys_load(ctl + 16) != 1u && wall_clock64() - t0 < 500000000ull)
            __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    unsigned stale = 0;
    for (size_t i = threadIdx.x; i < words; i += blockDim.x) {
        const unsigned v = sys_load(buf + i);
        if (v != want_base + (unsigned)i) {
            ++stale;
            if (atomicMin(&s_first, (unsigned)i) > (unsigned)i)
                s_val = v;
        }
    }
    atomicAdd(&s_stale, stale);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned *o = out + 8 * blockIdx.x;
        o[0] = xcc & 0xf;
        o[1] = s_nz;
        o[2] = s_stale;
        o[3] = s_first;
        o[4] = s_val;
    }
}

__global__ void writer(unsigned *buf, size_t words, unsigned *ctl, unsigned base, unsigned n_readers)
{
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (sys_load(ctl + 0) < n_readers && wall_clock64() - t0 < 500000000ull)
            __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    for (size_t i = threadIdx.x; i < words; i += blockDim.x)
        sys_store(buf + i, base + (unsigned)i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        sys_store(ctl + 16, 1u);
}

static void run(size_t kb, bool uncached, bool previous_life)
{
    const size_t bytes = kb << 10, words = bytes / 4;
    unsigned *sink = nullptr, *ctl = nullptr, *out = nullptr, *buf = nullptr;
    CHECK(hipMalloc((void **)&sink, 4096));
    CHECK(hipExtMallocWithFlags((void **)&ctl, 4096, hipDeviceMallocUncached));
    CHECK(hipMalloc((void **)&out, 64 * 8 * 4));
    CHECK(hipMemset(ctl, 0, 4096));
    CHECK(hipMemset(out, 0, 64 * 8 * 4));
    void *old = nullptr;
    if (previous_life) {
        CHECK(hipMalloc(&old, bytes));
        hipLaunchKernelGGL(pollute, dim3(64), dim3(256), 0, 0, (unsigned *)old, words, sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipFree(old));
    }
    if (uncached)
        CHECK(hipExtMallocWithFlags((void **)&buf, bytes, hipDeviceMallocUncached));
    else
        CHECK(hipMalloc((void **)&buf, bytes));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    hipStream_t sr, sw;
    CHECK(hipStreamCreateWithFlags(&sr, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sw, hipStreamNonBlocking));
    const unsigned n_readers = 16, base = 0xab000000u;
    hipLaunchKernelGGL(reader, dim3(n_readers), dim3(256), 0, sr, buf, words, ctl, out, base);
    hipLaunchKernelGGL(writer, dim3(1), dim3(256), 0, sw, buf, words, ctl, base, n_readers);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h(n_readers * 8);
    CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    unsigned nz = 0, stale = 0, first = 0xffffffffu, val = 0, xccs = 0;
    for (unsigned r = 0; r < n_readers; ++r) {
        nz += h[8 * r + 1];
        stale += h[8 * r + 2];
        xccs |= 1u << h[8 * r + 0];
        if (h[8 * r + 3] < first) {
            first = h[8 * r + 3];
            val = h[8 * r + 4];
        }
    }
    printf("%6zu KB %-9s %s same address as the polluted buffer: %-3s | XCDs of the readers 0x%02x | "
           "not zero at first: %u words | stale after the flag: %u of %zu word reads",
           kb, uncached ? "uncached" : "ordinary", previous_life ? "after a previous life," : "fresh,",
           previous_life ? ((void *)buf == old ? "yes" : "no") : "-", xccs, nz, stale,
           words * n_readers);
    if (stale)
        printf(" (first at word %u: 0x%08x)", first, val);
    printf("\n");
    CHECK(hipStreamDestroy(sr));
    CHECK(hipStreamDestroy(sw));
    CHECK(hipFree(buf));
    CHECK(hipFree(ctl));
    CHECK(hipFree(out));
    CHECK(hipFree(sink));
}

int main(int argc, char **argv)
{
    std::vector<size_t> sizes;
    for (int i = 1; i < argc; ++i)
        sizes.push_back((size_t)atol(argv[i]));
    if (sizes.empty())
        sizes = {4, 16, 32, 64, 256, 1024, 4096};
    for (size_t kb : sizes) {
        run(kb, true, false);
        run(kb, true, true);
        run(kb, false, true);
    }
    return 0;
}
