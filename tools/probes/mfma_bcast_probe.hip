// Probe (measurement only): the broadcast controls of v_mfma_f32_16x16x1_4b_f32.
// For every (cbsz, abid, blgp) tried: which A lane and which B lane does
// D[lane l][reg r] multiply?  Prints the mapping as formulas that fit, or dumps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CBSZ, int ABID, int BLGP> __global__ void k(float *out)
{
    const int l = threadIdx.x;
    v16f c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float a = (float)(l + 1);
    const float b = (float)(l + 1) * 1000.f;
    c = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, CBSZ, ABID, BLGP);
    for (int r = 0; r < 16; ++r)
        out[l * 16 + r] = c[r];
}

static void decode(const char *name, float *d)
{
    float h[1024];
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // value = (la + 1) * (lb + 1) * 1000; la, lb in 0..63.  Not unique in
    // general, so test hypotheses instead: A lane = 16 ba + 4 (l/16) + r%4,
    // B lane = 16 bb + l%16 with (ba, bb) functions of the block r/4
    printf("%s:", name);
    for (int blk = 0; blk < 4; ++blk) {
        int found = 0;
        for (int ba = 0; ba < 4 && !found; ++ba)
            for (int bb = 0; bb < 4 && !found; ++bb) {
                bool ok = true;
                for (int l = 0; l < 64 && ok; ++l)
                    for (int r = 4 * blk; r < 4 * blk + 4; ++r) {
                        const double want = (16.0 * ba + 4 * (l / 16) + r % 4 + 1) *
                                            (16.0 * bb + l % 16 + 1) * 1000.0;
                        if (fabs(want - h[l * 16 + r]) > 1e-3 * want) {
                            ok = false;
                            break;
                        }
                    }
                if (ok) {
                    printf("  block %d: A block %d, B block %d;", blk, ba, bb);
                    found = 1;
                }
            }
        if (!found)
            printf("  block %d: ???;", blk);
    }
    printf("\n");
}

int main()
{
    float *d;
    hipMalloc(&d, 4096);
#define RUN(C, A, B)                                                           \
    k<C, A, B><<<1, 64>>>(d);                                                  \
    decode("cbsz " #C " abid " #A " blgp " #B, d);
    RUN(0, 0, 0)
    RUN(0, 0, 1)
    RUN(0, 0, 2)
    RUN(0, 0, 3)
    RUN(0, 0, 4)
    RUN(0, 0, 5)
    RUN(0, 0, 6)
    RUN(0, 0, 7)
    RUN(1, 0, 0)
    RUN(1, 1, 0)
    RUN(2, 0, 0)
    RUN(2, 1, 0)
    RUN(2, 2, 0)
    RUN(2, 3, 0)
    RUN(2, 1, 6)
    return 0;
}
