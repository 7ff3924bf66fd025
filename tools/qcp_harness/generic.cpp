#define __device__
#define __forceinline__ inline
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include "ek_qcp.h"
int main(int argc, char **argv) {
    std::mt19937_64 rng(1);
    std::normal_distribution<double> N(0, 1);
    std::uniform_real_distribution<double> U(0, 1);
    int As[] = {2, 3, 4, 5, 8, 30};
    for (int A : As) {
        long bad = 0, exits = 0, total = 0, planar_bad = 0;
        for (int trial = 0; trial < 2000000; ++trial) {
            bool planar = (A >= 4) && (trial % 4 == 0);
            float x[90], y[90];
            double cx[3] = {0,0,0}, cy[3] = {0,0,0};
            double scale = 0.5 + 3 * U(rng);
            bool similar = U(rng) < 0.5;
            for (int a = 0; a < A; ++a) for (int k = 0; k < 3; ++k) {
                double v = N(rng) * scale;
                if (planar && k == 2) v = 0;
                x[3*a+k] = (float)v;
                y[3*a+k] = (float)(similar ? v + 0.05 * N(rng) * (planar && k==2 ? 0 : 1) : N(rng) * scale * ((planar && k==2)?0:1));
            }
            // random rotation of y skipped; centre both
            for (int a = 0; a < A; ++a) for (int k = 0; k < 3; ++k) { cx[k] += x[3*a+k]; cy[k] += y[3*a+k]; }
            double Gx = 0, Gy = 0;
            for (int a = 0; a < A; ++a) for (int k = 0; k < 3; ++k) {
                x[3*a+k] = (float)(x[3*a+k] - cx[k] / A); y[3*a+k] = (float)(y[3*a+k] - cy[k] / A);
                Gx += (double)x[3*a+k] * x[3*a+k]; Gy += (double)y[3*a+k] * y[3*a+k];
            }
            float S[9] = {0};
            for (int a = 0; a < A; ++a) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
                S[3*i+j] = fmaf(x[3*a+i], y[3*a+j], S[3*i+j]);
            float full = ek_rmsd_from_S(S, Gx, Gy, A);
            for (int r = 0; r < 4; ++r) {
                float cur = (float)(full * (0.2 + 1.3 * U(rng))) + (r == 3 ? 1e-4f : 0.f);
                float capped = ek_rmsd_from_S_below(S, Gx, Gy, A, cur);
                ++total;
                if (std::isinf(capped)) {
                    ++exits;
                    if (full < cur) { ++bad; if (planar) ++planar_bad; if (bad < 4) printf("  A=%d full=%.9g cur=%.9g\n", A, full, cur); }
                } else if (capped != full) { ++bad; printf("  A=%d differ %g %g\n", A, capped, full); }
            }
        }
        printf("A=%d: %ld checks, %ld early exits, %ld wrong (%ld planar)\n", A, total, exits, bad, planar_bad);
    }
}
