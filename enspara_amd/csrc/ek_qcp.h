// ek_qcp.h -- device-side QCP (Theobald) minimal RMSD from the 3x3 inner
// product matrix and the two traces.  gfx950 only.
//
// Replaces the arithmetic of mdtraj.rmsd, which enspara binds as metric
// 'rmsd' (reference enspara/cluster/util.py:289-291).  Operation order is a
// contract: tests compare against a CPU checker bit for bit, so this file is
// compiled with -ffp-contract=off and every fused multiply-add is explicit.
#pragma once
#include <hip/hip_runtime.h>

#define EK_EVALPREC 1e-11
#define EK_MAXIT 50

// coefficients of the quartic l^4 + C2 l^2 + C1 l + C0 whose largest root is
// the sum of the (signed) singular values of S; q = sum S_ij^2 = -C2 / 2.
// One copy, shared by the full and the early-stopped solve: the operation
// order below is the one the CPU checker follows.
struct EkQuartic {
    double q, C2, C1, C0;
};

// S is row-major: S[3*i+j] = sum_a x_ai * y_aj  (x = frame, y = center)
__device__ __forceinline__ EkQuartic ek_quartic_from_S(const float (&S)[9])
{
    const double Sxx = S[0], Sxy = S[1], Sxz = S[2];
    const double Syx = S[3], Syy = S[4], Syz = S[5];
    const double Szx = S[6], Szy = S[7], Szz = S[8];
    EkQuartic r;

    // C2 = -2 * sum S_ij^2
    double q = Sxx * Sxx;
    q = q + Sxy * Sxy;
    q = q + Sxz * Sxz;
    q = q + Syx * Syx;
    q = q + Syy * Syy;
    q = q + Syz * Syz;
    q = q + Szx * Szx;
    q = q + Szy * Szy;
    q = q + Szz * Szz;
    r.q = q;
    r.C2 = -2.0 * q;

    // C1 = -8 * det(S)
    const double m0 = Syy * Szz - Syz * Szy;
    const double m1 = Syx * Szz - Syz * Szx;
    const double m2 = Syx * Szy - Syy * Szx;
    const double detS = (Sxx * m0 - Sxy * m1) + Sxz * m2;
    r.C1 = -8.0 * detS;

    // C0 = det(K), K = symmetric traceless 4x4 key matrix of S
    const double k00 = (Sxx + Syy) + Szz;
    const double k01 = Syz - Szy;
    const double k02 = Szx - Sxz;
    const double k03 = Sxy - Syx;
    const double k11 = (Sxx - Syy) - Szz;
    const double k12 = Sxy + Syx;
    const double k13 = Szx + Sxz;
    const double k22 = (Syy - Sxx) - Szz;
    const double k23 = Syz + Szy;
    const double k33 = (Szz - Sxx) - Syy;

    const double s0 = k00 * k11 - k01 * k01;
    const double s1 = k00 * k12 - k01 * k02;
    const double s2 = k00 * k13 - k01 * k03;
    const double s3 = k01 * k12 - k11 * k02;
    const double s4 = k01 * k13 - k11 * k03;
    const double s5 = k02 * k13 - k12 * k03;
    const double c5 = k22 * k33 - k23 * k23;
    const double c4 = k12 * k33 - k13 * k23;
    const double c3 = k12 * k23 - k13 * k22;
    const double c2 = k02 * k33 - k03 * k23;
    const double c1 = k02 * k23 - k03 * k22;
    const double c0 = k02 * k13 - k03 * k12;
    double C0 = s0 * c5 - s1 * c4;
    C0 = C0 + s2 * c3;
    C0 = C0 + s3 * c2;
    C0 = C0 - s4 * c1;
    C0 = C0 + s5 * c0;
    r.C0 = C0;
    return r;
}

__device__ __forceinline__ double ek_msd_from_S(const float (&S)[9], double Gx,
                                                double Gy, int n_atoms)
{
    const EkQuartic p = ek_quartic_from_S(S);
    const double C2 = p.C2, C1 = p.C1, C0 = p.C0;

    // largest root of l^4 + C2 l^2 + C1 l + C0 by Newton from the upper bound
    const double Gsum = Gx + Gy;
    double lam = 0.5 * Gsum;
    for (int it = 0; it < EK_MAXIT; ++it) {
        const double x2 = lam * lam;
        const double b = (x2 + C2) * lam;
        const double a = b + C1;
        const double num = __builtin_fma(a, lam, C0);
        const double den = __builtin_fma(2.0 * x2, lam, b + a);
        if (den == 0.0)
            break;
        const double delta = num / den;
        lam = lam - delta;
        if (__builtin_fabs(delta) < __builtin_fabs(EK_EVALPREC * lam))
            break;
    }
    double msd = (Gsum - 2.0 * lam) / (double)n_atoms;
    if (!(msd > 0.0))
        msd = 0.0;
    return msd;
}

__device__ __forceinline__ float ek_rmsd_from_S(const float (&S)[9], double Gx,
                                                double Gy, int n_atoms)
{
    return __builtin_sqrtf((float)ek_msd_from_S(S, Gx, Gy, n_atoms));
}

// The same solve, abandoned as soon as its result is known to be >= `cur`.
//
// Beyond its largest root the quartic and its first two derivatives are
// positive, so the Newton iterates fall monotonically from the upper bound
// (Gx+Gy)/2 towards that root and (Gsum - 2*lam)/n_atoms, evaluated at any
// iterate, is a lower bound of the final msd; so is the value at sqrt(3 q),
// q = sum S_ij^2, because the largest root is s1+s2+s3 <= sqrt(3 q) (singular
// values of S).  Once such a bound exceeds cur^2 by a margin far above the
// rounding involved (1e-4 relative: the float32 roundings of the result err by
// ~1e-7; 1e-9 of Gsum absolute: see below) the distance cannot be below `cur`,
// which is all a strict "<" update (kcenters.py:304) asks, and +inf is returned.
// Otherwise the iteration is the one above, operation for operation, and the
// result has the same bits.
//
// That reasoning is about the exact roots; the reference iteration follows it
// only while the largest root is well separated.  Near a multiple root Newton's
// steps wander in rounding noise and now and then jump (measured: a largest
// root doubled to 1e-7 relative sends ~1e-6 of the solves far off), and then
// only the full iteration reproduces what the reference returns.  The roots are
// s1+s2+s3, s1-s2-s3, -s1+s2-s3, -s1-s2+s3 (s3 signed like det S) and the
// discriminant of the quartic is 4096 [(s2+s3)(s1+s3)(s1+s2)(s1-s2)(s1-s3)(s2-s3)]^2;
// abandoning is allowed only when it exceeds 1e-6 C2^6 = 6.4e-5 (s1^2+s2^2+s3^2)^6,
// which keeps every pair of roots at least ~1e-4 s1 apart (a rank-one S, two-atom
// or collinear structures, has discriminant 0).  The largest root is then found
// to ~1e-11 of its size, far inside the absolute margin.  cur = +inf or NaN
// never abandons (and skips the test).
__device__ __forceinline__ float ek_rmsd_from_S_below(const float (&S)[9],
                                                      double Gx, double Gy,
                                                      int n_atoms, float cur)
{
    // nothing can be abandoned against +inf (a frame's first distance) or a
    // NaN (never produced here, but then nothing is known): the plain solve,
    // without the separation test
    if (!(cur < __builtin_inff()))
        return ek_rmsd_from_S(S, Gx, Gy, n_atoms);
    const EkQuartic p = ek_quartic_from_S(S);
    const double q = p.q, C2 = p.C2, C1 = p.C1, C0 = p.C0;

    const double Gsum = Gx + Gy;
    // discriminant of l^4 + p l^2 + r1 l + r0 (only its size matters here)
    const double p2 = C2 * C2, r12 = C1 * C1, r02 = C0 * C0;
    const double disc = 16.0 * p2 * p2 * C0 - 4.0 * p2 * C2 * r12 -
                        128.0 * p2 * r02 + 144.0 * C2 * r12 * C0 -
                        27.0 * r12 * r12 + 256.0 * r02 * C0;
    const bool separated = disc > 1e-6 * (p2 * p2 * p2);
    const double far =
        separated ? (double)n_atoms * (((double)cur * (double)cur) * 1.0001) +
                        1e-9 * Gsum
                  : __builtin_inf();
    {
        const double t = Gsum - far;            // 2 sqrt(3 q) < t ?
        if (t > 0.0 && t * t > 12.000001 * q)
            return __builtin_inff();
    }
    double lam = 0.5 * Gsum;
    for (int it = 0; it < EK_MAXIT; ++it) {
        const double x2 = lam * lam;
        const double b = (x2 + C2) * lam;
        const double a = b + C1;
        const double num = __builtin_fma(a, lam, C0);
        const double den = __builtin_fma(2.0 * x2, lam, b + a);
        if (den == 0.0)
            break;
        const double delta = num / den;
        lam = lam - delta;
        if (__builtin_fabs(delta) < __builtin_fabs(EK_EVALPREC * lam))
            break;
        if (Gsum - 2.0 * lam > far)
            return __builtin_inff();
    }
    double msd = (Gsum - 2.0 * lam) / (double)n_atoms;
    if (!(msd > 0.0))
        msd = 0.0;
    return __builtin_sqrtf((float)msd);
}
