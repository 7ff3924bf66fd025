// ek_qcp.h -- device-side QCP (Theobald) minimal RMSD from the 3x3 inner
// product matrix and the two traces.  gfx950 only.
//
// Replaces the arithmetic of mdtraj.rmsd, which enspara binds as metric
// 'rmsd' (reference enspara/cluster/util.py:289-291).  Operation order is a
// contract: tests compare against a CPU checker bit for bit, so this file is
// compiled with -ffp-contract=off and every fused multiply-add is explicit.
#pragma once
#include <hip/hip_runtime.h>

// float32 reciprocal / square root where only BOUNDS are computed (1 ulp on the
// device, correctly rounded on the host: callers pad every use)
#if defined(__HIP_DEVICE_COMPILE__)
#define EK_RCPF(x) __builtin_amdgcn_rcpf(x)
#define EK_SQRTF(x) __builtin_amdgcn_sqrtf(x)
#else
#define EK_RCPF(x) (1.0f / (x))
#define EK_SQRTF(x) __builtin_sqrtf(x)
#endif

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

// ---- a float32 certificate for FAR pairs, before any float64 coefficient --------------
//
// Most distances a pass computes are never used: the pair is far, all a strict
// "<" needs to know.  The early-stopped solve below still pays the float64
// coefficients, the discriminant and two or three Newton steps with their
// float64 divisions for such a pair (~240 vector instructions, most of them
// half rate).  This test answers the same question from float32 arithmetic on
// S alone (~140 full-rate instructions, no division), for about 99 % of the far
// pairs of the bench's data; where it cannot, the solve below runs as before,
// so results never depend on it.
//
// With singular values s1 >= s2 >= |s3| of S, s3 signed like det S:
//   q = sum S_ij^2 = s1^2 + s2^2 + s3^2,  b = |cof S|_F^2 = sum (si sj)^2,
//   d = det S = s1 s2 s3;  x = s1^2 is the largest root of
//   c(x) = x^3 - q x^2 + b x - d^2,  s2^2 + s3^2 = (b - d^2 / x) / x,
//   (s2 + s3)^2 = s2^2 + s3^2 + 2 d / s1 =: G2,  lambda_max = s1 + sqrt(G2),
// and the quartic's largest root is apart from the next one by 2 (s2 + s3).
// (1) x is bracketed, x_lo <= s1^2 <= x_up: five Newton steps on c from
//     q - b / q >= s1^2, then SIGN checks of c at (1 +- 1e-3) times the
//     iterate with the float32 evaluation error as margin -- c increases beyond
//     its larger critical point x+ = (q + sqrt(q^2 - 3 b)) / 3 <= s1^2, so
//     x > x+ with c(x) > 0 is above s1^2 and x > x+ with c(x) < 0 below it;
//     x+ itself is a lower bound that needs no check;
// (2) SEPARATION: G2 >= 2e-4 q from the bracket, i.e. the two largest roots at
//     least ~0.03 sqrt(q) apart -- a hundred times what the discriminant test
//     below asks for: the reference iteration then converges (from above,
//     monotonically, by at least a quarter of its distance per step) to
//     lambda_max within 1e-6 (Gx + Gy) in its fifty steps;
// (3) FAR: Gx + Gy - 2 U > n_atoms cur^2 (1 + 1e-4) + 4e-6 (Gx + Gy) with
//     U >= lambda_max from the bracket.
// Every rounding of the float32 evaluation is covered by an explicit slack
// (1e-6 relative on b, d and the roots/reciprocals, 4e-6 q^3 on c); an
// overflow, a NaN or q = 0 fails the comparisons and certifies nothing.
// Quantities are normalised by q so that the slacks are plain numbers.
// Measured (round 4, profiles/r04/README.md): sound, and NOT faster.  float64
// multiply-adds issue at the float32 rate on this chip, so the test costs
// ~0.6-0.8 of the path it replaces, and a wave pays the old path as well as soon
// as ONE of its 64 pairs is not certified (0.989^64: half the waves).  Off by
// default; tests/test_qcp_host.py keeps it honest.
#ifndef EK_FAR_F32
#define EK_FAR_F32 0
#endif

__device__ __forceinline__ bool ek_far_certified_f32(const float (&S)[9], float Gsum,
                                                     int n_atoms, float cur)
{
    float q = S[0] * S[0];
#pragma unroll
    for (int j = 1; j < 9; ++j)
        q = __builtin_fmaf(S[j], S[j], q);
    if (!(q > 1e-30f && q < 1e30f))     // (also NaN)
        return false;
    // N = S / sqrt(q): |N|_F^2 = 1 within 1.5e-6, nothing below under- or overflows
    const float rs = EK_SQRTF(EK_RCPF(q));
    float N[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
        N[j] = S[j] * rs;
    // cofactors, b = |cof N|_F^2 <= 1/3, d = det N; absolute errors below 1e-6
    const float c00 = __builtin_fmaf(N[4], N[8], -(N[5] * N[7]));
    const float c01 = __builtin_fmaf(N[5], N[6], -(N[3] * N[8]));
    const float c02 = __builtin_fmaf(N[3], N[7], -(N[4] * N[6]));
    const float c10 = __builtin_fmaf(N[2], N[7], -(N[1] * N[8]));
    const float c11 = __builtin_fmaf(N[0], N[8], -(N[2] * N[6]));
    const float c12 = __builtin_fmaf(N[1], N[6], -(N[0] * N[7]));
    const float c20 = __builtin_fmaf(N[1], N[5], -(N[2] * N[4]));
    const float c21 = __builtin_fmaf(N[2], N[3], -(N[0] * N[5]));
    const float c22 = __builtin_fmaf(N[0], N[4], -(N[1] * N[3]));
    float bn = c00 * c00;
    bn = __builtin_fmaf(c01, c01, bn);
    bn = __builtin_fmaf(c02, c02, bn);
    bn = __builtin_fmaf(c10, c10, bn);
    bn = __builtin_fmaf(c11, c11, bn);
    bn = __builtin_fmaf(c12, c12, bn);
    bn = __builtin_fmaf(c20, c20, bn);
    bn = __builtin_fmaf(c21, c21, bn);
    bn = __builtin_fmaf(c22, c22, bn);
    float dn = N[0] * c00;
    dn = __builtin_fmaf(N[1], c01, dn);
    dn = __builtin_fmaf(N[2], c02, dn);
    const float dn2 = dn * dn;
    // c(x) = x^3 - x^2 + bn x - dn^2, x = s1^2 / q in [1/3, 1]; evaluated with an
    // absolute error below 8e-6 (the unit coefficient stands for 1 +- 1.5e-6)
    // x+ from above (guard) and from below (a lower bound of x that needs no check)
    const float disc = __builtin_fmaf(-3.0f, bn, 1.0f);
    const float xg = (1.0f + EK_SQRTF(__builtin_fmaxf(disc + 1e-5f, 0.0f))) *
                     (0.33333334f * 1.000003f);
    const float xpl = (1.0f + EK_SQRTF(__builtin_fmaxf(disc - 1e-5f, 0.0f))) *
                      (0.33333331f * 0.999997f);
    float x = 1.0f - bn;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const float cv = __builtin_fmaf(__builtin_fmaf(x - 1.0f, x, bn), x, -dn2);
        const float cp = __builtin_fmaf(__builtin_fmaf(3.0f, x, -2.0f), x, bn);
        x = __builtin_fmaf(-cv, EK_RCPF(cp), x);
    }
    const float x_up = __builtin_fmaxf(x * 1.001f, xg);
    const float c_up = __builtin_fmaf(__builtin_fmaf(x_up - 1.0f, x_up, bn), x_up, -dn2);
    if (!(c_up > 8e-6f))            // (also NaN)
        return false;
    const float x_try = x * 0.999f;
    const float c_lo = __builtin_fmaf(__builtin_fmaf(x_try - 1.0f, x_try, bn), x_try, -dn2);
    const float x_lo = (x_try > xg && c_lo < -8e-6f) ? x_try : xpl;
    // bounds of (s2^2 + s3^2) / q and of 2 |d| / (s1 q)
    const float bl = bn - 1e-6f, bu = bn + 1e-6f;
    const float da = __builtin_fabsf(dn);
    const float dl = __builtin_fmaxf(da - 1e-6f, 0.0f), du = da + 1e-6f;
    const float rx_lo = EK_RCPF(x_lo) * 1.000001f;     // >= 1 / x_lo
    const float rx_up = EK_RCPF(x_up) * 0.999999f;     // <= 1 / x_up
    const float y_lo = (bl - du * du * rx_lo) * rx_up;
    const float y_up = (bu - dl * dl * rx_up) * rx_lo;
    const float rs_lo = EK_SQRTF(rx_lo) * 1.000001f;   // >= sqrt(q) / s1
    const float rs_up = EK_SQRTF(rx_up) * 0.999999f;   // <= sqrt(q) / s1
    float g_lo, g_up;
    if (dn > 0.0f) {
        g_lo = __builtin_fmaf(2.0f * dl, rs_up, y_lo);
        g_up = __builtin_fmaf(2.0f * du, rs_lo, y_up);
    } else {
        g_lo = __builtin_fmaf(-2.0f * du, rs_lo, y_lo);
        g_up = __builtin_fmaf(-2.0f * dl, rs_up, y_up);
    }
    if (!(g_lo >= 2e-4f))           // the largest root is not certainly separated
        return false;
    // lambda_max <= sqrt(q) (s1_up + sqrt(G2_up)) / sqrt(q_n)
    const float u = (EK_SQRTF(x_up) + EK_SQRTF(__builtin_fmaxf(g_up, 0.0f))) * 1.000003f;
    const float U = u * EK_SQRTF(q) * 1.000002f;
    const float thr = __builtin_fmaf((float)n_atoms * (cur * cur), 1.0001f, 5e-6f * Gsum);
    return __builtin_fmaf(-2.0f, U, Gsum) > thr;
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
#if EK_FAR_F32
    if (ek_far_certified_f32(S, (float)(Gx + Gy), n_atoms, cur))
        return __builtin_inff();
#endif
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
        // A closer bound from what is already here (round 4).  With the singular
        // values s1 >= s2 >= |s3| of S (s3 signed like det S) the largest root is
        // L = s1 + s2 + s3, L^2 = q + 2 e2 with e2 = s1 s2 + s1 s3 + s2 s3 and
        // e2^2 = b + 2 det(S) L, where b = |cof S|_F^2 = sum (si sj)^2 -- and
        // C0 = q^2 - 4 b, so b costs one multiply-add.  Hence, with L <= sqrt(3 q),
        //   L <= sqrt(q + 2 sqrt(b + 2 max(det S, 0) sqrt(3 q))) =: U.
        // On the bench's data sqrt(3 q) certifies 27-49 % of the far pairs, U
        // 99.8 %: the two or three Newton steps (each with a float64 division) a
        // far pair took to pass its iterate bound are skipped.  Square roots in
        // float32, every rounding covered by a factor; an overflow gives inf or
        // NaN and fails the comparison.
        if (t > 0.0) {
            const float qf = (float)q * 1.000001f;
            const float bf = __builtin_fmaxf((float)(0.25 * __builtin_fma(q, q, -C0) +
                                                     1e-12 * (q * q)), 0.0f) * 1.000001f;
            const float df = __builtin_fmaxf((float)(-0.125 * C1), 0.0f) * 1.000001f;
            const float e2 = EK_SQRTF(__builtin_fmaf(2.0f * df,
                                                     EK_SQRTF(3.0f * qf) * 1.000001f,
                                                     bf)) * 1.000001f;
            const float U = EK_SQRTF(__builtin_fmaf(2.0f, e2, qf)) * 1.000002f;
            if (t > 2.0 * (double)U)
                return __builtin_inff();
        }
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
