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

// W pairs at once, statement by statement: the chains below are long and
// dependent (each step waits out the latency of the one before), W = 2 of them
// interleaved in program order fill the gaps.  Straight-line code, no early
// exit: a wave does not branch on its slowest lane.
#define EK_W_ for (int u = 0; u < W; ++u)
// (after every statement its W results pass through one empty asm statement:
// both are computed by then, and the next statement starts from both -- the
// compiler otherwise sinks each pair's chain to where its result is used, one
// chain after the other)
#if defined(__HIP_DEVICE_COMPILE__)
template <int W> __device__ __forceinline__ void ek_tie(float (&v)[W])
{
    if constexpr (W == 2)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]));
    else
        asm volatile("" : "+v"(v[0]));
}
#define EK_TIE_(V) ek_tie<W>(V);
#else
#define EK_TIE_(V)
#endif
template <int W>
__device__ __forceinline__ void ek_far_certified_f32_w(const float (&S)[W][9],
                                                       const float (&Gsum)[W], int n_atoms,
                                                       const float (&cur)[W], bool (&far)[W])
{
    float q[W], rs[W], N[W][9], bn[W], dn[W], dn2[W];
    bool ok[W];
#pragma unroll
    EK_W_ q[u] = S[u][0] * S[u][0];
    EK_TIE_(q)
#pragma unroll
    for (int j = 1; j < 9; ++j)
#pragma unroll
        EK_W_ q[u] = __builtin_fmaf(S[u][j], S[u][j], q[u]);
    EK_TIE_(q)
#pragma unroll
    EK_W_ ok[u] = q[u] > 1e-30f && q[u] < 1e30f;    // (false for NaN too)
    // N = S / sqrt(q): |N|_F^2 = 1 within 1.5e-6, nothing below under- or overflows
#pragma unroll
    EK_W_ rs[u] = EK_SQRTF(EK_RCPF(q[u]));
    EK_TIE_(rs)
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        EK_W_ N[u][j] = S[u][j] * rs[u];
    // cofactors, b = |cof N|_F^2 <= 1/3, d = det N; absolute errors below 1e-6
    float c[W][9];
#define EK_COF_(K, A0, A1, B0, B1)                                             \
    _Pragma("unroll") EK_W_ c[u][K] =                                         \
        __builtin_fmaf(N[u][A0], N[u][A1], -(N[u][B0] * N[u][B1]));
    EK_COF_(0, 4, 8, 5, 7)
    EK_COF_(1, 5, 6, 3, 8)
    EK_COF_(2, 3, 7, 4, 6)
    EK_COF_(3, 2, 7, 1, 8)
    EK_COF_(4, 0, 8, 2, 6)
    EK_COF_(5, 1, 6, 0, 7)
    EK_COF_(6, 1, 5, 2, 4)
    EK_COF_(7, 2, 3, 0, 5)
    EK_COF_(8, 0, 4, 1, 3)
#undef EK_COF_
#pragma unroll
    EK_W_ bn[u] = c[u][0] * c[u][0];
    EK_TIE_(bn)
#pragma unroll
    for (int j = 1; j < 9; ++j)
#pragma unroll
        EK_W_ bn[u] = __builtin_fmaf(c[u][j], c[u][j], bn[u]);
    EK_TIE_(bn)
#pragma unroll
    EK_W_ dn[u] = N[u][0] * c[u][0];
    EK_TIE_(dn)
#pragma unroll
    EK_W_ dn[u] = __builtin_fmaf(N[u][1], c[u][1], dn[u]);
    EK_TIE_(dn)
#pragma unroll
    EK_W_ dn[u] = __builtin_fmaf(N[u][2], c[u][2], dn[u]);
    EK_TIE_(dn)
#pragma unroll
    EK_W_ dn2[u] = dn[u] * dn[u];
    EK_TIE_(dn2)
    // c(x) = x^3 - x^2 + bn x - dn^2, x = s1^2 / q in [1/3, 1]; evaluated with an
    // absolute error below 8e-6 (the unit coefficient stands for 1 +- 1.5e-6)
    // x+ from above (guard) and from below (a lower bound of x that needs no check)
    float disc[W], xg[W], xpl[W], x[W];
#pragma unroll
    EK_W_ disc[u] = __builtin_fmaf(-3.0f, bn[u], 1.0f);
    EK_TIE_(disc)
#pragma unroll
    EK_W_ xg[u] = (1.0f + EK_SQRTF(__builtin_fmaxf(disc[u] + 1e-5f, 0.0f))) *
                  (0.33333334f * 1.000003f);
    EK_TIE_(xg)
#pragma unroll
    EK_W_ xpl[u] = (1.0f + EK_SQRTF(__builtin_fmaxf(disc[u] - 1e-5f, 0.0f))) *
                   (0.33333331f * 0.999997f);
    EK_TIE_(xpl)
#pragma unroll
    EK_W_ x[u] = 1.0f - bn[u];
    EK_TIE_(x)
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        float cv[W], cp[W];
#pragma unroll
        EK_W_ cv[u] = __builtin_fmaf(__builtin_fmaf(x[u] - 1.0f, x[u], bn[u]), x[u], -dn2[u]);
    EK_TIE_(cv)
#pragma unroll
        EK_W_ cp[u] = __builtin_fmaf(__builtin_fmaf(3.0f, x[u], -2.0f), x[u], bn[u]);
    EK_TIE_(cp)
#pragma unroll
        EK_W_ x[u] = __builtin_fmaf(-cv[u], EK_RCPF(cp[u]), x[u]);
    EK_TIE_(x)
    }
    float x_up[W], c_up[W], x_try[W], c_lo[W], x_lo[W];
#pragma unroll
    EK_W_ x_up[u] = __builtin_fmaxf(x[u] * 1.001f, xg[u]);
    EK_TIE_(x_up)
#pragma unroll
    EK_W_ c_up[u] = __builtin_fmaf(__builtin_fmaf(x_up[u] - 1.0f, x_up[u], bn[u]), x_up[u],
                                   -dn2[u]);
    EK_TIE_(c_up)
#pragma unroll
    EK_W_ ok[u] = ok[u] && c_up[u] > 8e-6f;         // (false for NaN too)
#pragma unroll
    EK_W_ x_try[u] = x[u] * 0.999f;
    EK_TIE_(x_try)
#pragma unroll
    EK_W_ c_lo[u] = __builtin_fmaf(__builtin_fmaf(x_try[u] - 1.0f, x_try[u], bn[u]),
                                   x_try[u], -dn2[u]);
    EK_TIE_(c_lo)
#pragma unroll
    EK_W_ x_lo[u] = (x_try[u] > xg[u] && c_lo[u] < -8e-6f) ? x_try[u] : xpl[u];
    EK_TIE_(x_lo)
    // bounds of (s2^2 + s3^2) / q and of 2 |d| / (s1 q)
    float dl[W], du[W], rx_lo[W], rx_up[W], y_lo[W], y_up[W], rs_lo[W], rs_up[W];
    float g_lo[W], g_up[W];
#pragma unroll
    EK_W_ dl[u] = __builtin_fmaxf(__builtin_fabsf(dn[u]) - 1e-6f, 0.0f);
    EK_TIE_(dl)
#pragma unroll
    EK_W_ du[u] = __builtin_fabsf(dn[u]) + 1e-6f;
    EK_TIE_(du)
#pragma unroll
    EK_W_ rx_lo[u] = EK_RCPF(x_lo[u]) * 1.000001f;      // >= 1 / x_lo
    EK_TIE_(rx_lo)
#pragma unroll
    EK_W_ rx_up[u] = EK_RCPF(x_up[u]) * 0.999999f;      // <= 1 / x_up
    EK_TIE_(rx_up)
#pragma unroll
    EK_W_ y_lo[u] = ((bn[u] - 1e-6f) - du[u] * du[u] * rx_lo[u]) * rx_up[u];
    EK_TIE_(y_lo)
#pragma unroll
    EK_W_ y_up[u] = ((bn[u] + 1e-6f) - dl[u] * dl[u] * rx_up[u]) * rx_lo[u];
    EK_TIE_(y_up)
#pragma unroll
    EK_W_ rs_lo[u] = EK_SQRTF(rx_lo[u]) * 1.000001f;    // >= sqrt(q) / s1
    EK_TIE_(rs_lo)
#pragma unroll
    EK_W_ rs_up[u] = EK_SQRTF(rx_up[u]) * 0.999999f;    // <= sqrt(q) / s1
    EK_TIE_(rs_up)
#pragma unroll
    EK_W_ g_lo[u] = dn[u] > 0.0f ? __builtin_fmaf(2.0f * dl[u], rs_up[u], y_lo[u])
                                 : __builtin_fmaf(-2.0f * du[u], rs_lo[u], y_lo[u]);
    EK_TIE_(g_lo)
#pragma unroll
    EK_W_ g_up[u] = dn[u] > 0.0f ? __builtin_fmaf(2.0f * du[u], rs_lo[u], y_up[u])
                                 : __builtin_fmaf(-2.0f * dl[u], rs_up[u], y_up[u]);
    EK_TIE_(g_up)
#pragma unroll
    EK_W_ ok[u] = ok[u] && g_lo[u] >= 2e-4f;    // else: the largest root is not certainly separated
    // lambda_max <= sqrt(q) (s1_up + sqrt(G2_up)) / sqrt(q_n)
    float U[W];
#pragma unroll
    EK_W_ U[u] = (EK_SQRTF(x_up[u]) + EK_SQRTF(__builtin_fmaxf(g_up[u], 0.0f))) * 1.000003f;
    EK_TIE_(U)
#pragma unroll
    EK_W_ U[u] = U[u] * EK_SQRTF(q[u]) * 1.000002f;
    EK_TIE_(U)
#pragma unroll
    EK_W_ far[u] = ok[u] && __builtin_fmaf(-2.0f, U[u], Gsum[u]) >
                                __builtin_fmaf((float)n_atoms * (cur[u] * cur[u]), 1.0001f,
                                               5e-6f * Gsum[u]);
}
#undef EK_W_
#undef EK_TIE_

__device__ __forceinline__ bool ek_far_certified_f32(const float (&S)[9], float Gsum,
                                                     int n_atoms, float cur)
{
    float S1[1][9], G1[1] = {Gsum}, c1[1] = {cur};
    bool f1[1];
#pragma unroll
    for (int j = 0; j < 9; ++j)
        S1[0][j] = S[j];
    ek_far_certified_f32_w<1>(S1, G1, n_atoms, c1, f1);
    return f1[0];
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
