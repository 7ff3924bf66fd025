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
#define EK_RSQF(x) __builtin_amdgcn_rsqf(x)
#else
#define EK_RCPF(x) (1.0f / (x))
#define EK_SQRTF(x) __builtin_sqrtf(x)
#define EK_RSQF(x) (1.0f / __builtin_sqrtf(x))
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
// coefficients and the discriminant for such a pair (~270 vector instructions).
// This test answers the same question from float32 arithmetic on S alone, in
// closed form: ~80 instructions, five of them quarter rate, no iteration, no
// branch.  Where it cannot decide, the solve below runs as before, so results
// never depend on it.
//
// With singular values s1 >= s2 >= |s3| of S, s3 signed like det S, and
// everything divided by the matching power of q = sum S_ij^2 = s1^2 + s2^2 + s3^2:
//   x = s1^2, y = s2^2 + s3^2 = 1 - x, p = s2^2 s3^2 <= y^2 / 4,
//   b = |cof S|_F^2 = x y + p,  d = det S,  sqrt(p) = |d| / sqrt(x),
//   (s2 + s3)^2 = y + 2 d / sqrt(x) =: G2,  lambda_max = s1 + s2 + s3,
// and the quartic's largest root is apart from the next one by 2 (s2 + s3).
// (1) x is BRACKETED by b alone.  p >= 0 gives x^2 - x + b >= 0, and x >= 1/3
//     is then on the upper branch whenever b < 2/9:  x >= (1 + sqrt(1 - 4 b)) / 2
//     (exact for rank two); p <= y^2 / 4 gives 3 x^2 - 2 x + 4 b - 1 <= 0:
//     x <= (1 + 2 sqrt(1 - 3 b)) / 3 (exact for s2^2 = s3^2).
// (2) SEPARATION: G2 >= (1 - x_up) - 2 max(-d, 0) / sqrt(x_lo) >= 2e-4, i.e. the
//     two largest roots at least ~0.03 sqrt(q) apart -- a hundred times what the
//     discriminant test below asks for: the reference iteration then converges
//     (from above, monotonically, by at least a quarter of its distance per
//     step) to lambda_max within 1e-6 (Gx + Gy) in its fifty steps.
// (3) FAR: lambda_max^2 = q + 2 e2, e2 = s1 s2 + s1 s3 + s2 s3, e2^2 = b + 2 d
//     lambda_max <= b + 2 max(d, 0) sqrt(3 q): with U the bound this gives,
//     Gx + Gy - 2 U > n_atoms cur^2 (1 + 1e-4) + 5e-6 (Gx + Gy), asked as
//     t > 0 and t^2 > 4 U^2 for t = (Gx + Gy)(1 - 6e-6) - n_atoms cur^2 1.00011
//     (the caller's: ek_far_t; float32 roundings of t are inside the 1e-6).
// Every rounding of the float32 evaluation is covered by an explicit slack
// (3e-6 absolute on b / q^2, 2e-6 on d / q^1.5 -- the cofactors are differences
// of products below q / 2 --, 2e-6 relative on roots and reciprocals); q outside
// [1e-12, 1e12] (where the raw cofactors could leave the float32 range), an
// overflow or a NaN fails a comparison and certifies nothing.
// Yield on the bench's data: 98.3 % of the far pairs (the iterated bracket of
// round 4's first version: 98.9 %, at ~190 instructions).  Alone it would still
// not pay -- a wave runs the float64 path as soon as ONE of its 64 pairs is not
// certified --: ek_pass16_kernel compacts the uncertified pairs into a queue.
// tests/test_qcp_host.py keeps it honest.
#ifndef EK_FAR_F32
#define EK_FAR_F32 0
#endif

// the caller's side of (3): t for a pair from the two traces and the frame's
// current distance (ek_pass16_kernel splits it into a per-frame and a
// per-candidate part, computed once each)
__device__ __forceinline__ float ek_far_t_frame(float G, int n_atoms, float cur)
{
    return __builtin_fmaf(G, 0.999994f, -(((float)n_atoms * (cur * cur)) * 1.00011f));
}
__device__ __forceinline__ float ek_far_t_center(float G) { return G * 0.999994f; }

// W pairs at once, statement by statement (straight-line code, no early exit:
// a wave does not branch on its slowest lane)
#define EK_W_ for (int u = 0; u < W; ++u)
template <int W>
__device__ __forceinline__ void ek_far_certified_f32_w(const float (&S)[W][9],
                                                       const float (&t)[W], bool (&far)[W])
{
    float q[W], c[W][9], b[W], d[W];
#pragma unroll
    EK_W_ q[u] = S[u][0] * S[u][0];
#pragma unroll
    for (int j = 1; j < 9; ++j)
#pragma unroll
        EK_W_ q[u] = __builtin_fmaf(S[u][j], S[u][j], q[u]);
    // cofactors, b = |cof S|_F^2 <= q^2 / 3, d = det S
#define EK_COF_(K, A0, A1, B0, B1)                                             \
    _Pragma("unroll") EK_W_ c[u][K] =                                         \
        __builtin_fmaf(S[u][A0], S[u][A1], -(S[u][B0] * S[u][B1]));
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
    EK_W_ b[u] = c[u][0] * c[u][0];
#pragma unroll
    for (int j = 1; j < 9; ++j)
#pragma unroll
        EK_W_ b[u] = __builtin_fmaf(c[u][j], c[u][j], b[u]);
#pragma unroll
    EK_W_ d[u] = __builtin_fmaf(
        S[u][2], c[u][2], __builtin_fmaf(S[u][1], c[u][1], S[u][0] * c[u][0]));
#pragma unroll
    EK_W_
    {
        const float s = EK_RSQF(q[u]), rq = s * s;
        const float bn = (b[u] * rq) * rq;          // b / q^2 within 3e-6
        const float dn = (d[u] * rq) * s;           // d / q^1.5 within 2e-6
        // (1) the bracket of x = s1^2 / q
        const float h3 = EK_SQRTF(__builtin_fmaxf(__builtin_fmaf(-3.0f, bn, 1.00001f), 0.0f));
        const float x_up = __builtin_fmaf(h3, 0.666668f, 0.333334f);
        const float h4 = EK_SQRTF(__builtin_fmaxf(__builtin_fmaf(-4.0f, bn, 0.99998f), 0.0f));
        const float x_lo = bn < 0.22f ? __builtin_fmaf(h4, 0.499999f, 0.499999f) : 0.333333f;
        // (2) (s2 + s3)^2 / q from below
        const float nd = __builtin_fmaxf(2e-6f - dn, 0.0f);
        const float g_lo = __builtin_fmaf(-2.000006f * nd, EK_RSQF(x_lo), 1.0f - x_up);
        // (3) (2 lambda_max)^2 from above
        const float pd = __builtin_fmaxf(dn + 2e-6f, 0.0f);
        const float e2 = EK_SQRTF(__builtin_fmaf(pd, 3.464109f, bn + 3e-6f)) * 1.000001f;
        const float U2 = q[u] * __builtin_fmaf(e2, 8.00004f, 4.00002f);
        // (false for NaN too; "&", not "&&": no branch, no lane waits for another)
        far[u] = (q[u] > 1e-12f) & (q[u] < 1e12f) & (g_lo >= 2e-4f) & (t[u] > 0.0f) &
                 (t[u] * t[u] > U2);
    }
}
#undef EK_W_

// ---- second level (round 5): the same verdict from a TIGHTER upper bound -----------------
// The pairs the closed form above does not settle are, almost all of them, still
// far: its bound U of the largest root is a crude one (it replaces lambda_max by
// sqrt(3 q) inside a square root).  Dividing the quartic by q^2,
//   p(x) = x^4 - 2 x^2 - 8 dn x + 1 - 4 bn,   x = lambda / sqrt(q),
// is convex and increasing beyond its largest root x_max >= 1 (p'' = 12 x^2 - 4), so
// a Newton step from any u >= x_max lands in [x_max, u]: two steps from U / sqrt(q)
// bring the bound within ~1e-4 of the root.  For soundness a step must not be
// LONGER than the true one: p(u) is taken from below (dn, bn at the upper ends of
// their error intervals -- the slacks of the certificate above --, 1e-5 for the
// roundings of terms below 10) and p'(u) from above, and the quotient is shortened
// by 1e-5 more.  The separation test and the range of q are the certificate's, so
// the reference iteration is known to end at lambda_max here too.  ~110
// instructions.  Measured in ek_pass16_kernel (-DEK_P16_LEVEL2=1: densely over a
// wave's queue, the float64 path only if a lane is still undecided) and NOT
// faster -- 0.7906 against 0.7888 ms per pass --, so no kernel calls it by
// default; it stays with its soundness tests (tests/test_qcp_host.py,
// tests/test_gpu_qcp_device.py through ek_qcp_probe: the same adversarial
// families as the first level) for the form of the pass that could use it.
__device__ __forceinline__ bool ek_far_certified2_f32(const float (&S)[9], float t)
{
    float q = S[0] * S[0];
#pragma unroll
    for (int j = 1; j < 9; ++j)
        q = __builtin_fmaf(S[j], S[j], q);
    float c[9];
#define EK_COF2_(K, A0, A1, B0, B1) c[K] = __builtin_fmaf(S[A0], S[A1], -(S[B0] * S[B1]));
    EK_COF2_(0, 4, 8, 5, 7)
    EK_COF2_(1, 5, 6, 3, 8)
    EK_COF2_(2, 3, 7, 4, 6)
    EK_COF2_(3, 2, 7, 1, 8)
    EK_COF2_(4, 0, 8, 2, 6)
    EK_COF2_(5, 1, 6, 0, 7)
    EK_COF2_(6, 1, 5, 2, 4)
    EK_COF2_(7, 2, 3, 0, 5)
    EK_COF2_(8, 0, 4, 1, 3)
#undef EK_COF2_
    float b = c[0] * c[0];
#pragma unroll
    for (int j = 1; j < 9; ++j)
        b = __builtin_fmaf(c[j], c[j], b);
    const float d = __builtin_fmaf(S[2], c[2], __builtin_fmaf(S[1], c[1], S[0] * c[0]));
    const float s = EK_RSQF(q), rq = s * s;
    const float bn = (b * rq) * rq;             // b / q^2 within 3e-6
    const float dn = (d * rq) * s;              // d / q^1.5 within 2e-6
    // the separation of the two largest roots, as in the first level
    const float h3 = EK_SQRTF(__builtin_fmaxf(__builtin_fmaf(-3.0f, bn, 1.00001f), 0.0f));
    const float x_up = __builtin_fmaf(h3, 0.666668f, 0.333334f);
    const float h4 = EK_SQRTF(__builtin_fmaxf(__builtin_fmaf(-4.0f, bn, 0.99998f), 0.0f));
    const float x_lo = bn < 0.22f ? __builtin_fmaf(h4, 0.499999f, 0.499999f) : 0.333333f;
    const float nd = __builtin_fmaxf(2e-6f - dn, 0.0f);
    const float g_lo = __builtin_fmaf(-2.000006f * nd, EK_RSQF(x_lo), 1.0f - x_up);
    // u >= x_max: the closed form's bound, (U / sqrt(q))^2 = 1 + 2 e2
    const float pd = __builtin_fmaxf(dn + 2e-6f, 0.0f);
    const float e2 = EK_SQRTF(__builtin_fmaf(pd, 3.464109f, bn + 3e-6f)) * 1.000001f;
    float u = EK_SQRTF(__builtin_fmaf(e2, 2.00001f, 1.000005f)) * 1.000001f;
    const float dn_hi = dn + 2e-6f, dn_lo = dn - 2e-6f, bn_hi = bn + 3e-6f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const float x2 = u * u;
        // p(u) from below, p'(u) from above
        const float pl = __builtin_fmaf(x2, x2 - 2.0f, 1.0f) - 8.0f * dn_hi * u -
                         4.0f * bn_hi - 1e-5f;
        const float dp = (4.0f * u * (x2 - 1.0f) - 8.0f * dn_lo) * 1.00001f + 1e-5f;
        const float step = (dp > 0.0f && pl > 0.0f) ? pl * EK_RCPF(dp) * 0.99999f : 0.0f;
        u = u - step;
    }
    const float U2 = (4.000016f * q) * (u * u);         // (2 lambda_up)^2
    return (q > 1e-12f) & (q < 1e12f) & (g_lo >= 2e-4f) & (t > 0.0f) & (t * t > U2);
}

__device__ __forceinline__ bool ek_far_certified_f32(const float (&S)[9], float Gsum,
                                                     int n_atoms, float cur)
{
    float S1[1][9], t1[1] = {ek_far_t_frame(Gsum, n_atoms, cur)};
    bool f1[1];
#pragma unroll
    for (int j = 0; j < 9; ++j)
        S1[0][j] = S[j];
    ek_far_certified_f32_w<1>(S1, t1, f1);
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
        // NaN and fails the comparison.  An UNDERFLOW does not: with q below
        // ~1e-19 (coordinates in metres, say) b = O(q^2) is zero in float32, U
        // comes out as sqrt(q), below the root, and a pair at exactly `cur` was
        // abandoned (round 5: tests/test_gpu_qcp_device.py, family "tiny", found
        // it; the host test only asked the certificate).  The bound is used only
        // where the certificate accepts q: [1e-12, 1e12].
        if (t > 0.0 && q > 1e-12 && q < 1e12) {
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
