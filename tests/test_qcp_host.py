"""The device header enspara_amd/csrc/ek_qcp.h compiled for the host (g++,
-ffp-contract=off as in the device build) against the oracle, on the CPU.

 * ek_rmsd_from_S gives the oracle's bits (oracle/qcp_oracle.c eko_msd_from_S:
   same inner-product matrix in, same float32 distance out);
 * ek_rmsd_from_S_below(.., cur) returns either those bits or +inf, and +inf
   only where the distance is not below `cur` -- for generic structures, for
   two- and three-atom ones, for nearly collinear ones, and for inner-product
   matrices built to have a (nearly) double largest root (singular values
   s1, s2, -s2): there the reference iteration ends in rounding noise or jumps,
   its result is whatever it is, and the early stop must not apply.
"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import qcp
from _qcp_cases import (FAMILIES, coincident_case, family_case,
                        structure_pairs as _pairs_of)

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "enspara_amd", "csrc")

SHIM = r'''
#define __device__
#define __forceinline__ inline
#include <cstdint>
#include "ek_qcp.h"
extern "C" void h_batch(const float *S, const double *Gx, const double *Gy,
                        int n_atoms, const float *cur, int64_t m, float *full,
                        float *below)
{
    for (int64_t i = 0; i < m; ++i) {
        float s[9];
        for (int j = 0; j < 9; ++j)
            s[j] = S[9 * i + j];
        full[i] = ek_rmsd_from_S(s, Gx[i], Gy[i], n_atoms);
        below[i] = ek_rmsd_from_S_below(s, Gx[i], Gy[i], n_atoms, cur[i]);
    }
}
extern "C" void h_cert(const float *S, const double *Gsum, int n_atoms,
                       const float *cur, int64_t m, unsigned char *out)
{
    for (int64_t i = 0; i < m; ++i) {
        float s[9];
        for (int j = 0; j < 9; ++j)
            s[j] = S[9 * i + j];
        out[i] = ek_far_certified_f32(s, (float)Gsum[i], n_atoms, cur[i]) ? 1 : 0;
        // bit 1: the second level (two Newton steps from the closed form's bound)
        if (ek_far_certified2_f32(s, ek_far_t_frame((float)Gsum[i], n_atoms, cur[i])))
            out[i] |= 2;
    }
}
'''


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("qcp_host")
    os.makedirs(d / "hip")
    (d / "hip" / "hip_runtime.h").write_text("")
    (d / "shim.cpp").write_text(SHIM)
    so = str(d / "qcp_host.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC",
                           "-I", str(d), "-I", CSRC, str(d / "shim.cpp"),
                           "-o", so])
    lib = C.CDLL(so)
    lib.h_batch.restype = None
    lib.h_cert.restype = None
    return lib


def _pairs(rng, A, m, squash):
    return _pairs_of(rng, A, m, squash)


def _run(host, rng, A, m, squash=1.0):
    x, y = _pairs(rng, A, m, squash)
    cx, Gx = qcp.center_and_trace(x)
    cy, Gy = qcp.center_and_trace(y)
    S = np.empty((m, 9), dtype=np.float32)
    for i in range(m):
        S[i] = qcp.S_matrices(cx[i:i + 1], cy[i])[0]
    want = np.array([np.sqrt(np.float32(qcp.msd_from_S(S[i], Gx[i], Gy[i], A)))
                     for i in range(m)], dtype=np.float32)
    out = []
    for factor in (0.2, 0.7, 0.999, 1.0, 1.001, 1.5, np.inf):
        with np.errstate(invalid="ignore"):      # 0 * inf: a NaN bound, never stops
            cur = (want * np.float32(factor)).astype(np.float32)
        full = np.empty(m, dtype=np.float32)
        below = np.empty(m, dtype=np.float32)
        host.h_batch(S.ctypes.data_as(C.c_void_p), Gx.ctypes.data_as(C.c_void_p),
                     Gy.ctypes.data_as(C.c_void_p), A,
                     cur.ctypes.data_as(C.c_void_p), C.c_int64(m),
                     full.ctypes.data_as(C.c_void_p),
                     below.ctypes.data_as(C.c_void_p))
        np.testing.assert_array_equal(full.view(np.uint32), want.view(np.uint32))
        gave_up = np.isinf(below) & ~np.isinf(full)
        kept = ~gave_up
        np.testing.assert_array_equal(below[kept].view(np.uint32),
                                      want[kept].view(np.uint32))
        with np.errstate(invalid="ignore"):
            assert not np.any(want[gave_up] < cur[gave_up])  # never a winner
        if factor >= 1.0 and np.isfinite(factor):
            assert not np.any(gave_up & (want > 0))          # d <= cur: solved
        if not np.isfinite(factor):
            assert not gave_up.any()
        out.append(gave_up.mean())
    return out


@pytest.mark.parametrize("A", [1, 2, 3, 4, 7, 30])
def test_host_compiled_device_quartic(host, A):
    rng = np.random.default_rng(100 + A)
    rates = _run(host, rng, A, 3000)
    if A >= 4:
        assert rates[0] > 0.9          # cur = 0.2 d: almost every solve stops early
    if A <= 2:
        assert max(rates) == 0.0       # rank-one S: always the full solve


@pytest.mark.parametrize("squash", [1e-1, 1e-2, 1e-3, 1e-5, 0.0])
def test_nearly_collinear_structures(host, squash):
    rng = np.random.default_rng(int(-np.log10(squash + 1e-9) * 10))
    for A in (3, 5, 12):
        _run(host, rng, A, 1500, squash)


@pytest.mark.parametrize("eps", [0.0, 1e-12, 1e-9, 1e-7, 1e-5, 1e-3, 1e-1])
def test_coincident_largest_roots(host, eps):
    """S = U diag(s1, s2, -s2 (1+eps)) V^T: the two largest roots of the quartic,
    s1+s2+s3 and s1-s2-s3, differ by 2 s2 eps.  A version of the early stop that
    trusted the iterates here returned +inf for ~1e-6 of these pairs although the
    full iteration ends below `cur`."""
    m, A = 200000, 30
    S, Gx = coincident_case(eps, m, A)
    Gy = Gx.copy()
    full = np.empty(m, dtype=np.float32)
    below = np.empty(m, dtype=np.float32)
    inf = np.full(m, np.inf, dtype=np.float32)
    args = (S.ctypes.data_as(C.c_void_p), Gx.ctypes.data_as(C.c_void_p),
            Gy.ctypes.data_as(C.c_void_p), A)
    host.h_batch(*args, inf.ctypes.data_as(C.c_void_p), C.c_int64(m),
                 full.ctypes.data_as(C.c_void_p), below.ctypes.data_as(C.c_void_p))
    np.testing.assert_array_equal(full.view(np.uint32), below.view(np.uint32))
    stops = 0
    for factor in (0.6, 0.9, 1.05, 1.2):
        cur = (full * np.float32(factor)).astype(np.float32)
        host.h_batch(*args, cur.ctypes.data_as(C.c_void_p), C.c_int64(m),
                     full.ctypes.data_as(C.c_void_p),
                     below.ctypes.data_as(C.c_void_p))
        gave_up = np.isinf(below) & ~np.isinf(full)
        np.testing.assert_array_equal(below[~gave_up].view(np.uint32),
                                      full[~gave_up].view(np.uint32))
        assert not np.any(full[gave_up] < cur[gave_up])
        stops += int(gave_up.sum())
    if eps <= 1e-5:
        assert stops == 0          # never trusted
    if eps >= 1e-1:
        assert stops > m           # separated again: most far solves stop early


@pytest.mark.parametrize("family", FAMILIES)
def test_float32_far_certificate_is_sound(host, family):
    """ek_far_certified_f32: S = U diag(s1, s2, s3) V^T with the spectrum drawn
    from the family named -- generic, and every way two roots of the quartic
    can come close -- and `cur` placed around the true distance.  Where the
    certificate says "far": the full reference iteration ends at or above
    `cur`; the largest root really is separated ((s2 + s3)^2 >= 1e-4 q, from a
    float64 SVD); on generic spectra -- also scaled by 1e-5 and 1e4, inside the
    range of q it accepts -- it certifies most pairs that are far by a margin;
    scaled by 1e-15 or 1e9 (raw cofactors outside the float32 range) nothing."""
    m, A = 300000, 30
    S, Gsum, sv, t3, q, lam = family_case(family, m, A)
    Gx = np.ascontiguousarray(Gsum / 2)
    full = np.empty(m, dtype=np.float32)
    below = np.empty(m, dtype=np.float32)
    inf = np.full(m, np.inf, dtype=np.float32)
    host.h_batch(S.ctypes.data_as(C.c_void_p), Gx.ctypes.data_as(C.c_void_p),
                 Gx.ctypes.data_as(C.c_void_p), A, inf.ctypes.data_as(C.c_void_p),
                 C.c_int64(m), full.ctypes.data_as(C.c_void_p),
                 below.ctypes.data_as(C.c_void_p))
    cert = np.empty(m, dtype=np.uint8)
    total = 0
    for factor in (0.3, 0.8, 0.97, 0.9999, 1.0, 1.0002, 1.3):
        cur = (full * np.float32(factor)).astype(np.float32)
        host.h_cert(S.ctypes.data_as(C.c_void_p), Gsum.ctypes.data_as(C.c_void_p), A,
                    cur.ctypes.data_as(C.c_void_p), C.c_int64(m),
                    cert.ctypes.data_as(C.c_void_p))
        yes = (cert & 1).astype(bool)
        assert not np.any(full[yes] < cur[yes])
        # the second level (round 5): sound the same way, separated the same way, and
        # it settles what the first does (its bound starts from the first's)
        yes2 = (cert & 2).astype(bool)
        assert not np.any(full[yes2] < cur[yes2])
        assert np.all((sv[yes2, 1] + t3[yes2]) ** 2 >= 1e-4 * q[yes2])
        assert (yes & ~yes2).sum() <= 2e-3 * max(int(yes.sum()), 1)
        if factor >= 1.0:
            assert not np.any(yes2 & (full > 0))
        if family in ("tiny", "huge"):
            assert not yes2.any()
        if factor == 0.97 and family == "generic":
            # pairs 3 % beyond `cur`: the closed form leaves many, two Newton steps few
            sep = (sv[:, 1] + t3) ** 2 > 0.01 * q
            assert yes2[sep].mean() > yes[sep].mean() + 0.05, (yes[sep].mean(), yes2[sep].mean())
        # the early-stopped solve on the same matrices: the same bits or +inf, +inf
        # never where the distance is below `cur` (round 5: at scale 1e-15 the
        # closed-form bound of round 4 underflowed in float32 and was not sound)
        f2 = np.empty(m, dtype=np.float32)
        host.h_batch(S.ctypes.data_as(C.c_void_p), Gx.ctypes.data_as(C.c_void_p),
                     Gx.ctypes.data_as(C.c_void_p), A, cur.ctypes.data_as(C.c_void_p),
                     C.c_int64(m), f2.ctypes.data_as(C.c_void_p),
                     below.ctypes.data_as(C.c_void_p))
        gave_up = np.isinf(below) & ~np.isinf(full)
        assert not np.any(full[gave_up] < cur[gave_up])
        np.testing.assert_array_equal(below[~gave_up].view(np.uint32),
                                      full[~gave_up].view(np.uint32))
        assert np.all((sv[yes, 1] + t3[yes]) ** 2 >= 1e-4 * q[yes])
        if factor >= 1.0:
            assert not np.any(yes & (full > 0))
        if factor == 0.3 and family in ("generic", "small", "large"):
            # (pairs whose mean square distance is not small beside lambda / A --
            # the closed-form bound of lambda_max is a crude one --, with the
            # largest root clearly apart, and q inside [1e-12, 1e12])
            roomy = ((Gsum - 2 * lam) > 2 * lam) & ((sv[:, 1] + t3) ** 2 > 0.01 * q) & \
                (q < 1e11)
            assert roomy.sum() > 1000 and yes[roomy].mean() > 0.8
        if family in ("tiny", "huge"):
            assert not yes.any()
        total += int(yes.sum())
    if family in ("s2~-s3", "rank1"):
        assert total < 0.12 * 7 * m     # (only where `tiny` is not tiny)
