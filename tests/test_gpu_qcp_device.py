"""The QCP solve, its early stop and the float32 far certificate of
enspara_amd/csrc/ek_qcp.h evaluated BY A KERNEL (ek_qcp_probe) on the inputs of
tests/test_qcp_host.py: there the header is compiled with g++, where 1/x, sqrt
and 1/sqrt are correctly rounded; the device issues v_rcp_f32 / v_sqrt_f32 /
v_rsq_f32 (1 ulp), and "callers pad every use" is a claim about THOSE
instructions.  Reference arithmetic: mdtraj.rmsd as bound at
enspara/cluster/util.py:289-291 (the QCP method; oracle/qcp_oracle.c).

 * full solve: the oracle's bits;
 * early stop: the same bits or +inf, +inf only where the distance is not below
   `cur` -- generic structures, 1-3 atoms, nearly collinear ones, coincident
   largest roots (where it must never stop);
 * certificate: the eleven adversarial spectrum families -- never "far" where
   the full iteration ends below `cur`, the largest root really separated,
   nothing certified outside the accepted range of q -- and the yield on
   generic spectra.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import qcp
from _qcp_cases import FAMILIES, coincident_case, family_case, structure_pairs

pytestmark = pytest.mark.gpu


def _probe(S, Gx, Gy, A, cur):
    from enspara_amd import _lib
    L = _lib.load()
    m = len(S)
    S = np.ascontiguousarray(S, dtype=np.float32)
    Gx = np.ascontiguousarray(Gx, dtype=np.float64)
    Gy = np.ascontiguousarray(Gy, dtype=np.float64)
    cur = np.ascontiguousarray(cur, dtype=np.float32)
    full = np.empty(m, dtype=np.float32)
    below = np.empty(m, dtype=np.float32)
    cert = np.empty(m, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    _lib.check(L.ek_qcp_probe(0, p(S), p(Gx), p(Gy), int(A), p(cur), C.c_int64(m),
                              p(full), p(below), p(cert)))
    # (bit 0: the closed-form certificate, bit 1: the second level; both have to be
    # sound wherever they speak, so the tests below ask their union)
    _probe.level2 = (cert & 2).astype(bool)
    return full, below, (cert & 3).astype(bool)


def _structures(rng, A, m, squash=1.0):
    x, y = structure_pairs(rng, A, m, squash)
    cx, Gx = qcp.center_and_trace(x)
    cy, Gy = qcp.center_and_trace(y)
    S = np.empty((m, 9), dtype=np.float32)
    for i in range(m):
        S[i] = qcp.S_matrices(cx[i:i + 1], cy[i])[0]
    want = np.array([np.sqrt(np.float32(qcp.msd_from_S(S[i], Gx[i], Gy[i], A)))
                     for i in range(m)], dtype=np.float32)
    return S, Gx, Gy, want.astype(np.float32)


def _check_below(S, Gx, Gy, A, want):
    rates = []
    for factor in (0.2, 0.7, 0.999, 1.0, 1.001, 1.5, np.inf):
        with np.errstate(invalid="ignore"):
            cur = (want * np.float32(factor)).astype(np.float32)
        full, below, cert = _probe(S, Gx, Gy, A, cur)
        np.testing.assert_array_equal(full.view(np.uint32), want.view(np.uint32))
        gave_up = np.isinf(below) & ~np.isinf(full)
        np.testing.assert_array_equal(below[~gave_up].view(np.uint32),
                                      want[~gave_up].view(np.uint32))
        with np.errstate(invalid="ignore"):
            assert not np.any(want[gave_up] < cur[gave_up])     # never a winner
            assert not np.any(want[cert] < cur[cert])
        if factor >= 1.0 and np.isfinite(factor):
            assert not np.any(gave_up & (want > 0))
            assert not np.any(cert & (want > 0))
        if not np.isfinite(factor):
            assert not gave_up.any()
        rates.append(gave_up.mean())
    return rates


@pytest.mark.parametrize("A", [1, 2, 3, 4, 7, 30])
def test_device_quartic_and_early_stop(A):
    rng = np.random.default_rng(100 + A)
    S, Gx, Gy, want = _structures(rng, A, 3000)
    rates = _check_below(S, Gx, Gy, A, want)
    if A >= 4:
        assert rates[0] > 0.9
    if A <= 2:
        assert max(rates) == 0.0       # rank-one S: always the full solve


@pytest.mark.parametrize("squash", [1e-1, 1e-2, 1e-3, 1e-5, 0.0])
def test_nearly_collinear_structures_on_the_device(squash):
    rng = np.random.default_rng(int(-np.log10(squash + 1e-9) * 10))
    for A in (3, 5, 12):
        S, Gx, Gy, want = _structures(rng, A, 1500, squash)
        _check_below(S, Gx, Gy, A, want)


@pytest.mark.parametrize("eps", [0.0, 1e-12, 1e-9, 1e-7, 1e-5, 1e-3, 1e-1])
def test_coincident_largest_roots_on_the_device(eps):
    m, A = 200000, 30
    S, Gx = coincident_case(eps, m, A)
    inf = np.full(m, np.inf, dtype=np.float32)
    full, below, cert = _probe(S, Gx, Gx, A, inf)
    np.testing.assert_array_equal(full.view(np.uint32), below.view(np.uint32))
    assert not cert.any()
    stops = 0
    for factor in (0.6, 0.9, 1.05, 1.2):
        cur = (full * np.float32(factor)).astype(np.float32)
        f2, below, cert = _probe(S, Gx, Gx, A, cur)
        np.testing.assert_array_equal(f2.view(np.uint32), full.view(np.uint32))
        gave_up = np.isinf(below) & ~np.isinf(full)
        np.testing.assert_array_equal(below[~gave_up].view(np.uint32),
                                      full[~gave_up].view(np.uint32))
        assert not np.any(full[gave_up] < cur[gave_up])
        assert not np.any(full[cert] < cur[cert])
        stops += int(gave_up.sum())
    if eps <= 1e-5:
        assert stops == 0          # never trusted
    if eps >= 1e-1:
        assert stops > m


@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("family", FAMILIES)
def test_float32_far_certificate_is_sound_on_the_device(family, seed):
    """tests/test_qcp_host.py::test_float32_far_certificate_is_sound through
    v_rsq_f32 / v_sqrt_f32 (two seeds of 150 000 matrices x 7 values of `cur`)"""
    m, A = 150000, 30
    S, Gsum, sv, t3, q, lam = family_case(family, m, A, seed)
    Gx = np.ascontiguousarray(Gsum / 2)
    inf = np.full(m, np.inf, dtype=np.float32)
    full, _, none = _probe(S, Gx, Gx, A, inf)
    assert not none.any()           # nothing is far from +inf
    total = 0
    for factor in (0.3, 0.8, 0.97, 0.9999, 1.0, 1.0002, 1.3):
        cur = (full * np.float32(factor)).astype(np.float32)
        f2, below, yes = _probe(S, Gx, Gx, A, cur)
        np.testing.assert_array_equal(f2.view(np.uint32), full.view(np.uint32))
        assert not np.any(full[yes] < cur[yes])
        assert np.all((sv[yes, 1] + t3[yes]) ** 2 >= 1e-4 * q[yes])
        gave_up = np.isinf(below) & ~np.isinf(full)
        assert not np.any(full[gave_up] < cur[gave_up])
        np.testing.assert_array_equal(below[~gave_up].view(np.uint32),
                                      full[~gave_up].view(np.uint32))
        if factor >= 1.0:
            assert not np.any(yes & (full > 0))
        if factor == 0.3 and family in ("generic", "small", "large"):
            roomy = ((Gsum - 2 * lam) > 2 * lam) & ((sv[:, 1] + t3) ** 2 > 0.01 * q) & \
                (q < 1e11)
            assert roomy.sum() > 500 and yes[roomy].mean() > 0.8
        if factor == 0.97 and family == "generic":
            # pairs 3 % beyond `cur`, roots clearly apart: the second level settles most
            sep = (sv[:, 1] + t3) ** 2 > 0.01 * q
            assert _probe.level2[sep].mean() > 0.3
        if family in ("tiny", "huge"):
            assert not yes.any()
        total += int(yes.sum())
    if family in ("s2~-s3", "rank1"):
        assert total < 0.12 * 7 * m
