"""Host side of the MSM eigensolver (Arnoldi / Krylov-Schur driver in
enspara_amd/msm/transition_matrices.py) with a numpy Krylov space standing in
for the device, against scipy and the reference's outputs."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse
import scipy.sparse.linalg

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _numpy_krylov import NumpyKrylov  # noqa: E402
from enspara_amd.msm import transition_matrices as tm  # noqa: E402


def _factory(A, m):
    return NumpyKrylov(A, m)


def _rowstoch(n, density, seed):
    rng = np.random.RandomState(seed)
    C = scipy.sparse.random(n, n, density=density, random_state=rng,
                            format="csr")
    C = C + scipy.sparse.diags(np.ones(n)) + \
        scipy.sparse.diags(np.ones(n - 1) * 0.5, 1) + \
        scipy.sparse.diags(np.ones(n - 1) * 0.5, -1)
    C = scipy.sparse.csr_matrix(C)
    w = np.asarray(C.sum(axis=1)).ravel()
    return scipy.sparse.diags(1.0 / w) @ C


def test_small_dense_all_eigs_matches_reference(golden_dir):
    M = np.load(os.path.join(golden_dir, "msm_golden.npz"))
    T = scipy.sparse.csr_matrix(M["norm_T"])
    vals, vecs = tm.eigenspectrum(T, n_eigs=5, _space_factory=_factory)
    np.testing.assert_allclose(vals, M["eig_vals"], atol=1e-9)
    np.testing.assert_allclose(vecs[:, 0], M["eig_vecs"][:, 0], atol=1e-9)
    np.testing.assert_allclose(vecs[:, 0], M["norm_eq"], atol=1e-9)
    # other vectors: eigenvectors of T^T where the eigenvalue is real (for a
    # complex pair the real part of the vector depends on an arbitrary phase,
    # in the reference's output as well)
    w = np.linalg.eigvals(M["norm_T"].T)
    w = w[np.argsort(-w.real)][:5]
    for i in range(1, 5):
        if abs(w[i].imag) > 1e-12:
            continue
        x = vecs[:, i]
        r = T.T @ x - vals[i] * x
        assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(x)


def test_three_state_matrix():
    """reference test_msm_funcs.py:96-117 style: tiny dense input"""
    T = np.array([[0.9, 0.1, 0.0], [0.05, 0.9, 0.05], [0.0, 0.2, 0.8]])
    vals, vecs = tm.eigenspectrum(T, n_eigs=3, _space_factory=_factory)
    w, v = np.linalg.eig(T.T)
    o = np.argsort(-w.real)
    np.testing.assert_allclose(vals, w.real[o], atol=1e-12)
    pi = v[:, o[0]].real
    pi /= pi.sum()
    np.testing.assert_allclose(vecs[:, 0], pi, atol=1e-12)
    with pytest.raises(ValueError):
        tm.eigenspectrum(T, n_eigs=1, _space_factory=_factory)


def test_restarted_krylov_schur_matches_arpack():
    T = _rowstoch(3000, 0.002, 3)
    vals, vecs = tm.eigenspectrum(T, n_eigs=8, _space_factory=_factory)
    ref = scipy.sparse.linalg.eigs(scipy.sparse.csr_matrix(T.T), 8,
                                   which="LR", tol=1e-12)[0]
    ref = np.sort(ref.real)[::-1]
    np.testing.assert_allclose(vals, ref, atol=1e-8)
    assert abs(vals[0] - 1.0) < 1e-10
    pi = vecs[:, 0]
    assert abs(pi.sum() - 1) < 1e-12 and np.all(pi > -1e-12)
    np.testing.assert_allclose(T.T @ pi, pi, atol=1e-9)


def _hopping_blocks(n_blocks, size, seed, frames=400000):
    """the bench's kind of chain: a symmetric banded walk inside blocks, rare hops
    from block to block in ONE direction -- n_blocks eigenvalues in a tight, slightly
    complex cluster at 1, a real bulk below"""
    rng = np.random.RandomState(seed)
    steps = rng.choice(np.array([-3, -2, -1, 0, 0, 1, 2, 3]), size=frames)
    inb = (rng.randint(size) + np.cumsum(steps)) % size
    blk = (rng.randint(n_blocks) + 7 * np.cumsum(rng.rand(frames) < 0.002)) % n_blocks
    a = blk * size + inb
    K = n_blocks * size
    C = scipy.sparse.coo_matrix((np.ones(frames - 1), (a[:-1], a[1:])), shape=(K, K)).tocsr()
    C = C + scipy.sparse.diags(np.full(K, 1e-3))        # (no empty rows)
    return scipy.sparse.diags(1.0 / np.asarray(C.sum(axis=1)).ravel()) @ C


def test_polynomial_filter_same_eigenpairs_fewer_restarts(monkeypatch):
    """Round 5: once a cycle has shown where the wanted eigenvalues end, the restarted
    iteration goes on with a Chebyshev polynomial of the matrix (FILTER = 1) and takes
    the MATRIX's pairs from the converged subspace by Rayleigh-Ritz.  Same eigenvalues
    as ARPACK and as the plain iteration (FILTER = 0) where the leading ones cluster at
    1 -- in a fraction of the restarts --, an interval that starts out too high is
    lowered, and a spectrum with a complex periphery (a polynomial on a real interval
    would lift that above the wanted real ones) is left to the plain iteration."""
    T = _hopping_blocks(15, 100, 5)
    ref = np.sort(scipy.sparse.linalg.eigs(scipy.sparse.csr_matrix(T.T), 10, which="LR",
                                           tol=1e-12)[0].real)[::-1]
    runs = {}
    for f in (0, 1):
        monkeypatch.setattr(tm, "FILTER", f)
        vals, vecs = tm.eigenspectrum(T, n_eigs=10, _space_factory=_factory)
        np.testing.assert_allclose(vals, ref, atol=1e-10)
        runs[f] = dict(tm.LAST_RUN)
    assert runs[0]["filter"] is None and runs[1]["filter"] is not None
    assert not runs[1]["fallback"]
    assert sum(runs[1]["restarts"]) * 3 <= sum(runs[0]["restarts"])
    # the first vector is the stationary distribution either way
    assert abs(vecs[:, 0].sum() - 1.0) < 1e-9 and np.all(vecs[:, 0] > -1e-12)
    # an interval that ends above wanted eigenvalues: found out, lowered
    plan = tm._plan_filter
    monkeypatch.setattr(tm, "_plan_filter",
                        lambda th, k, scout=None: (48, -1.0, 0.9999, k + 4)
                        if scout is not None else plan(th, k, scout))
    vals, _ = tm.eigenspectrum(T, n_eigs=10, _space_factory=_factory)
    np.testing.assert_allclose(vals, ref, atol=1e-10)
    assert tm.LAST_RUN["plans"] >= 2 and tm.LAST_RUN["filter"]["b"] < 0.9999
    monkeypatch.setattr(tm, "_plan_filter", plan)
    # complex periphery: no polynomial
    T2 = _rowstoch(3000, 0.002, 3)
    ref2 = np.sort(scipy.sparse.linalg.eigs(scipy.sparse.csr_matrix(T2.T), 8, which="LR",
                                            tol=1e-12)[0].real)[::-1]
    vals2, _ = tm.eigenspectrum(T2, n_eigs=8, _space_factory=_factory)
    np.testing.assert_allclose(vals2, ref2, atol=1e-8)
    assert tm.LAST_RUN["filter"] is None


def test_polynomial_filter_planning_units():
    """The pieces of the filtered iteration on their own: the Chebyshev map and its
    inverse beyond the interval, the first plan's interval from Ritz values read as
    samples of the spectrum, and its refusal where eigenvalues off the real axis would
    come out of the polynomial above the band the damped real ones are confined to."""
    x = np.linspace(-2.0, 2.0, 41)
    for d in (2, 5, 12):
        want = np.polynomial.chebyshev.chebval(x, [0] * d + [1])
        np.testing.assert_allclose(tm._cheb(x, d).real, want, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(tm._cheb(x, d).imag, 0.0, atol=1e-9 * np.abs(want).max())
    a, b, d = -1.0, 0.9, 14
    c, e = 0.5 * (a + b), 0.5 * (b - a)
    lam = np.array([0.91, 0.95, 0.999, 1.0, 0.97 + 0.002j])
    back = tm._unfilter(tm._cheb((lam - c) / e, d), (d, a, b))
    np.testing.assert_allclose(back, lam, atol=1e-10)
    # what the polynomial leaves inside [-1, 1] is only known to lie below b
    inside = tm._unfilter(tm._cheb((np.array([0.3, -0.5]) - c) / e, d), (d, a, b))
    np.testing.assert_allclose(inside, b)
    # the first plan: 60 Ritz values of a 5000-state matrix, 20 wanted
    theta = np.concatenate([[1.0, 0.998, 0.998, 0.993, 0.99, 0.987, 0.97],
                            np.linspace(0.95, -0.2, 53)]).astype(complex)
    plan = tm._plan_filter(theta, 20, scout=(5000, 60))
    assert plan is not None
    deg, pa, pb, kf = plan
    assert kf == 25 and pa <= -1.0 and 0.98 < pb < 0.995 and 4 <= deg <= 64
    assert tm._cheb(np.array([(1.0 - 0.5 * (pa + pb)) / (0.5 * (pb - pa))]), deg).real[0] > 20
    # a pair far off the real axis below the interval: no polynomial
    theta2 = theta.copy()
    theta2[-2:] = [0.1 + 0.7j, 0.1 - 0.7j]
    assert tm._plan_filter(theta2, 20, scout=(5000, 60)) is None
    # the plan after a cycle on a polynomial: from estimates, the interval below what a
    # restart keeps, every pair to converge lifted well clear of the band
    est = np.concatenate([np.linspace(1.0, 0.9979, 26), np.linspace(0.9972, 0.99, 34)])
    plan2 = tm._plan_filter(est.astype(complex), 20)
    assert plan2 is not None and plan2[2] < est[25]
    c2, e2 = 0.5 * (plan2[1] + plan2[2]), 0.5 * (plan2[2] - plan2[1])
    assert tm._cheb(np.array([(est[24] - c2) / e2]), plan2[0]).real[0] >= 4.0


def test_reducible_matrix_breakdown():
    # two disconnected blocks: Arnoldi breaks down, must continue
    A = _rowstoch(40, 0.2, 1)
    B = _rowstoch(30, 0.2, 2)
    T = scipy.sparse.block_diag([A, B]).tocsr()
    vals, _ = tm.eigenspectrum(T, n_eigs=4, _space_factory=_factory)
    w = np.sort(np.linalg.eigvals(T.toarray().T).real)[::-1]
    np.testing.assert_allclose(vals, w[:4], atol=1e-9)
    assert abs(vals[1] - 1.0) < 1e-10          # eigenvalue 1 twice


def _metastable(n_blocks, size, eps, seed):
    """n_blocks nearly uncoupled chains: the leading n_blocks eigenvalues
    cluster at 1 within ~eps"""
    blocks = [_rowstoch(size, 0.05, seed + b).toarray() for b in range(n_blocks)]
    T = scipy.linalg.block_diag(*blocks) * (1.0 - eps)
    n = n_blocks * size
    rng = np.random.RandomState(seed)
    for i in range(n):                  # eps of every row's weight anywhere
        T[i, rng.randint(0, n)] += eps
    return scipy.sparse.csr_matrix(T / T.sum(axis=1, keepdims=True))


@pytest.mark.parametrize("eps", [1e-6, 1e-9])
def test_eigenvalues_clustered_at_one(eps):
    """six metastable blocks coupled at 1e-6 / 1e-9: six eigenvalues within
    ~eps of 1 -- where a Schur form's eigenvalues may be "too close to swap"
    (dtrsen info = 1); the restarts must go on, not raise"""
    T = _metastable(6, 250, eps, 11)
    vals, vecs = tm.eigenspectrum(T, n_eigs=8, _space_factory=_factory)
    w = np.sort(np.linalg.eigvals(T.toarray().T).real)[::-1]
    np.testing.assert_allclose(vals, w[:8], atol=1e-8)
    assert np.all(vals[:6] > 1 - 100 * eps)
    pi = vecs[:, 0]
    np.testing.assert_allclose(T.T @ pi, pi, atol=1e-9)


def test_restart_survives_a_partial_reordering(monkeypatch):
    """dtrsen reports info = 1 (eigenvalues too close to swap: the Schur form is
    only partly reordered): round 4 raised LinAlgError there; the restart goes
    on with what was kept and converges to the same eigenvalues"""
    real = scipy.linalg.lapack.dtrsen
    calls = {"n": 0}

    def flaky(select, S, Z, **kw):
        out = list(real(select, S, Z, **kw))
        calls["n"] += 1
        if calls["n"] % 3 == 1:         # every third restart: info 1, one fewer kept
            out[-1] = 1
            out[4] = max(1, int(out[4]) - 1)
        return tuple(out)

    monkeypatch.setattr(scipy.linalg.lapack, "dtrsen", flaky)
    T = _rowstoch(3000, 0.002, 3)
    vals, _ = tm.eigenspectrum(T, n_eigs=8, _space_factory=_factory)
    assert calls["n"] > 1
    ref = scipy.sparse.linalg.eigs(scipy.sparse.csr_matrix(T.T), 8,
                                   which="LR", tol=1e-12)[0]
    np.testing.assert_allclose(vals, np.sort(ref.real)[::-1], atol=1e-8)


# ---- ergodic trimming (host-side graph work) ----------------------------------
TRIM_ARR_TYPES = [np.array, scipy.sparse.lil_matrix, scipy.sparse.csr_matrix,
                  scipy.sparse.coo_matrix, scipy.sparse.csc_matrix,
                  scipy.sparse.dia_matrix, scipy.sparse.dok_matrix]


def _dense(m):
    return m.toarray() if hasattr(m, "toarray") else np.asarray(m)


@pytest.mark.parametrize("arr_type", TRIM_ARR_TYPES)
def test_trim_disconnected_known_answers(arr_type):
    """inputs and expected outputs of the reference's own test
    (enspara/test/test_msm_funcs.py:273-311)"""
    from enspara_amd.msm import trim_disconnected, TrimMapping
    given = arr_type([[1, 2, 0, 0],
                      [2, 1, 0, 1],
                      [0, 0, 1, 0],
                      [0, 1, 0, 2]])
    mapping, trimmed = trim_disconnected(given)
    assert type(trimmed) is type(given)
    np.testing.assert_array_equal(_dense(trimmed), [[1, 2, 0], [2, 1, 1],
                                                    [0, 1, 2]])
    assert mapping == TrimMapping([(0, 0), (1, 1), (3, 2)])
    mapping, trimmed = trim_disconnected(given, threshold=2)
    np.testing.assert_array_equal(_dense(trimmed), [[1, 2], [2, 1]])
    assert mapping == TrimMapping([(0, 0), (1, 1)])


def test_trim_mapping_surface(tmp_path):
    """construction, inverse, CSV round trip (test_msm_funcs.py:26-60)"""
    from enspara_amd.msm import TrimMapping
    a = TrimMapping()
    a.to_original = {0: 0, 1: 1, 2: 3, 3: 7}
    b = TrimMapping()
    b.to_mapped = {0: 0, 1: 1, 3: 2, 7: 3}
    assert a == b
    tm = TrimMapping([(0, 0), (1, -1), (2, 1), (3, 2)])
    path = tmp_path / "m.csv"
    tm.save(str(path))
    assert path.read_text().split("\n") == ["original,mapped", "0,0", "1,-1",
                                            "2,1", "3,2", ""]
    assert TrimMapping.load(str(path)) == tm
    assert tm == [(0, 0), (1, -1), (2, 1), (3, 2)]
    assert not (tm == TrimMapping([(0, 0)]))


def test_trim_disconnected_matches_reference_outputs(golden_dir):
    """tests/golden/trim_golden.npz: outputs of the real trim_disconnected on a
    40-state count matrix with a closed small component, a source-only and a
    sink-only state, for three thresholds, renumbered or not."""
    from enspara_amd.msm import trim_disconnected
    G = np.load(os.path.join(golden_dir, "trim_golden.npz"))
    C = scipy.sparse.coo_matrix(G["counts"])
    for thr in (1, 2, 5):
        for ren in (True, False):
            key = "thr%d_ren%d" % (thr, int(ren))
            for given in (C, C.tocsr(), G["counts"]):
                m, tc = trim_disconnected(given, threshold=thr,
                                          renumber_states=ren)
                assert type(tc) is type(given)
                np.testing.assert_array_equal(_dense(tc), G[key + "_counts"])
                np.testing.assert_array_equal(
                    np.array(sorted(m.to_original.items())), G[key + "_map"])


def _hand_fitted_msm():
    """An MSM whose fitted fields are filled in on the host (fit itself runs
    the device count kernel; tests/test_gpu_msm.py covers fit -> save -> load)."""
    from enspara_amd.msm import MSM, builders
    from enspara_amd.msm.trimming import TrimMapping
    m = MSM(lag_time=3, method=builders.normalize, trim=True)
    rng = np.random.default_rng(5)
    C = scipy.sparse.random(40, 40, density=0.2, random_state=rng,
                            format="csr", dtype=np.float64)
    C.data = np.ceil(C.data * 20)
    C = C + scipy.sparse.identity(40, format="csr")
    w = np.asarray(C.sum(axis=1)).ravel()
    m.tcounts_ = C
    m.tprobs_ = scipy.sparse.diags(1.0 / w) @ C
    m.eq_probs_ = w / w.sum()
    m.mapping_ = TrimMapping(zip(range(2, 42), range(40)))
    return m


def test_msm_save_load_roundtrip(tmp_path):
    """reference test_msm_obj.py:60-111"""
    from enspara_amd.msm import MSM
    m = _hand_fitted_msm()
    d = str(tmp_path / "model")
    m.save(d)
    assert sorted(os.listdir(d)) == ["config.pkl", "eq-probs.dat",
                                     "manifest.json", "mapping.csv",
                                     "tcounts.mtx", "tprobs.mtx"]
    back = MSM.load(d)
    assert back == m and back.lag_time == 3 and back.trim is True
    assert (back.tprobs_ != m.tprobs_).nnz == 0          # bit-for-bit

    os.rename(os.path.join(d, "manifest.json"), os.path.join(d, "other.json"))
    assert MSM.load(d, manifest="other.json") == m

    d2 = str(tmp_path / "renamed")
    names = {"tprobs_": "p.mtx", "tcounts_": "c.mtx", "eq_probs_": "pi.dat",
             "mapping_": "map.csv"}
    m.save(d2, **names)
    for name in names.values():
        assert os.path.isfile(os.path.join(d2, name))
    assert MSM.load(d2) == m

    with pytest.raises(FileExistsError):
        m.save(d2)
    m.save(d2, force=True)
    assert MSM.load(d2) == m
    with pytest.raises(NotImplementedError):
        m.save(str(tmp_path / "z"), zipfile=True)
    with pytest.raises(NotImplementedError):
        MSM.load(os.path.join(d2, "p.mtx"))


def test_msm_pickle_roundtrip():
    """reference test_msm_obj.py:114-127"""
    import pickle
    m = _hand_fitted_msm()
    assert pickle.loads(pickle.dumps(m)) == m
