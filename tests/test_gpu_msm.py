"""MSM kernels on the GPU against outputs of the real reference
(tests/golden/msm_golden.npz: enspara.msm.assigns_to_counts /
builders.normalize / builders.transpose / eigenspectrum run here by
tests/golden/make_golden.py) and against scipy at larger sizes."""
import os

import numpy as np
import pytest
import scipy.sparse
import scipy.sparse.linalg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M(golden_dir):
    return np.load(os.path.join(golden_dir, "msm_golden.npz"))


def test_counts_match_reference(M):
    from enspara_amd.msm import assigns_to_counts
    from enspara_amd import ra
    A = M["assigns"]
    K = 60
    for lag in (1, 5):
        for sw in (True, False):
            C = assigns_to_counts(A, lag_time=lag, max_n_states=K,
                                  sliding_window=sw)
            want = M["counts_lag%d_sw%d" % (lag, int(sw))]
            assert C.shape == (K, K) and scipy.sparse.issparse(C)
            np.testing.assert_array_equal(np.asarray(C.todense()), want)
            assert np.issubdtype(C.dtype, np.integer)
    lens = M["rag_lengths"]
    flat = np.concatenate([A[i, :n] for i, n in enumerate(lens)])
    R = ra.RaggedArray(flat, lengths=lens)
    C = assigns_to_counts(R, lag_time=3, max_n_states=K)
    np.testing.assert_array_equal(np.asarray(C.todense()),
                                  M["rag_counts_lag3"])
    # max_n_states inferred
    C = assigns_to_counts(A, lag_time=1)
    assert C.shape == (A.max() + 1, A.max() + 1)


def test_counts_edge_cases():
    from enspara_amd.msm import assigns_to_counts
    from enspara_amd.exception import DataInvalid
    a = np.array([[0, 1, -1, 1, 2, -1, -1, 0]])
    # -1 frames are dropped first: 0 1 1 2 0
    C = assigns_to_counts(a, lag_time=1).toarray()
    want = np.zeros((3, 3), dtype=int)
    for i, j in [(0, 1), (1, 1), (1, 2), (2, 0)]:
        want[i, j] += 1
    np.testing.assert_array_equal(C, want)
    C = assigns_to_counts(a, lag_time=2, sliding_window=False).toarray()
    want = np.zeros((3, 3), dtype=int)
    for i, j in [(0, 1), (1, 0)]:
        want[i, j] += 1
    np.testing.assert_array_equal(C, want)
    # trajectory shorter than the lag contributes nothing
    C = assigns_to_counts([np.array([0, 1]), np.array([2, 2, 2, 2])],
                          lag_time=3).toarray()
    assert C.sum() == 1 and C[2, 2] == 1
    # 20 000 states (BASELINE.json configs[3]'s center count): a 1.6 GB count table
    C = assigns_to_counts(a, lag_time=1, max_n_states=20000)
    assert C.shape == (20000, 20000) and C.nnz == 4
    np.testing.assert_array_equal(C.tocsr()[:3, :3].toarray(),
                                  [[0, 1, 0], [0, 1, 1], [1, 0, 0]])
    big = np.array([[19999, 0, 19999, 19999, -1, 17000, 0]])
    C = assigns_to_counts(big, lag_time=1, max_n_states=20000).tocoo()
    got = sorted(zip(C.row.tolist(), C.col.tolist(), C.data.tolist()))
    assert got == [(0, 19999, 1), (17000, 0, 1), (19999, 0, 1), (19999, 17000, 1),
                   (19999, 19999, 1)]
    with pytest.raises(DataInvalid):
        assigns_to_counts(np.array([0, 1, 2]), lag_time=1)
    with pytest.raises(DataInvalid):
        assigns_to_counts(a, lag_time=0)
    with pytest.raises(DataInvalid):
        assigns_to_counts(a, lag_time=1.5)
    with pytest.raises(ValueError):
        assigns_to_counts(a, lag_time=1, max_n_states=2)


def test_normalize_and_transpose_match_reference(M):
    from enspara_amd.msm import assigns_to_counts, builders
    C = assigns_to_counts(M["assigns"], lag_time=1, max_n_states=60)
    C2, T, eq = builders.normalize(C, calculate_eq_probs=True)
    assert type(T) is type(C)
    np.testing.assert_array_equal(np.asarray(T.todense()), M["norm_T"])
    np.testing.assert_allclose(eq, M["norm_eq"], atol=1e-10)
    Cs, Tt, eqt = builders.transpose(C, calculate_eq_probs=True)
    np.testing.assert_array_equal(np.asarray(Tt.todense()), M["transpose_T"])
    np.testing.assert_allclose(eqt, M["transpose_eq"], atol=1e-14)
    # dense input, zero rows stay zero
    D = np.array([[2., 2., 0.], [0., 0., 0.], [1., 0., 3.]])
    Tn = builders._row_normalize(D)
    np.testing.assert_array_equal(Tn, [[.5, .5, 0], [0, 0, 0], [.25, 0, .75]])


def test_eigenspectrum_matches_reference(M):
    from enspara_amd.msm import eigenspectrum
    T = scipy.sparse.csr_matrix(M["norm_T"])
    vals, vecs = eigenspectrum(T, n_eigs=5)
    np.testing.assert_allclose(vals, M["eig_vals"], atol=1e-9)
    np.testing.assert_allclose(vecs[:, 0], M["eig_vecs"][:, 0], atol=1e-9)


def test_polynomial_filter_on_the_device():
    """Round 5: the restarted iteration applies a Chebyshev polynomial of the matrix
    inside its Arnoldi steps (ek_krylov_set_filter, kr_spmv_cheb_kernel) where the
    leading eigenvalues cluster at 1 -- the bench's kind of chain: banded walks inside
    blocks, rare one-way hops between them.  Same eigenvalues as ARPACK and as the
    plain iteration on the device, in a fraction of the restarts; the same plan and
    restarts as the numpy stand-in makes (tests/test_msm_host.py holds that one
    against ARPACK on the CPU)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_msm_host import _hopping_blocks, _factory
    from enspara_amd.msm import transition_matrices as tm
    T = _hopping_blocks(15, 100, 5)
    ref = np.sort(scipy.sparse.linalg.eigs(scipy.sparse.csr_matrix(T.T), 10, which="LR",
                                           tol=1e-12)[0].real)[::-1]
    runs = {}
    old = tm.FILTER
    try:
        for f in (0, 1):
            tm.FILTER = f
            vals, vecs = tm.eigenspectrum(T, n_eigs=10)
            np.testing.assert_allclose(vals, ref, atol=1e-10)
            runs[f] = dict(tm.LAST_RUN)
        assert abs(vecs[:, 0].sum() - 1.0) < 1e-9 and np.all(vecs[:, 0] > -1e-12)
        tm.FILTER = 1
        tm.eigenspectrum(T, n_eigs=10, _space_factory=_factory)
        host = dict(tm.LAST_RUN)
    finally:
        tm.FILTER = old
    assert runs[0]["filter"] is None and runs[1]["filter"] is not None
    assert not runs[1]["fallback"]
    assert sum(runs[1]["restarts"]) * 3 <= sum(runs[0]["restarts"])
    assert runs[1]["filter"]["degree"] == host["filter"]["degree"]
    assert abs(runs[1]["filter"]["b"] - host["filter"]["b"]) < 1e-9


def test_msm_build_at_scale():
    """seeded 2000-state, 1e6-frame synthetic assignments: counts equal a
    scipy construction exactly; top-10 eigenvalues equal ARPACK's to 1e-8"""
    from enspara_amd.msm import MSM, eigenspectrum
    rng = np.random.RandomState(5)
    K, n_trj, L = 2000, 100, 10000
    # banded walk inside 20 metastable blocks of 100 states with rare hops
    # between blocks: a spectrum with gaps (a plain ring walk has eigenvalues
    # packed at 1 - O(1/K^2) that ARPACK itself does not resolve)
    steps = rng.choice([-3, -2, -1, 0, 0, 1, 2, 3], size=(n_trj, L))
    inblock = (rng.randint(100, size=(n_trj, 1)) + np.cumsum(steps, axis=1)) % 100
    hops = np.cumsum(rng.rand(n_trj, L) < 0.002, axis=1)
    block = (rng.randint(20, size=(n_trj, 1)) + hops * 7) % 20
    A = block * 100 + inblock
    A[rng.rand(n_trj, L) < 0.001] = -1
    m = MSM(lag_time=2, method="normalize", max_n_states=K)
    m.fit(A)
    rows, cols = [], []
    for a in A:
        a = a[a != -1]
        rows.append(a[:-2])
        cols.append(a[2:])
    ref = scipy.sparse.coo_matrix(
        (np.ones(sum(len(r) for r in rows), dtype=int),
         (np.concatenate(rows), np.concatenate(cols))), shape=(K, K)).tocsr()
    assert (m.tcounts_.tocsr() != ref).nnz == 0
    Tref = scipy.sparse.diags(1.0 / np.asarray(ref.sum(axis=1)).ravel()) @ ref
    assert abs(m.tprobs_.tocsr() - Tref).max() < 1e-15
    vals, vecs = eigenspectrum(m.tprobs_, n_eigs=10)
    want = np.linalg.eigvals(Tref.toarray())
    want = want[np.argsort(-want.real)][:10]
    np.testing.assert_allclose(vals, want.real, atol=1e-8)
    np.testing.assert_allclose(m.eq_probs_, vecs[:, 0], atol=1e-9)
    assert abs(m.eq_probs_.sum() - 1) < 1e-12


def _scipy_counts(A, lag, K):
    rows, cols = [], []
    for a in A:
        a = a[a != -1]
        rows.append(a[:-lag])
        cols.append(a[lag:])
    return scipy.sparse.coo_matrix(
        (np.ones(sum(len(r) for r in rows), dtype=np.int64),
         (np.concatenate(rows), np.concatenate(cols))), shape=(K, K)).tocsr()


def test_msm_build_at_full_size():
    """BASELINE.json configs[4] at its own size: 10^7 frames of assignments
    (1000 trajectories of 10^4, ~0.1 % of them -1) over 5000 states -> counts
    (exact against a scipy construction), row-normalised probabilities, top-20
    eigenvalues (against ARPACK on the same matrix, 1e-8)"""
    import scipy.sparse.linalg
    from enspara_amd.msm import MSM, eigenspectrum
    rng = np.random.RandomState(11)
    K, n_trj, L, lag = 5000, 1000, 10000, 1
    # 50 metastable blocks of 100 states, rare hops between blocks (a spectrum
    # with gaps: see test_msm_build_at_scale)
    steps = rng.choice(np.array([-3, -2, -1, 0, 0, 1, 2, 3], dtype=np.int8),
                       size=(n_trj, L))
    inblock = (rng.randint(100, size=(n_trj, 1)) +
               np.cumsum(steps, axis=1, dtype=np.int32)) % 100
    hops = np.cumsum(rng.rand(n_trj, L) < 0.002, axis=1, dtype=np.int32)
    block = (rng.randint(50, size=(n_trj, 1)) + hops * 7) % 50
    A = (block * 100 + inblock).astype(np.int32)
    A[rng.rand(n_trj, L) < 0.001] = -1
    del steps, inblock, hops, block
    m = MSM(lag_time=lag, method="normalize", max_n_states=K)
    m.fit(A)
    ref = _scipy_counts(A, lag, K)
    assert (m.tcounts_.tocsr() != ref).nnz == 0
    Tref = scipy.sparse.diags(1.0 / np.asarray(ref.sum(axis=1)).ravel()) @ ref
    assert abs(m.tprobs_.tocsr() - Tref).max() < 1e-15
    vals, vecs = eigenspectrum(m.tprobs_, n_eigs=20)
    want = scipy.sparse.linalg.eigs(Tref.T.tocsr(), k=20, which="LR", tol=1e-12,
                                    return_eigenvectors=False)
    want = np.sort(want.real)[::-1]
    np.testing.assert_allclose(vals, want, atol=1e-8)
    assert abs(vecs[:, 0].sum() - 1) < 1e-12


def test_counts_over_resident_labels():
    """cluster -> assigns_to_counts without the labels leaving the device
    (reference flow, transition_matrices.py:113-170): the same COO as counting
    the downloaded labels, for equal and ragged trajectory lengths, both window
    forms; the scratch of the first call is reused by the next"""
    from enspara_amd import synth
    from enspara_amd.device import FrameStore
    from enspara_amd.exception import DataInvalid
    from enspara_amd.msm import assigns_to_counts
    n, A, K = 30000, 12, 150
    x = synth.synth(n, A, 40, seed=8)
    with FrameStore.from_array(x) as st:
        st.reset_state()
        st.kcenters_run(0, K, 0.0)
        _, labels = st.download_state()
        for lengths in ([n // 10] * 10, [1, 7000, 0, 12999, 10000]):
            for lag, sliding in ((1, True), (7, True), (5, False)):
                got = assigns_to_counts(st, lag, max_n_states=K, lengths=lengths,
                                        sliding_window=sliding)
                rows = np.split(labels, np.cumsum(lengths)[:-1])
                want = assigns_to_counts(rows, lag, max_n_states=K,
                                         sliding_window=sliding)
                assert (got.tocsr() != want.tocsr()).nnz == 0
                assert got.sum() == want.sum() > 0
        with pytest.raises(DataInvalid):
            assigns_to_counts(st, 1, max_n_states=K, lengths=[n - 1])
        with pytest.raises(DataInvalid):
            assigns_to_counts(st, 1, lengths=[n])


def test_implied_timescales_match_reference(M):
    from enspara_amd.msm import implied_timescales, builders
    got = implied_timescales(M["assigns"], [int(t) for t in M["implied_lags"]],
                             builders.normalize, n_times=4)
    assert got.shape == M["implied_times"].shape
    np.testing.assert_allclose(got, M["implied_times"], rtol=1e-7)


def test_msm_with_ergodic_trimming(golden_dir):
    """MSM(trim=True).fit against the reference's MSM on the same assignments
    (tests/golden/trim_golden.npz): counts exact, probabilities to 1e-12."""
    from enspara_amd.msm import MSM, TrimMapping
    G = np.load(os.path.join(golden_dir, "trim_golden.npz"))
    m = MSM(lag_time=1, method="normalize", trim=True, max_n_states=40)
    m.fit(G["assigns"])
    np.testing.assert_array_equal(np.asarray(m.tcounts_.todense()),
                                  G["msm_tcounts"])
    np.testing.assert_allclose(np.asarray(m.tprobs_.todense()),
                               G["msm_tprobs"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(m.eq_probs_, G["msm_eq"], rtol=0, atol=1e-10)
    assert m.mapping_ == TrimMapping([(int(o), int(t))
                                      for t, o in G["msm_map"]])
    assert m.n_states_ == 32
    same = MSM(lag_time=1, method="normalize", trim=True, max_n_states=40)
    same.fit(G["assigns"])
    assert m == same
    other = MSM(lag_time=1, method="normalize", trim=False, max_n_states=40)
    assert not (m == other)
    assert other.result_ is None


def test_fitted_msm_save_load(golden_dir, tmp_path):
    """fit on the device -> save -> load gives an equal model (reference
    test_msm_obj.py:60-111), for a trimmed and an untrimmed fit."""
    from enspara_amd.msm import MSM
    G = np.load(os.path.join(golden_dir, "trim_golden.npz"))
    for trim, method in ((True, "normalize"), (False, "transpose")):
        m = MSM(lag_time=1, method=method, trim=trim, max_n_states=40)
        m.fit(G["assigns"])
        d = str(tmp_path / ("msm_%d" % trim))
        m.save(d)
        back = MSM.load(d)
        assert back == m
        assert back.n_states_ == m.n_states_


def test_implied_timescales_with_trimming(golden_dir):
    from enspara_amd.msm import builders, implied_timescales
    G = np.load(os.path.join(golden_dir, "trim_golden.npz"))
    got = implied_timescales(G["assigns"], [int(t) for t in G["implied_lags"]],
                             builders.normalize, n_times=3, trim=True)
    np.testing.assert_allclose(got, G["implied_times_trim"], rtol=1e-8)


_HIST_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from enspara_amd.msm import assigns_to_counts
from enspara_amd import ra
rng = np.random.RandomState(int(sys.argv[2]))
out = []
# (a) a banded walk: few cells per stretch; (b) transitions all over a 300 x 300 table:
# the LDS table overflows and the global path takes over; (c) many short trajectories,
# some empty, some all -1; (d) every lag-th frame instead of the sliding window
K = 700
steps = rng.choice([-2, -1, 0, 1, 2], size=(40, 30000))
A = (rng.randint(K, size=(40, 1)) + np.cumsum(steps, axis=1)) % K
A[rng.rand(*A.shape) < 0.002] = -1
out.append(assigns_to_counts(A, lag_time=1, max_n_states=K))
out.append(assigns_to_counts(A, lag_time=7, max_n_states=K, sliding_window=False))
B = rng.randint(300, size=(6, 100000))
out.append(assigns_to_counts(B, lag_time=3, max_n_states=300))
lens = rng.randint(0, 90, size=4000)
lens[::17] = 0
flat = rng.randint(50, size=int(lens.sum()))
flat[rng.rand(len(flat)) < 0.05] = -1
R = ra.RaggedArray(flat, lengths=lens)
out.append(assigns_to_counts(R, lag_time=2, max_n_states=50))
out.append(assigns_to_counts(R, lag_time=40, max_n_states=50))
np.savez(sys.argv[3], **{"c%d" % i: np.asarray(c.todense()) for i, c in enumerate(out)})
"""


def test_counts_gathered_in_lds_equal_one_atomic_per_transition(tmp_path):
    """round 6: the histogram gathers a stretch of the walk's additions in an LDS hash
    table and adds once per different cell (EK_MSM_HIST_LDS, default) -- against one
    global atomic per transition (= 0: rounds 2-5) and against scipy, on a banded
    walk, on transitions all over the table (the LDS table overflows), on thousands of
    short, empty and all-(-1) trajectories, with and without the sliding window"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for form in ("1", "0"):
        env = dict(os.environ)
        env["EK_MSM_HIST_LDS"] = form
        path = str(tmp_path / ("hist%s.npz" % form))
        p = subprocess.run([sys.executable, "-c", _HIST_CHILD, root, "12", path], env=env,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        got[form] = np.load(path)
    for k in got["1"].files:
        np.testing.assert_array_equal(got["1"][k], got["0"][k])
        assert got["1"][k].sum() > 0
    # and scipy's construction of the first (the reference's own way)
    rng = np.random.RandomState(12)
    K = 700
    steps = rng.choice([-2, -1, 0, 1, 2], size=(40, 30000))
    A = (rng.randint(K, size=(40, 1)) + np.cumsum(steps, axis=1)) % K
    A[rng.rand(*A.shape) < 0.002] = -1
    np.testing.assert_array_equal(got["1"]["c0"], np.asarray(_scipy_counts(A, 1, K).todense()))
