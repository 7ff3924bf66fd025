"""Host-side surface that needs no GPU: argument handling and exceptions
(reference kcenters.py:58-73,177-193; hybrid.py:64-70; util.py:69-72,
289-313), the callable-metric loops (the reference's plug-in contract,
kcenters.py:132-137), center lookup and partitioning."""
import numpy as np
import pytest

from enspara_amd.cluster import KCenters, KHybrid, KMedoids, util
from enspara_amd.cluster.hybrid import hybrid
from enspara_amd.cluster.kcenters import kcenters
from enspara_amd.cluster.kmedoids import _kmedoids_pam_update, kmedoids
from enspara_amd.exception import DataInvalid, ImproperlyConfigured


def euc(X, y):
    return np.sqrt(np.square(X - y).sum(axis=1))


@pytest.fixture(scope="module")
def blobs():
    rng = np.random.RandomState(0)
    cs = rng.uniform(-20, 20, size=(5, 3))
    return np.concatenate([c + rng.normal(size=(60, 3)) for c in cs])


def test_configuration_errors():
    with pytest.raises(ImproperlyConfigured):
        KCenters(metric=euc)
    with pytest.raises(ImproperlyConfigured):
        KHybrid(metric=euc, kmedoids_updates=10)
    with pytest.raises(ImproperlyConfigured):
        KCenters(metric="not-a-metric", n_clusters=3)
    with pytest.raises(ImproperlyConfigured):
        kcenters(np.zeros((4, 2)), euc)              # neither limit given
    with pytest.raises(ImproperlyConfigured):
        kcenters(np.zeros((4, 2)), euc, n_clusters=None, dist_cutoff=None)
    with pytest.raises(NotImplementedError):
        kcenters(np.zeros((4, 2)), euc, n_clusters=2, random_first_center=True)
    with pytest.raises(ImproperlyConfigured):
        KCenters(metric=euc, n_clusters=2).predict(np.zeros((4, 2)))
    with pytest.raises(ImproperlyConfigured):
        kmedoids(np.zeros((4, 2)), euc)
    assert util._get_distance_method("rmsd") is util.rmsd
    assert util._get_distance_method(euc) is euc


def test_kcenters_callable_metric(blobs):
    r = kcenters(blobs, euc, n_clusters=5)
    assert r.center_indices[0] == 0                 # frame 0 is always first
    assert len(np.unique(r.assignments)) == 5
    assert r.assignments.dtype == np.int64 and r.distances.dtype == np.float64
    # brute force
    D = np.stack([euc(blobs, blobs[i]) for i in r.center_indices])
    np.testing.assert_array_equal(r.assignments, D.argmin(0))
    np.testing.assert_allclose(r.distances, D.min(0))
    # each new center was the farthest point at its time
    d = np.full(len(blobs), np.inf)
    for i in r.center_indices:
        assert i == int(np.argmax(d))
        d = np.minimum(d, euc(blobs, blobs[i]))
    r2 = kcenters(blobs, euc, dist_cutoff=4.0)
    assert r2.distances.max() <= 4.0
    r3 = kcenters(blobs, euc, n_clusters=3, dist_cutoff=0.1)
    assert len(r3.center_indices) == 3


def test_triangle_inequality_equals_plain(blobs):
    """reference test_cluster.py:710-770"""
    a = kcenters(blobs, euc, n_clusters=12)
    b = kcenters(blobs, euc, n_clusters=12, use_triangle_inequality=True)
    assert a.center_indices == b.center_indices
    np.testing.assert_array_equal(a.assignments, b.assignments)
    np.testing.assert_allclose(a.distances, b.distances)


def test_assign_equals_bruteforce(blobs):
    """reference test_cluster_util.py:88-123"""
    centers = blobs[[3, 77, 140, 200, 290]]
    a, d = util.assign_to_nearest_center(blobs, centers, euc)
    D = np.stack([euc(blobs, c) for c in centers])
    np.testing.assert_array_equal(a, D.argmin(0))
    np.testing.assert_allclose(d, D.min(0))
    # ties go to the lower center index
    a2, _ = util.assign_to_nearest_center(blobs, [centers[0], centers[0]], euc)
    assert not a2.any()


def test_find_cluster_centers():
    a = np.array([2, 0, 2, 0, 5, 5, 5])
    d = np.array([.3, .2, .1, .2, .9, .4, .4])
    np.testing.assert_array_equal(util.find_cluster_centers(a, d), [1, 2, 5])
    with pytest.raises(DataInvalid):
        util.find_cluster_centers(a, d[:3])
    rng = np.random.RandomState(1)
    a = rng.randint(0, 40, 5000)
    d = rng.rand(5000).round(2)
    want = [np.where(a == c)[0][np.argmin(d[a == c])] for c in np.unique(a)]
    np.testing.assert_array_equal(util.find_cluster_centers(a, d), want)


def test_hybrid_and_pam_callable_metric(blobs):
    r0 = kcenters(blobs, euc, n_clusters=5)
    r = hybrid(blobs, euc, n_clusters=5, n_iters=3, random_state=0)
    assert np.square(r.distances).mean() <= np.square(r0.distances).mean()
    for k, i in enumerate(r.center_indices):
        assert r.assignments[i] == k and r.distances[i] == 0
    # a proposal that is the current medoid changes nothing
    inds = list(r.center_indices)
    mi, d, a, _ = _kmedoids_pam_update(blobs, euc, list(inds),
                                       r.assignments.copy(),
                                       r.distances.copy(), proposals=inds)
    assert mi == inds
    np.testing.assert_array_equal(a, r.assignments)
    with pytest.raises(DataInvalid):
        _kmedoids_pam_update(blobs, euc, list(inds), r.assignments,
                             r.distances, proposals=inds[:2])
    est = KHybrid(euc, n_clusters=5, kmedoids_updates=2, random_state=1)
    est.fit(blobs)
    assert est.labels_.shape == (300,) and est.runtime_ > 0
    p = est.predict(blobs[:50])
    np.testing.assert_array_equal(p.assignments, est.labels_[:50])
    km = KMedoids(euc, n_clusters=5, n_iters=2).fit(blobs)
    assert len(km.center_indices_) == 5


def test_partition():
    res = util.ClusterResult(center_indices=[0, 7, 12],
                             assignments=np.arange(15) % 3,
                             distances=np.arange(15) / 10., centers=[])
    p = res.partition([5, 5, 5])
    assert isinstance(p.assignments, np.ndarray) and p.assignments.shape == (3, 5)
    assert p.center_indices == [(0, 0), (1, 2), (2, 2)]
    p = res.partition([4, 11])
    assert p.center_indices == [(0, 0), (1, 3), (1, 8)]
    np.testing.assert_array_equal(p.distances[1], np.arange(4, 15) / 10.)
    with pytest.raises(DataInvalid):
        res.partition([5, 11])


def test_draw_stream_is_randomstate_choice():
    """kmedoids._DrawStream: every draw is what RandomState.choice(m) returns
    (masked rejection on 32-bit Mersenne Twister outputs, m == 1 consumes
    nothing), look-ahead sees the same numbers, and close() leaves the caller's
    RandomState where that many choice() calls would have (reference
    kmedoids.py:514 draws proposals with choice(state_inds))."""
    from enspara_amd.cluster.kmedoids import _DrawStream
    rng = np.random.default_rng(1)
    for trial in range(60):
        seed = int(rng.integers(1 << 30))
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        b.normal()
        a.normal()                  # a cached gaussian must survive as well
        st = _DrawStream(a)
        ms = [int(x) for x in rng.integers(1, 10 ** int(rng.integers(1, 10)),
                                           size=int(rng.integers(1, 200)))]
        if trial % 5 == 0:
            ms[0] = 1
        ahead = st.peek(ms[:8])
        got = [st.draw(m) for m in ms]
        want = [int(b.choice(m)) for m in ms]
        assert got == want and ahead == want[:len(ahead)]
        st.close()
        assert a.normal() == b.normal()
        assert int(a.randint(0, 10 ** 9)) == int(b.randint(0, 10 ** 9))
    st = _DrawStream(np.random.RandomState(0))
    assert st.peek([5, 0, 7]) == st.peek([5])       # stops before an empty list
    with pytest.raises(ValueError):
        st.draw(0)


def test_library_draws_are_randomstate_choice():
    """The draws ek_pam_sweep makes inside the library (ek_np_choice_draws: the
    same routine, host only) against numpy itself: RandomState.choice(m) for
    list lengths from 1 to 2**32, the number of raw outputs consumed, a stream
    that runs out in the middle of a rejection loop, an empty list."""
    import ctypes as C
    from enspara_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7)
    for trial in range(40):
        seed = int(rng.integers(1 << 30))
        ref = np.random.RandomState(seed)
        raw = np.random.RandomState(seed).randint(0, 2 ** 32, size=4096, dtype=np.uint32)
        ms = rng.integers(1, 10 ** int(rng.integers(1, 10)), size=int(rng.integers(1, 300)))
        ms = np.ascontiguousarray(ms, dtype=np.int64)
        if trial % 4 == 0:
            ms[::7] = 1                         # consumes nothing
        if trial % 9 == 0:
            ms[-1] = 2 ** 32                    # the widest mask
        out = np.zeros(len(ms), dtype=np.int64)
        pos = C.c_int64(0)
        made = lib.ek_np_choice_draws(raw.ctypes.data_as(C.POINTER(C.c_uint32)), len(raw),
                                      C.byref(pos), _lib.i64p(ms), len(ms), _lib.i64p(out))
        assert made == len(ms)
        want = [int(ref.choice(int(m))) for m in ms]
        assert [int(v) for v in out] == want
        # the RandomState is where `pos` raw outputs would have left it
        check = np.random.RandomState(seed)
        if pos.value:
            check.randint(0, 2 ** 32, size=pos.value, dtype=np.uint32)
        assert int(check.randint(0, 10 ** 9)) == int(ref.randint(0, 10 ** 9))
    # the outputs run out: nothing of the unfinished draw is consumed
    raw = np.random.RandomState(3).randint(0, 2 ** 32, size=5, dtype=np.uint32)
    ms = np.full(50, 3 * 10 ** 8, dtype=np.int64)
    out = np.zeros(50, dtype=np.int64)
    pos = C.c_int64(0)
    made = lib.ek_np_choice_draws(raw.ctypes.data_as(C.POINTER(C.c_uint32)), len(raw),
                                  C.byref(pos), _lib.i64p(ms), 50, _lib.i64p(out))
    ref = np.random.RandomState(3)
    assert 0 < made < 50 and pos.value <= 5
    assert [int(v) for v in out[:made]] == [int(ref.choice(3 * 10 ** 8)) for _ in range(made)]
    # an empty list ends the draws (RandomState.choice(0) raises)
    ms = np.array([4, 0, 4], dtype=np.int64)
    pos = C.c_int64(0)
    raw = np.random.RandomState(1).randint(0, 2 ** 32, size=64, dtype=np.uint32)
    assert lib.ek_np_choice_draws(raw.ctypes.data_as(C.POINTER(C.c_uint32)), 64, C.byref(pos),
                                  _lib.i64p(ms), 3, _lib.i64p(out)) == 1


def test_feature_sweep_on_the_device_only_where_it_is_the_same_computation():
    """kmedoids._feature_sweep_applies: the resident PAM sweep takes 'euclidean' /
    'manhattan' over a real matrix without NaN and a finite state; everything
    else (other metrics, callables, NaN or infinite data, objects that are not
    arrays) keeps the reference-shaped loop."""
    from enspara_amd.cluster import kmedoids as km
    from enspara_amd.geometry import libdist
    X = np.random.RandomState(0).normal(size=(50, 3)).astype(np.float32)
    d = np.ones(50)
    assert km._feature_sweep_applies(X, libdist.euclidean, d, None)
    assert km._feature_sweep_applies(X.astype(np.float64), libdist.manhattan, d, None)
    assert km._feature_sweep_applies((X * 10).astype(np.int64), libdist.euclidean, d, None)
    assert not km._feature_sweep_applies(X, libdist.hamming, d, None)     # (float samples)
    assert km._feature_sweep_applies((X * 10).astype(np.int64), libdist.hamming, d, None)
    assert not km._feature_sweep_applies(X, lambda A, y: libdist.euclidean(A, y), d, None)
    assert not km._feature_sweep_applies(X.tolist(), libdist.euclidean, d, None)
    assert not km._feature_sweep_applies(X[0], libdist.euclidean, d, None)
    bad = X.copy()
    bad[3, 1] = np.nan
    assert not km._feature_sweep_applies(bad, libdist.euclidean, d, None)
    dinf = d.copy()
    dinf[7] = np.inf
    assert not km._feature_sweep_applies(X, libdist.euclidean, dinf, None)
    assert not km._feature_sweep_applies(X.astype(np.complex64), libdist.euclidean, d, None)
    # a float32 state: the reference rounds every accepted proposal's distances
    # to float32 (zeros_like(distances), kmedoids.py:639) -- not the resident
    # sweep's float64 arithmetic; labels that are not integers neither
    a = np.zeros(50, dtype=np.int64)
    assert km._feature_sweep_applies(X, libdist.euclidean, d, None, a)
    assert not km._feature_sweep_applies(X, libdist.euclidean,
                                         d.astype(np.float32), None, a)
    assert not km._feature_sweep_applies(X, libdist.euclidean, d, None,
                                         a.astype(np.float64))


def test_a_bound_metric_is_bound_once():
    """libdist.Bound.bind: the loops bind on entry (kmedoids.py once per sweep,
    kcenters.py once per fit); a metric already bound to the same array is
    handed on, not uploaded again -- checked here without a device"""
    from enspara_amd.geometry import libdist
    X = np.zeros((0, 3), dtype=np.float32)      # (no rows: nothing is uploaded)
    b = libdist.euclidean.bind(X)
    assert b.device_metric_id == 0 and b.bind(X) is b
    Y = np.zeros((0, 3), dtype=np.float32)
    assert b.bind(Y) is not b and b.bind(Y).X is Y
