"""Feature-space metrics (the reference's libdist): the oracle against the
reference's compiled module (CPU), the HIP kernels against both (GPU)."""
import os

import numpy as np
import pytest

from oracle import features as of


@pytest.fixture(scope="module")
def F(golden_dir):
    return np.load(os.path.join(golden_dir, "features_golden.npz"))


def test_oracle_matches_reference_libdist(F):
    for name in ("f32", "f64", "i64"):
        X, y = F["X_" + name], F["y_" + name]
        np.testing.assert_array_equal(of.euclidean(X, y), F["euclidean_" + name])
        np.testing.assert_array_equal(of.manhattan(X, y), F["manhattan_" + name])
    np.testing.assert_array_equal(of.hamming(F["X_ham"], F["y_ham"]),
                                  F["hamming"])


@pytest.mark.gpu
def test_device_metrics_match_reference(F):
    from enspara_amd.geometry import libdist
    for name in ("f32", "f64", "i64"):
        X, y = F["X_" + name], F["y_" + name]
        d = libdist.euclidean(X, y)
        assert d.dtype == np.float64 and d.shape == (len(X),)
        np.testing.assert_array_equal(d, F["euclidean_" + name])
        np.testing.assert_array_equal(libdist.manhattan(X, y),
                                      F["manhattan_" + name])
        b = libdist.euclidean.bind(X)
        np.testing.assert_array_equal(b(X, y), F["euclidean_" + name])
        np.testing.assert_array_equal(b(X[:50], y), F["euclidean_" + name][:50])
    np.testing.assert_array_equal(libdist.hamming(F["X_ham"], F["y_ham"]),
                                  F["hamming"])
    out = np.zeros(len(F["X_f32"]))
    r = libdist.manhattan(F["X_f32"], F["y_f32"], out=out)
    assert r is out
    np.testing.assert_array_equal(out, F["manhattan_f32"])


@pytest.mark.gpu
def test_device_metric_validation():
    """reference test_libdist.py:36-89"""
    from enspara_amd.geometry import libdist
    from enspara_amd.exception import DataInvalid
    X = np.array([[1, 1], [2, 2], [3, 3], [-1, 3]])
    y = np.array([0, 0])
    for f in (libdist.euclidean, libdist.manhattan):
        with pytest.raises(DataInvalid):
            f(X, y.reshape(1, -1))
        with pytest.raises(DataInvalid):
            f(X.flatten(), y)
        with pytest.raises(DataInvalid):
            f(X, y[1:])
        with pytest.raises(DataInvalid):
            f(X, y, out=np.zeros(4, dtype=np.float32))
    np.testing.assert_array_equal(libdist.manhattan(X, y), [2, 4, 6, 4])
    np.testing.assert_allclose(libdist.euclidean(X, y),
                               np.sqrt([2, 8, 18, 10]), rtol=0, atol=0)
    for dt in ("int8", "uint16", "int32", "uint64"):
        Xh = np.array([[1, 3, 8], [3, 1, 8], [1, 1, 7]]).astype(dt)
        yh = np.array([1, 2, 3]).astype(dt)
        np.testing.assert_array_equal(libdist.hamming(Xh, yh),
                                      [2 / 3, 1.0, 2 / 3])


@pytest.mark.gpu
def test_feature_clustering_matches_reference(F):
    from enspara_amd.cluster.kcenters import kcenters
    from enspara_amd.cluster.hybrid import hybrid
    X = F["kc_X"]
    r = kcenters(X, "euclidean", n_clusters=6)
    np.testing.assert_array_equal(r.center_indices, F["kc_idx"])
    np.testing.assert_array_equal(r.assignments, F["kc_assign"])
    np.testing.assert_array_equal(r.distances, F["kc_dist"])
    r = hybrid(X, "euclidean", n_clusters=6, n_iters=2,
               random_state=np.random.RandomState(3))
    np.testing.assert_array_equal(r.center_indices, F["hy_idx"])
    np.testing.assert_array_equal(r.assignments, F["hy_assign"])
    np.testing.assert_array_equal(r.distances, F["hy_dist"])


@pytest.mark.gpu
def test_feature_metric_many_features():
    """more features than one LDS chunk; rows not a multiple of the tile"""
    from enspara_amd.geometry import libdist
    rng = np.random.RandomState(2)
    X = rng.normal(size=(1031, 5000)).astype(np.float32)
    y = rng.normal(size=5000).astype(np.float32)
    np.testing.assert_array_equal(libdist.euclidean(X, y), of.euclidean(X, y))
    np.testing.assert_array_equal(libdist.manhattan(X, y), of.manhattan(X, y))


@pytest.mark.gpu
def test_resident_feature_kcenters_equals_the_host_loop():
    """kcenters(X, 'euclidean' | 'manhattan') on an array runs the whole loop
    on the device (ek_feat_kcenters); a wrapped callable of the same metric
    runs the reference-shaped host loop (kcenters.py:217-231, :243-311).  Same
    centers, labels, distances -- float32, float64 and integer features (exact
    ties), count and cut-off stop rules, a warm start, more centers than
    distinct points."""
    from enspara_amd.cluster.kcenters import kcenters
    from enspara_amd.geometry import libdist
    rng = np.random.RandomState(5)
    cases = [
        (rng.normal(size=(3001, 17)).astype(np.float32), dict(n_clusters=40)),
        (rng.normal(size=(2048, 9)), dict(n_clusters=np.inf, dist_cutoff=2.5)),
        (rng.randint(0, 4, size=(1500, 6)).astype(np.int64), dict(n_clusters=70)),
        (rng.randint(0, 3, size=(300, 3)).astype(np.float32),
         dict(n_clusters=60)),                  # 27 distinct points, 60 centers
        (rng.normal(size=(700, 2100)).astype(np.float32), dict(n_clusters=12)),
        (rng.normal(size=(5, 4)), dict(n_clusters=3)),
    ]
    for X, kw in cases:
        for name, fn in (("euclidean", libdist.euclidean),
                         ("manhattan", libdist.manhattan)):
            got = kcenters(X, name, **kw)
            want = kcenters(X, lambda A, y, f=fn: f(A, y), **kw)
            assert list(got.center_indices) == list(want.center_indices), name
            np.testing.assert_array_equal(got.assignments, want.assignments)
            np.testing.assert_array_equal(got.distances, want.distances)
            assert got.distances.dtype == want.distances.dtype
            assert got.assignments.dtype == want.assignments.dtype
    # warm start: init centers, then more
    X = rng.normal(size=(1200, 8)).astype(np.float32)
    init = [X[3], X[700], X[11]]
    got = kcenters(X, "euclidean", n_clusters=15, init_centers=init)
    want = kcenters(X, lambda A, y: libdist.euclidean(A, y), n_clusters=15,
                    init_centers=init)
    assert list(got.center_indices) == list(want.center_indices)
    np.testing.assert_array_equal(got.assignments, want.assignments)
    np.testing.assert_array_equal(got.distances, want.distances)


@pytest.mark.gpu
def test_resident_feature_pam_equals_the_host_loop():
    """A PAM sweep over 'euclidean' / 'manhattan' features runs with distances,
    labels and medoids resident on the device (ek_feat_pam_sweep,
    kmedoids.PAM_FEATURE_DEVICE); with the switch off, the reference-shaped loop
    (kmedoids.py:575-699) around the device metric.  Same medoids, labels and
    float64 distances, and the caller's RandomState left in the same place --
    float32, float64 and integer features (exact ties in distances and in
    costs), random and explicit proposals, two sweeps, a last chunk of its own
    shape in the cost sums, more features than one LDS chunk."""
    from enspara_amd.cluster import kmedoids as km
    from enspara_amd.cluster.kcenters import kcenters
    rng = np.random.RandomState(11)
    cases = [
        (rng.normal(size=(3001, 17)).astype(np.float32), 40),
        (rng.normal(size=(4000, 9)), 25),
        (rng.randint(0, 4, size=(1500, 6)).astype(np.int64), 70),
        (rng.randint(0, 3, size=(400, 3)).astype(np.float32), 20),
        (rng.normal(size=(700, 2100)).astype(np.float32), 12),
        (rng.normal(size=(8200, 5)).astype(np.float32), 60),
    ]
    moved = 0
    for X, K in cases:
        for name in ("euclidean", "manhattan"):
            r = kcenters(X, name, n_clusters=K)
            for explicit in (False, True):
                props = None
                if explicit:
                    props = [int(v) for v in rng.randint(0, len(X), size=K)]
                out = {}
                old = km.PAM_FEATURE_DEVICE
                try:
                    for dev in (1, 0):
                        km.PAM_FEATURE_DEVICE = dev
                        rs = np.random.RandomState(4)
                        inds = [int(i) for i in r.center_indices]
                        d, a = r.distances.copy(), r.assignments.copy()
                        for _ in range(2):
                            inds, d, a, ctrs = km._kmedoids_pam_update(
                                X, name, inds, a, d, proposals=props, random_state=rs)
                        out[dev] = (list(inds), d, a, ctrs, rs.randint(1 << 30, size=3))
                finally:
                    km.PAM_FEATURE_DEVICE = old
                assert out[1][0] == out[0][0], (name, X.shape, explicit)
                np.testing.assert_array_equal(out[1][1], out[0][1])
                np.testing.assert_array_equal(out[1][2], out[0][2])
                assert out[1][1].dtype == out[0][1].dtype
                assert out[1][2].dtype == out[0][2].dtype
                for x, y in zip(out[1][3], out[0][3]):
                    np.testing.assert_array_equal(x, y)
                np.testing.assert_array_equal(out[1][4], out[0][4])
                moved += out[1][0] != [int(i) for i in r.center_indices]
    assert moved >= 12          # (most sweeps accept proposals)
    # the raw random outputs run out again and again: the sweep stops at that
    # cluster, gets more, goes on from there -- same medoids, state and stream
    X, K = cases[0]
    r = kcenters(X, "euclidean", n_clusters=K)
    out = {}
    for ahead in (None, 2):
        monkey = km.FEATURE_RAW_AHEAD
        km.FEATURE_RAW_AHEAD = ahead
        block = km._DrawStream.BLOCK
        km._DrawStream.BLOCK = 4096 if ahead is None else 3
        try:
            rs = np.random.RandomState(9)
            inds, d, a, _ = km._kmedoids_pam_update(
                X, "euclidean", [int(i) for i in r.center_indices],
                r.assignments.copy(), r.distances.copy(), random_state=rs)
            out[ahead] = (list(inds), d, a, rs.randint(1 << 30, size=3))
        finally:
            km.FEATURE_RAW_AHEAD = monkey
            km._DrawStream.BLOCK = block
    assert out[None][0] == out[2][0]
    np.testing.assert_array_equal(out[None][1], out[2][1])
    np.testing.assert_array_equal(out[None][2], out[2][2])
    np.testing.assert_array_equal(out[None][3], out[2][3])
    # an empty cluster: choice([]) raises in both
    X = rng.normal(size=(500, 4)).astype(np.float32)
    r = kcenters(X, "euclidean", n_clusters=10)
    a = r.assignments.copy()
    a[a == 0] = 3
    for dev in (1, 0):
        old = km.PAM_FEATURE_DEVICE
        km.PAM_FEATURE_DEVICE = dev
        try:
            with pytest.raises(ValueError):
                km._kmedoids_pam_update(X, "euclidean", [int(i) for i in r.center_indices],
                                        a.copy(), r.distances.copy(),
                                        random_state=np.random.RandomState(1))
        finally:
            km.PAM_FEATURE_DEVICE = old


@pytest.mark.gpu
def test_feature_pam_windows_equal_one_proposal_at_a_time(monkeypatch):
    """Round 5: a window of proposals takes every sample's distance to each of them
    in one pass (EK_FEAT_PAM_WINDOWS=1 forces the form; by itself the library takes
    it for samples beyond 64 MB).  A drawn proposal's window ends where an accepted
    earlier one moved a sample into or out of its cluster -- noise in few dimensions
    does that every few proposals, tight clusters hardly ever.  Same medoids, labels,
    float64 distances and random stream as one proposal at a time (=0), which the
    test above holds against the reference-shaped loop; also where the raw random
    outputs run out inside a window and where a cluster of a window is empty."""
    from enspara_amd.cluster import kmedoids as km
    from enspara_amd.cluster.kcenters import kcenters
    rng = np.random.RandomState(21)
    tight = (rng.normal(size=(90, 6))[rng.randint(0, 90, size=6000)] +
             0.02 * rng.normal(size=(6000, 6))).astype(np.float32)
    cases = [
        (rng.normal(size=(5000, 4)).astype(np.float32), 100),     # draws stop holding
        (tight, 90),                                              # windows run to their end
        (rng.randint(0, 4, size=(1500, 6)).astype(np.int64), 70), # exact ties
        (rng.normal(size=(3000, 300)), 37),                       # float64, several LDS slices
    ]
    for X, K in cases:
        for name in ("euclidean", "manhattan"):
            r = kcenters(X, name, n_clusters=K)
            for explicit in (False, True):
                props = None
                if explicit:
                    props = [int(v) for v in rng.randint(0, len(X), size=K)]
                out = {}
                for w in ("1", "0"):
                    monkeypatch.setenv("EK_FEAT_PAM_WINDOWS", w)
                    rs = np.random.RandomState(4)
                    inds = [int(i) for i in r.center_indices]
                    d, a = r.distances.copy(), r.assignments.copy()
                    for _ in range(2):
                        inds, d, a, _c = km._kmedoids_pam_update(
                            X, name, inds, a, d, proposals=props, random_state=rs)
                    out[w] = (list(inds), d, a, rs.randint(1 << 30, size=3))
                assert out["1"][0] == out["0"][0], (name, X.shape, explicit)
                np.testing.assert_array_equal(out["1"][1], out["0"][1])
                np.testing.assert_array_equal(out["1"][2], out["0"][2])
                np.testing.assert_array_equal(out["1"][3], out["0"][3])
    # the raw outputs run out inside windows, again and again
    X, K = cases[0]
    r = kcenters(X, "euclidean", n_clusters=K)
    out = {}
    for w, ahead in (("1", 2), ("0", None)):
        monkeypatch.setenv("EK_FEAT_PAM_WINDOWS", w)
        old_ahead, block = km.FEATURE_RAW_AHEAD, km._DrawStream.BLOCK
        km.FEATURE_RAW_AHEAD = ahead
        km._DrawStream.BLOCK = 4096 if ahead is None else 3
        try:
            rs = np.random.RandomState(9)
            inds, d, a, _c = km._kmedoids_pam_update(
                X, "euclidean", [int(i) for i in r.center_indices],
                r.assignments.copy(), r.distances.copy(), random_state=rs)
            out[w] = (list(inds), d, a, rs.randint(1 << 30, size=3))
        finally:
            km.FEATURE_RAW_AHEAD, km._DrawStream.BLOCK = old_ahead, block
    assert out["1"][0] == out["0"][0]
    for i in (1, 2, 3):
        np.testing.assert_array_equal(out["1"][i], out["0"][i])
    # an empty cluster in the middle of a window: choice([]) raises
    monkeypatch.setenv("EK_FEAT_PAM_WINDOWS", "1")
    a = r.assignments.copy()
    a[a == 5] = 3
    with pytest.raises(ValueError):
        km._kmedoids_pam_update(X, "euclidean", [int(i) for i in r.center_indices],
                                a.copy(), r.distances.copy(),
                                random_state=np.random.RandomState(1))


@pytest.mark.gpu
def test_nan_features_keep_the_reference_loop():
    """np.argmax / .max() treat a NaN distance as the maximum and the
    reference's loop (kcenters.py:217, :282) stops on it; the device arg-max
    does not reproduce that, so data with NaN (and dtypes the device does not
    compute in) stays on the reference-shaped host loop (round-2 advisor
    finding): same result as the wrapped callable."""
    from enspara_amd.cluster.kcenters import kcenters
    from enspara_amd.geometry import libdist
    rng = np.random.RandomState(2)
    X = rng.normal(size=(400, 6))
    X[37, 2] = np.nan
    got = kcenters(X, "euclidean", n_clusters=9)
    want = kcenters(X, lambda A, y: libdist.euclidean(A, y), n_clusters=9)
    assert list(got.center_indices) == list(want.center_indices)
    np.testing.assert_array_equal(got.assignments, want.assignments)
    np.testing.assert_array_equal(got.distances, want.distances)


@pytest.mark.gpu
def test_hamming_clustering_on_the_device():
    """libdist.hamming (libdist.pyx:77-95, :187-203) as a clustering metric on
    integer samples: k-centers with the loop resident on the device
    (ek_feat_kcenters, metric 2) and -- round 6 -- the PAM sweep resident too
    (ek_feat_pam_sweep, metric 2: the proposal's distances, the three masks, the
    ambiguous members against all medoids, numpy's cost sums), against the
    reference-shaped host loops around the ORACLE's hamming (oracle/features.py,
    pinned to the reference's compiled module by features_golden.npz).  Distances
    are multiples of 1 / n_features: ties in distances and in costs
    everywhere.  int8 ... uint64 inputs like the reference's fused type."""
    from enspara_amd.cluster import kmedoids as km
    from enspara_amd.cluster.kcenters import kcenters
    from enspara_amd.geometry import libdist
    rng = np.random.RandomState(21)
    cases = [
        (rng.randint(0, 3, size=(2500, 12)).astype(np.int64), 30),
        (rng.randint(0, 2, size=(1800, 40)).astype(np.int8), 25),
        (rng.randint(0, 5, size=(900, 7)).astype(np.uint16), 40),
        (rng.randint(0, 4, size=(3100, 2100)).astype(np.int32), 9),
        (rng.randint(0, 2, size=(8300, 5)).astype(np.uint64), 20),
    ]
    moved = 0
    for X, K in cases:
        got = kcenters(X, libdist.hamming, n_clusters=K)
        want = kcenters(X, lambda A, y: of.hamming(np.asarray(A), np.asarray(y)),
                        n_clusters=K)
        assert list(got.center_indices) == list(want.center_indices), X.dtype
        np.testing.assert_array_equal(got.assignments, want.assignments)
        np.testing.assert_array_equal(got.distances, want.distances)
        for explicit in (False, True):
            props = None
            if explicit:
                props = [int(v) for v in rng.randint(0, len(X), size=K)]
            assert km._feature_sweep_applies(X, libdist.hamming, want.distances, props,
                                             want.assignments)
            out = {}
            for dev in (1, 0):
                rs = np.random.RandomState(4)
                inds = [int(i) for i in want.center_indices]
                d, a = want.distances.copy(), want.assignments.copy()
                metric = libdist.hamming if dev else \
                    (lambda A, y: of.hamming(np.asarray(A), np.asarray(y)))
                for _ in range(2):
                    inds, d, a, ctrs = km._kmedoids_pam_update(
                        X, metric, inds, a, d, proposals=props, random_state=rs)
                out[dev] = (list(inds), d, a, ctrs, rs.randint(1 << 30, size=3))
            assert out[1][0] == out[0][0], (X.dtype, explicit)
            np.testing.assert_array_equal(out[1][1], out[0][1])
            np.testing.assert_array_equal(out[1][2], out[0][2])
            for x, y in zip(out[1][3], out[0][3]):
                np.testing.assert_array_equal(x, y)
            np.testing.assert_array_equal(out[1][4], out[0][4])
            moved += out[1][0] != [int(i) for i in want.center_indices]
    assert moved >= 5
    # floating-point samples: hamming refuses them like the reference's fused type
    with pytest.raises(TypeError):
        libdist.hamming(np.zeros((4, 3), dtype=np.float32), np.zeros(3, dtype=np.float32))
