"""GPU parity of the k-centers hot path against the CPU oracle.

Everything here calls the HIP kernels through the C ABI
(enspara_amd/_lib.py -> libenspara_hip.so) and compares with oracle/ on the
same seeded inputs.  Bars: labels / center indices bit-exact; distances
bit-exact as float32 (the contract is stronger than the 1e-5 relative the
north star asks for -- both are asserted).
"""
import numpy as np
import pytest

from enspara_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qcp():
    from oracle import qcp
    return qcp


@pytest.fixture(scope="module")
def ocl():
    from oracle import cluster
    return cluster


def _store(x, **kw):
    from enspara_amd.device import FrameStore
    return FrameStore.from_array(x, **kw)


@pytest.mark.parametrize("n,A", [(1000, 100), (257, 22), (64, 3), (1, 5),
                                 (5000, 301), (300, 4)])
@pytest.mark.parametrize("fpl", [1, 2, 4])
def test_rmsd_one_vs_all_bit_exact(qcp, n, A, fpl):
    x = synth.synth(n, A, max(1, min(10, n)), seed=n + A)
    P = qcp.Prepared(x)
    with _store(x) as st:
        st.set_frames_per_lane(fpl)
        for c in sorted({0, n // 2, n - 1}):
            got = st.rmsd_to_frame(c)
            want = P.rmsd_to_frame(c)
            assert got.dtype == np.float32
            np.testing.assert_array_equal(got, want)
            assert got[c] < 1e-6      # self distance cancels (G matches diag of S)
        y = synth.synth(1, A, 1, seed=99)[0]
        got = st.rmsd_to_xyz(y)
        want = qcp.rmsd(x, y)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_allclose(got, want, rtol=1e-5)


@pytest.mark.parametrize("fpl", [1, 2, 4])
def test_kcenters_nclusters_matches_oracle(ocl, fpl):
    from enspara_amd.cluster import kcenters as kc
    x = synth.synth(3000, 50, 12, seed=7)
    inds, a, d = ocl.kcenters(x, n_clusters=25)
    with _store(x) as st:
        st.set_frames_per_lane(fpl)
        r = kc._kcenters_device(x, 25, 0, None, 0, store=st)
    assert list(r.center_indices) == [int(i) for i in inds]
    np.testing.assert_array_equal(r.assignments, a)
    np.testing.assert_array_equal(r.distances, d)
    assert r.assignments.dtype == np.int64 and r.distances.dtype == np.float64


def test_kcenters_cutoff_matches_oracle(ocl):
    from enspara_amd.cluster.kcenters import kcenters
    x = synth.synth(2000, 40, 8, seed=3)
    for cutoff in (0.9, 0.5, 0.3):
        inds, a, d = ocl.kcenters(x, dist_cutoff=cutoff)
        r = kcenters(x, "rmsd", dist_cutoff=cutoff)
        assert list(r.center_indices) == [int(i) for i in inds]
        np.testing.assert_array_equal(r.assignments, a)
        np.testing.assert_array_equal(r.distances, d)
        assert r.distances.max() <= cutoff


def test_kcenters_both_limits(ocl):
    from enspara_amd.cluster.kcenters import kcenters
    x = synth.synth(1500, 30, 6, seed=5)
    inds, a, d = ocl.kcenters(x, n_clusters=7, dist_cutoff=0.05)
    r = kcenters(x, "rmsd", n_clusters=7, dist_cutoff=0.05)
    assert list(r.center_indices) == [int(i) for i in inds]
    assert len(inds) == 7
    np.testing.assert_array_equal(r.assignments, a)


def test_assign_nearest_matches_oracle(qcp):
    from enspara_amd.cluster.util import assign_to_nearest_center
    x = synth.synth(2500, 35, 9, seed=11)
    ctrs = synth.synth(21, 35, 21, seed=12)
    P = qcp.Prepared(x)
    cc, Gc = qcp.center_and_trace(ctrs)
    wa, wd = qcp.assign_nearest(P.c, P.G, cc, Gc)
    a, d = assign_to_nearest_center(x, [c for c in ctrs], "rmsd")
    np.testing.assert_array_equal(a, wa.astype(np.int64))
    np.testing.assert_array_equal(d, wd.astype(np.float64))


def test_kcenters_warm_start(ocl):
    from enspara_amd.cluster.kcenters import kcenters
    x = synth.synth(1800, 25, 7, seed=21)
    init = [x[5], x[100], x[700]]
    inds, a, d = ocl.kcenters(x, n_clusters=9, init_centers=init)
    r = kcenters(x, "rmsd", n_clusters=9, init_centers=init)
    assert list(r.center_indices) == [int(i) for i in inds]
    np.testing.assert_array_equal(r.assignments, a)
    np.testing.assert_array_equal(r.distances, d)


def test_large_atom_counts(qcp):
    """LDS center tiles beyond the 48 KiB default (assign) and long rows"""
    from enspara_amd.cluster.util import assign_to_nearest_center
    x = synth.synth(300, 900, 4, seed=40)
    ctrs = synth.synth(11, 900, 11, seed=41)
    P = qcp.Prepared(x)
    with _store(x) as st:
        np.testing.assert_array_equal(st.rmsd_to_frame(7), P.rmsd_to_frame(7))
    cc, Gc = qcp.center_and_trace(ctrs)
    wa, wd = qcp.assign_nearest(P.c, P.G, cc, Gc)
    a, d = assign_to_nearest_center(x, [c for c in ctrs], "rmsd")
    np.testing.assert_array_equal(a, wa.astype(np.int64))
    np.testing.assert_array_equal(d, wd.astype(np.float64))


def test_sampled_parity_at_scale(qcp):
    """200k frames x 300 atoms, 64 centers: every center index and a sample
    of 2000 frames (label = arg-min over all centers, distance bit-equal)
    against the checker"""
    from enspara_amd.cluster.kcenters import kcenters
    n, A, K = 200_000, 300, 64
    x = synth.synth(n, A, 500, seed=77)
    r = kcenters(x, "rmsd", n_clusters=K)
    assert len(set(r.center_indices)) == K and r.center_indices[0] == 0
    rng = np.random.RandomState(0)
    sample = np.unique(np.concatenate([rng.randint(0, n, 2000),
                                       np.array(r.center_indices)]))
    cc, Gc = qcp.center_and_trace(x[r.center_indices])
    cs, Gs = qcp.center_and_trace(x[sample])
    wa, wd = qcp.assign_nearest(cs, Gs, cc, Gc)
    np.testing.assert_array_equal(r.assignments[sample], wa)
    np.testing.assert_array_equal(r.distances[sample], wd.astype(np.float64))
    # farthest-point property: each center was at distance >= the final
    # maximum from all earlier centers, and the final max is attained
    assert r.distances.max() == r.distances[np.argmax(r.distances)]
    for k, i in enumerate(r.center_indices):
        assert r.assignments[i] == k


@pytest.mark.parametrize("n,A,K", [(1000, 35, 21), (64, 22, 32), (777, 301, 97),
                                   (2500, 100, 64), (130, 7, 300), (5, 3, 2),
                                   (4100, 50, 333), (300, 13, 1000)])
def test_assign_variants_bit_identical(qcp, n, A, K):
    """vector-FMA and both MFMA nearest-center kernels (32x32x2 on the
    frame-minor tiles, 16x16x4 on the quad copy with the centers in blocks of
    16) against the checker"""
    x = synth.synth(n, A, 6, seed=n + K)
    ctrs = synth.synth(K, A, K, seed=A + K)
    P = qcp.Prepared(x)
    cc, Gc = qcp.center_and_trace(ctrs)
    wa, wd = qcp.assign_nearest(P.c, P.G, cc, Gc)
    with _store(x) as st:
        for variant in (1, 2, 3, 0):
            st.set_option(2, variant)
            st.assign_nearest(ctrs)
            d, a = st.download_state()
            np.testing.assert_array_equal(a, wa)
            np.testing.assert_array_equal(d, wd)
    # duplicated centers: the lower index must win in both kernels
    dup = np.concatenate([ctrs[:3], ctrs[:3], ctrs[3:]])
    with _store(x) as st:
        for variant in (1, 2, 3):
            st.set_option(2, variant)
            st.assign_nearest(dup)
            d, a = st.download_state()
            assert not np.isin(a, [3, 4, 5]).any()


@pytest.mark.parametrize("cands", [1, 4, 8, 16, 32])
@pytest.mark.parametrize("chain", [1, 0])
def test_candidates_per_pass_do_not_change_results(ocl, cands, chain):
    """multi-candidate rounds are the same algorithm: identical centers,
    labels and distances whatever the number of candidates per pass"""
    from enspara_amd.cluster import kcenters as kc
    x = synth.synth(4000, 33, 40, seed=17)
    inds, a, d = ocl.kcenters(x, n_clusters=120)
    with _store(x) as st:
        st.set_option(4, cands)
        st.set_option(5, chain)       # chained or one-by-one cheap steps
        r = kc._kcenters_device(x, 120, 0, None, 0, store=st)
    assert list(r.center_indices) == [int(i) for i in inds]
    np.testing.assert_array_equal(r.assignments, a)
    np.testing.assert_array_equal(r.distances, d)
    # cut-off mode and a continuation from existing labels
    inds, a, d = ocl.kcenters(x, dist_cutoff=0.4)
    with _store(x) as st:
        st.set_option(4, cands)
        st.set_option(5, chain)
        r = kc._kcenters_device(x, np.inf, 0.4, None, 0, store=st)
    assert list(r.center_indices) == [int(i) for i in inds]
    np.testing.assert_array_equal(r.assignments, a)
    np.testing.assert_array_equal(r.distances, d)
    init = [x[9], x[1234]]
    inds, a, d = ocl.kcenters(x, n_clusters=31, init_centers=init)
    with _store(x) as st:
        st.set_option(4, cands)
        st.set_option(5, chain)
        r = kc._kcenters_device(x, 31, 0, init, 0, store=st)
    assert list(r.center_indices) == [int(i) for i in inds]
    np.testing.assert_array_equal(r.assignments, a)


@pytest.mark.parametrize("T", [16, 32])
@pytest.mark.parametrize("A", [1, 2, 3, 5, 7, 13, 16, 17, 31, 47, 61])
def test_rounds_of_16_at_every_atom_count_mod_4(ocl, A, T):
    """the 16-candidate pass takes FOUR atoms per matrix instruction
    (v_mfma_f32_16x16x4_f32) and 16 atoms per candidate load: atom counts that
    leave the last trip, and the last group of 16, partly empty -- the zeros of
    the padding must not change a bit of any sum (A = 1, 2: S has rank one,
    nothing may be abandoned early either).  T = 32: a round is two such
    passes behind one plan and one chain (candidates 16 .. 31 from a second
    candidate tile, sixteen more kept vectors)"""
    from enspara_amd.cluster import kcenters as kc
    x = synth.synth(3000, A, 60, seed=100 + A)
    inds, a, d = ocl.kcenters(x, n_clusters=90)
    with _store(x) as st:
        st.set_option(8, 0)           # no ladder: every round with ...
        st.set_option(4, T)           # ... 16 / 32 candidates
        r = kc._kcenters_device(x, 90, 0, None, 0, store=st)
    assert list(r.center_indices) == [int(i) for i in inds]
    np.testing.assert_array_equal(r.assignments, a)
    np.testing.assert_array_equal(r.distances, d)


def test_duplicate_frames_and_exhaustion(ocl):
    """fewer distinct frames than requested centers: the stop rule
    (max distance 0 is not > 0) ends the run identically"""
    from enspara_amd.cluster.kcenters import kcenters
    base = synth.synth(7, 12, 7, seed=5)
    x = np.concatenate([base] * 40)
    inds, a, d = ocl.kcenters(x, n_clusters=50)
    r = kcenters(x, "rmsd", n_clusters=50)
    assert list(r.center_indices) == [int(i) for i in inds]
    np.testing.assert_array_equal(r.assignments, a)
    np.testing.assert_array_equal(r.distances, d)


@pytest.mark.parametrize("n,A,k", [(1, 5, 3), (3, 4, 10), (9, 7, 9), (257, 3, 40),
                                   (600, 11, 600)])
def test_tiny_and_exhaustive_runs(ocl, n, A, k):
    """fewer frames than candidates per pass, more centers requested than
    frames, every frame becoming a center"""
    from enspara_amd.cluster.kcenters import kcenters
    x = synth.synth(n, A, max(1, n // 2), seed=3 * n + A)
    inds, a, d = ocl.kcenters(x, n_clusters=k)
    r = kcenters(x, "rmsd", n_clusters=k)
    assert list(r.center_indices) == [int(i) for i in inds]
    np.testing.assert_array_equal(r.assignments, a)
    np.testing.assert_array_equal(r.distances, d)


def test_triangle_inequality_is_invisible(ocl):
    """reference test_cluster.py:710-770: kcenters(use_triangle_inequality=True)
    gives what the plain run gives.  Frames in blocks of one template each (a
    trajectory in time order): most tiles are skipped once the templates have
    centers; shuffled frames: hardly any -- same centers, labels, distances."""
    from enspara_amd.cluster.kcenters import kcenters
    from enspara_amd.device import FrameStore
    rng = np.random.RandomState(3)
    A, T, per = 40, 24, 512
    tmpl = synth.templates(T, A, 9)
    x = np.concatenate([tmpl[t] + rng.normal(scale=0.05, size=(per, A, 3))
                        for t in range(T)]).astype(np.float32)
    for order in (np.arange(len(x)), rng.permutation(len(x))):
        xx = np.ascontiguousarray(x[order])
        for kw in (dict(n_clusters=60), dict(n_clusters=np.inf, dist_cutoff=0.35)):
            plain = kcenters(xx, "rmsd", **kw)
            ti = kcenters(xx, "rmsd", use_triangle_inequality=True, **kw)
            want = ocl.kcenters(xx, n_clusters=kw.get("n_clusters"),
                                dist_cutoff=kw.get("dist_cutoff"))
            for r in (plain, ti):
                assert list(r.center_indices) == [int(i) for i in want[0]]
                np.testing.assert_array_equal(r.assignments, want[1])
                np.testing.assert_array_equal(r.distances, want[2])
    # the blocks-of-one-template order really skips tiles (one center per pass:
    # the accounting of round 2 -- a (center, tile) pair per tile and step)
    with FrameStore.from_array(x) as st:
        st.set_option(11, 1)
        st.set_option(4, 1)
        st.reset_state()
        st.kcenters_run(0, 60, 0.0)
        tiles, skipped = st.ti_stats()
    assert tiles == 59 * ((len(x) + 255) // 256)
    assert skipped > 0.5 * tiles


@pytest.mark.parametrize("cands", [16, 32, -1])
def test_triangle_inequality_in_rounds(ocl, cands):
    """round 5: with the option on, rounds of 16 / 32 candidates leave out the
    tiles none of their candidates can change (ek_round_ti_*) instead of falling
    back to one center per pass.  200 templates of 256 consecutive frames each (a
    tile per template): once every template has a center a candidate matters to
    its own tile only.  Same centers, labels, distances as the plain run and the
    oracle -- time-ordered and shuffled, count and cut-off -- and most (tile,
    candidate) pairs are left out on the ordered frames."""
    from enspara_amd.cluster import kcenters as kc
    from enspara_amd.device import FrameStore
    rng = np.random.RandomState(4)
    A, T, per = 20, 200, 256
    tmpl = synth.templates(T, A, 9)
    x = np.concatenate([tmpl[t] + rng.normal(scale=0.05, size=(per, A, 3))
                        for t in range(T)]).astype(np.float32)
    for ordered, order in ((True, np.arange(len(x))), (False, rng.permutation(len(x)))):
        xx = np.ascontiguousarray(x[order])
        for n_clusters, cutoff in ((500, 0.0), (np.inf, 0.17)):
            want = ocl.kcenters(xx, n_clusters=None if np.isinf(n_clusters) else n_clusters,
                                dist_cutoff=cutoff or None)
            for tri in (0, 1):
                with FrameStore.from_array(xx) as st:
                    st.set_option(4, cands)
                    r = kc._kcenters_device(xx, n_clusters, cutoff, None, 0, store=st,
                                            use_triangle_inequality=bool(tri))
                    tiles, skipped = st.ti_stats()
                assert list(r.center_indices) == [int(i) for i in want[0]]
                np.testing.assert_array_equal(r.assignments, want[1])
                np.testing.assert_array_equal(r.distances, want[2])
                if tri and ordered and cutoff == 0.0:
                    assert tiles > 0 and skipped > 0.5 * tiles, (tiles, skipped)
                if not tri:
                    assert tiles == 0


def test_adaptive_run_never_passes_its_goal(ocl):
    """a probe of the one-center form near the end of a run must not accept
    centers past first_label + max_new (round-2 advisor finding): iid
    coordinates are low-yield data (rounds accept ~1 center), so the adaptive
    loop keeps switching forms; every small max_new returns exactly max_new
    centers, equal to the oracle's"""
    from enspara_amd.device import FrameStore
    rng = np.random.RandomState(11)
    x = rng.normal(size=(3000, 12, 3)).astype(np.float32)
    inds, _, _ = ocl.kcenters(x, n_clusters=48)
    with FrameStore.from_array(x) as st:
        st.set_option(8, 1)
        for k in list(range(1, 20)) + [31, 48]:
            st.reset_state()
            idx, cd, _ = st.kcenters_run(0, k, 0.0)
            assert len(idx) == k
            assert [int(i) for i in idx] == [int(i) for i in inds[:k]]
        # continuation in small steps from a warm state
        st.reset_state()
        got = []
        while len(got) < 48:
            step = min(3, 48 - len(got))
            idx, _, _ = st.kcenters_run(len(got), step, 0.0)
            assert len(idx) == step
            got += [int(i) for i in idx]
        assert got == [int(i) for i in inds]


def test_upload_paths_give_the_same_frames(ocl, monkeypatch):
    """ek_load_frames from host memory: many small chunks through both pinned
    buffers (EK_UPLOAD_CHUNK_MB=1, three copy threads), a load in two pieces at
    tile-aligned offsets, and a RE-load of other data into the same context --
    the quad copy of the 16-candidate pass has to follow it."""
    from enspara_amd.cluster import kcenters as kc
    from enspara_amd.device import FrameStore
    n, A = 30000, 41
    x = synth.synth(n, A, 60, seed=5)
    y = synth.synth(n, A, 60, seed=6)
    with FrameStore.from_array(x) as st:
        want = st.rmsd_to_frame(123)
    monkeypatch.setenv("EK_UPLOAD_CHUNK_MB", "1")
    monkeypatch.setenv("EK_UPLOAD_THREADS", "3")
    with FrameStore(n, A) as st:
        st.load(x)
        np.testing.assert_array_equal(st.rmsd_to_frame(123), want)
        # two pieces, the second from a tile-aligned offset
        cut = 256 * 37
        st.load(y[:cut])
        st.load(y[cut:], first=cut)
        inds, a, d = ocl.kcenters(y, n_clusters=70)
        st.set_option(4, 16)
        r = kc._kcenters_device(y, 70, 0, None, 0, store=st)
        assert list(r.center_indices) == [int(i) for i in inds]
        np.testing.assert_array_equal(r.assignments, a)
        np.testing.assert_array_equal(r.distances, d)
        # ... and back to the first data: the quad copy is made again
        st.load(x)
        inds, a, d = ocl.kcenters(x, n_clusters=70)
        r = kc._kcenters_device(x, 70, 0, None, 0, store=st)
        assert list(r.center_indices) == [int(i) for i in inds]
        np.testing.assert_array_equal(r.assignments, a)
        np.testing.assert_array_equal(r.distances, d)


@pytest.mark.parametrize("sweep", [0, 2])
@pytest.mark.parametrize("n,A,K", [(4000, 33, 120), (600, 5, 60), (260, 7, 200),
                                   (70000, 21, 500)])
def test_per_prefix_maxima_in_the_pass_or_in_the_chain_kernel(ocl, sweep, n, A, K):
    """round 6: in a round of 16 the states every prefix of the round's chain would
    leave are reduced to their per-tile arg-max by the pass itself (option
    pass_sweep = 2; 1, the default, does so on shards of up to 524 288 frames) or by
    the chain kernel's sweep over the kept vectors (0: rounds 3-5).  Same centers,
    labels and distances -- also where a round has fewer than four candidates left
    (few frames, many centers: the maxima per 64 frames are written whatever the
    number of candidates), with the last tile partly and some waves wholly empty"""
    from enspara_amd.cluster import kcenters as kc
    x = synth.synth(n, A, 40, seed=n + A)
    inds, a, d = ocl.kcenters(x, n_clusters=K)
    for cands in (16, -1):
        with _store(x) as st:
            st.set_option("candidates", cands)
            st.set_option("pass_sweep", sweep)
            assert st.get_option("pass_sweep") == sweep
            r = kc._kcenters_device(x, K, 0, None, 0, store=st)
        assert list(r.center_indices) == [int(i) for i in inds]
        np.testing.assert_array_equal(r.assignments, a)
        np.testing.assert_array_equal(r.distances, d)
    # the cut-off stop rule (kcenters.py:217) inside a round's chain
    cut = float(d.max()) * 1.2
    inds, a, d = ocl.kcenters(x, dist_cutoff=cut)
    with _store(x) as st:
        st.set_option("candidates", 16)
        st.set_option("pass_sweep", sweep)
        r = kc._kcenters_device(x, np.inf, cut, None, 0, store=st)
    assert list(r.center_indices) == [int(i) for i in inds]
    np.testing.assert_array_equal(r.assignments, a)
    np.testing.assert_array_equal(r.distances, d)
