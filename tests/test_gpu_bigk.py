"""BASELINE.json configs[3]'s CENTER COUNT -- 20 000 -- against the oracle.
Nothing else in the suite runs more than 5000 centers; 20 000 crosses limits of
its own: the candidate pick's label table (EK_PICK_SLOTS = 2048 labels alias),
the PAM window's distance tables T / O [32][K], the triangle-inequality table
[labels][32], the history buffers, the MFMA nearest-center kernel's blocks of
centers.  Shapes chosen so that the oracle's loops (qcp_oracle.c, OpenMP) take
about a minute and a half in all: 40 000 frames x 500 atoms (configs[3]'s atom
count) for the fit + one PAM sweep over the 20 000 medoids + the nearest-center
assignment, 100 000 x 100 for the fit with and without the reference's
``use_triangle_inequality``.  (The whole 10^7 x 500 data set of configs[3]:
tools/c4_one_gpu.py, profiles/r06/.)"""
import os

import numpy as np
import pytest

from enspara_amd import synth

pytestmark = pytest.mark.gpu

N, A, K = 40_000, 500, 20_000


def _threads():
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def _oracle_fit(P, n, k_max):
    """kcenters.py:217-231 / :282-306, fused per iteration (qcp_oracle.c)"""
    dist = np.full(n, np.inf, dtype=np.float32)
    assign = np.full(n, -1, dtype=np.int32)
    centers, cdist, nxt, mx = [], [], 0, np.inf
    for k in range(k_max):
        centers.append(nxt)
        cdist.append(mx)
        mx, nxt = P.kcenters_step(P.c[nxt], P.G[nxt], k, dist, assign)
    return centers, cdist, dist, assign


@pytest.fixture(scope="module")
def world():
    from oracle import qcp
    qcp.set_num_threads(_threads())
    x = synth.synth(N, A, 16000, seed=6)
    P = qcp.Prepared(x)
    _ = P.tiled
    centers, cdist, dist, assign = _oracle_fit(P, N, K)
    yield {"x": x, "P": P, "centers": centers, "cdist": cdist, "dist": dist,
           "assign": assign}
    qcp.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))


def test_fit_to_twenty_thousand_centers(world):
    """kcenters() through the default ladder: every center, label and distance"""
    from enspara_amd.cluster.kcenters import kcenters
    r = kcenters(world["x"], "rmsd", n_clusters=K)
    assert [int(i) for i in r.center_indices] == world["centers"]
    np.testing.assert_array_equal(r.assignments, world["assign"])
    np.testing.assert_array_equal(r.distances.astype(np.float32), world["dist"])
    assert len(r.centers) == K and r.assignments.max() == K - 1


def test_pam_sweep_over_twenty_thousand_medoids(world):
    """one complete sweep (kmedoids.py:575-699): 20 000 proposals, two members
    per cluster on average -- many clusters of one or two frames, where accepting
    is decided by the rounding of numpy's sums"""
    from enspara_amd.cluster import kmedoids as km
    from enspara_amd.device import FrameStore
    from oracle import cluster as oc
    with FrameStore.from_array(world["x"]) as st:
        st.reset_state()
        idx, cd, _ = st.kcenters_run(0, K, 0.0)
        assert [int(i) for i in idx] == world["centers"]
        np.testing.assert_array_equal(cd[1:], np.array(world["cdist"][1:], dtype=np.float32))
        med = km._pam_sweep_device(st, [int(i) for i in idx], None,
                                   np.random.RandomState(5))
        d1, a1 = st.download_state()
    done = []
    want, wd, wa = oc.pam_update(world["P"], world["centers"],
                                 world["assign"].astype(np.int64),
                                 world["dist"].astype(np.float64),
                                 random_state=np.random.RandomState(5), done=done)
    assert done == [K]
    assert [int(m) for m in med] == [int(m) for m in want]
    assert sum(int(a) != int(b) for a, b in zip(med, world["centers"])) > 100
    np.testing.assert_array_equal(a1, wa)
    np.testing.assert_array_equal(np.asarray(d1, dtype=np.float64), wd)


def test_assign_against_twenty_thousand_centers(world):
    """assign_to_nearest_center (util.py:159-205) of the fit's own frames to its
    20 000 centers IS the fit's final state: strict <, the earlier center keeps
    ties.  All three kernels; then KCenters.predict on frames the fit never saw
    against the oracle."""
    from enspara_amd.cluster import util
    from enspara_amd.device import FrameStore
    from oracle import qcp
    x = world["x"]
    ctr = x[world["centers"]]
    a, d = util.assign_to_nearest_center(x, ctr, "rmsd")
    np.testing.assert_array_equal(a, world["assign"])
    np.testing.assert_array_equal(d.astype(np.float32), world["dist"])
    with FrameStore.from_array(x[:6000]) as st:
        for variant in (1, 2, 3):
            st.set_option("assign_kernel", variant)
            st.assign_nearest(ctr)
            dv, av = st.download_state()
            np.testing.assert_array_equal(av, world["assign"][:6000])
            np.testing.assert_array_equal(dv, world["dist"][:6000])
    y = synth.synth(3000, A, 16000, seed=7)
    a2, d2 = util.assign_to_nearest_center(y, ctr, "rmsd")
    cy, Gy = qcp.center_and_trace(y)
    wa2, wd2 = qcp.assign_nearest(cy, Gy, world["P"].c[world["centers"]],
                                  world["P"].G[world["centers"]])
    np.testing.assert_array_equal(a2, wa2)
    np.testing.assert_array_equal(d2.astype(np.float32), wd2)


def test_twenty_thousand_centers_at_100_atoms_with_the_triangle_inequality():
    """100 000 frames x 100 atoms, 20 000 centers: the plain fit against the
    oracle's loop; then the same frames stored cluster by cluster (so that tiles
    of 256 frames can really be left out) with the reference's
    ``use_triangle_inequality`` (kcenters.py:287-296; here a table [labels][32] of
    center-to-candidate distances per round): equal to the plain fit of those
    frames, whose first 2000 centers the oracle replays"""
    from enspara_amd.cluster.kcenters import kcenters
    from enspara_amd.device import FrameStore
    from oracle import qcp
    qcp.set_num_threads(_threads())
    n, a_, k_ = 100_000, 100, 20_000
    x = synth.synth(n, a_, 2500, seed=8)
    P = qcp.Prepared(x)
    centers, _, dist, assign = _oracle_fit(P, n, k_)
    r = kcenters(x, "rmsd", n_clusters=k_)
    assert [int(i) for i in r.center_indices] == centers
    np.testing.assert_array_equal(r.assignments, assign)
    np.testing.assert_array_equal(r.distances.astype(np.float32), dist)
    order = np.argsort(assign, kind="stable")
    xo = np.ascontiguousarray(x[order])
    co, _, _, _ = _oracle_fit(qcp.Prepared(xo), n, 2000)
    with FrameStore.from_array(xo) as st:
        st.reset_state()
        idx0, cd0, _ = st.kcenters_run(0, k_, 0.0)
        d0, a0 = st.download_state()
        st.set_option("triangle", 1)
        st.reset_state()
        idx1, cd1, _ = st.kcenters_run(0, k_, 0.0)
        looked, left_out = st.ti_stats()
        d1, a1 = st.download_state()
    assert [int(i) for i in idx0[:2000]] == co
    np.testing.assert_array_equal(idx1, idx0)
    np.testing.assert_array_equal(cd1, cd0)
    np.testing.assert_array_equal(a1, a0)
    np.testing.assert_array_equal(d1, d0)
    assert looked > 0 and left_out > 0, (looked, left_out)
    qcp.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
