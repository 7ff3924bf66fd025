"""BASELINE.json configs[1] and configs[2] at their own size, checked COMPLETELY
against the oracle: every one of the 5000 center indices, every frame's label
and float32 distance after the k-centers fit (default ladder of 1 / 8 / 16
candidates per pass), then one complete PAM sweep of 5000 proposals whose first
2000 the oracle replays.  The oracle's loops are qcp_oracle.c's (OpenMP): about
two minutes for the fit and one for the proposals on the GPU box's 16 host
threads; the COMPLETE replay of the sweep (and of the sweep after it) is
bench.py --cpu-seconds 0's, kept under profiles/."""
import os

import numpy as np
import pytest

from enspara_amd import synth

pytestmark = pytest.mark.gpu

N, A, K = 1_000_000, 300, 5000


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


@pytest.fixture(scope="module")
def world():
    from oracle import qcp
    qcp.set_num_threads(_threads())
    x = synth.synth(N, A, 5000, seed=1)         # the bench's data
    P = qcp.Prepared(x)
    _ = P.tiled
    # the oracle's own fit: kcenters.py:217-231 / :282-306, fused per iteration
    dist = np.full(N, np.inf, dtype=np.float32)
    assign = np.full(N, -1, dtype=np.int32)
    centers, nxt = [], 0
    for k in range(K):
        centers.append(nxt)
        _, nxt = P.kcenters_step(P.c[nxt], P.G[nxt], k, dist, assign)
    yield {"x": x, "P": P, "centers": centers, "dist": dist, "assign": assign}
    qcp.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))


def test_whole_fit_at_the_bench_shape(world):
    """all 5000 centers, all 10^6 labels and distances (configs[1])"""
    from enspara_amd.cluster.kcenters import kcenters
    from enspara_amd.device import FrameStore
    r = kcenters(world["x"], "rmsd", n_clusters=K)
    assert [int(i) for i in r.center_indices] == world["centers"]
    np.testing.assert_array_equal(r.assignments, world["assign"])
    np.testing.assert_array_equal(r.distances.astype(np.float32), world["dist"])
    assert r.distances.dtype == np.float64 and r.assignments.dtype == np.int64
    # and the forms the ladder did not spend most of its time in
    with FrameStore.from_array(world["x"]) as st:
        st.set_option(4, 8)
        st.reset_state()
        idx, _, _ = st.kcenters_run(0, 600, 0.0)
        assert [int(i) for i in idx] == world["centers"][:600]


def test_pam_sweep_at_the_bench_shape(world):
    """one complete sweep of 5000 proposals on the device (configs[2]); the oracle
    replays its first 2000 proposals from the same state and random stream --
    medoid for medoid -- and the state the WHOLE sweep leaves is checked on 6000
    sampled frames: every distance is, bit for bit, the RMSD to the medoid the label
    names.  (The complete replay, and the second sweep behind it: bench.py
    --cpu-seconds 0, profiles/r06/bench_full_parity.json -- five minutes of host
    time that round 5 spent inside this suite.)"""
    from enspara_amd.cluster import kmedoids as km
    from enspara_amd.device import FrameStore
    from oracle import cluster as oc
    from oracle import qcp
    with FrameStore.from_array(world["x"]) as st:
        st.reset_state()
        idx, _, _ = st.kcenters_run(0, K, 0.0)
        assert [int(i) for i in idx] == world["centers"]
        med = km._pam_sweep_device(st, [int(i) for i in idx], None,
                                   np.random.RandomState(1))
        restricted, full = st.pam_prefetch_passes()
        hits, misses = st.pam_prefetch_stats()
        d1, a1 = st.download_state()
    # (the restricted prefetch -- exact distances only for the frames a
    # proposal can touch -- is what served every proposal)
    assert restricted > 0 and full == 0 and hits == K and misses == 0
    done, first = [], 2000
    want, _, _ = oc.pam_update(world["P"], world["centers"],
                               world["assign"].astype(np.int64),
                               world["dist"].astype(np.float64),
                               random_state=np.random.RandomState(1), done=done,
                               stop_after=first)
    assert done == [first]
    assert [int(m) for m in med[:first]] == [int(m) for m in want[:first]]
    assert sum(int(a) != int(b) for a, b in zip(med, world["centers"])) > K // 4
    P = world["P"]
    pick = np.random.RandomState(3).choice(N, size=6000, replace=False)
    assert a1.min() >= 0 and a1.max() < K
    for lab in np.unique(a1[pick]):
        fr = pick[a1[pick] == lab]
        m = int(med[lab])
        np.testing.assert_array_equal(
            qcp.rmsd_centered(np.ascontiguousarray(P.c[fr]), np.ascontiguousarray(P.G[fr]),
                              P.c[m], float(P.G[m])), d1[fr])
    # every medoid sits at (numerically) zero distance under its own label
    assert np.all(d1[[int(m) for m in med]] < 1e-3)
    assert np.array_equal(a1[[int(m) for m in med]], np.arange(K))


def test_rounds_at_500_atoms():
    """BASELINE.json configs[3]'s atom count on a slice of its per-GPU shard --
    300 000 frames x 500 atoms, 600 centers through the default ladder (rounds of
    16 on the quad copy: 125 trips of 4 atoms, 7 groups of 16 + 13 atoms for the
    candidates' broadcast layout) -- every center, label and distance against
    the oracle's loop"""
    from enspara_amd.cluster.kcenters import kcenters
    from oracle import qcp
    qcp.set_num_threads(_threads())
    n, A5, K5 = 300_000, 500, 600
    x = synth.synth(n, A5, 2000, seed=3)
    P = qcp.Prepared(x)
    dist = np.full(n, np.inf, dtype=np.float32)
    assign = np.full(n, -1, dtype=np.int32)
    centers, nxt = [], 0
    for k in range(K5):
        centers.append(nxt)
        _, nxt = P.kcenters_step(P.c[nxt], P.G[nxt], k, dist, assign)
    r = kcenters(x, "rmsd", n_clusters=K5)
    assert [int(i) for i in r.center_indices] == centers
    np.testing.assert_array_equal(r.assignments, assign)
    np.testing.assert_array_equal(r.distances.astype(np.float32), dist)
    qcp.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
