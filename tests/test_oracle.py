"""Pin the CPU oracle (oracle/) -- no GPU needed.

1. RMSD arithmetic against the reference's own known answers: the statistics
   enspara/test/test_cluster.py:209-218 and :231-238 assert for k-centers on
   test/data/frame0.xtc were produced by the real mdtraj.rmsd.  The fixture is
   decoded here by oracle/xtc.py (tests/golden/frame0.xtc is that data file).
2. RMSD arithmetic against an independent float64 SVD/Kabsch implementation.
3. The oracle's control flow (oracle/cluster.py) against outputs of the real
   reference functions (tests/golden/cluster_golden.npz, made by
   tests/golden/make_golden.py).
"""
import os

import numpy as np
import pytest

from oracle import cluster as oc
from oracle import qcp, xtc
from enspara_amd import synth


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "cluster_golden.npz"))


def kabsch(X, y):
    X = X.astype(np.float64)
    y = y.astype(np.float64)
    Xc = X - X.mean(1, keepdims=True)
    yc = y - y.mean(0)
    H = np.einsum("nai,aj->nij", Xc, yc)
    U, s, Vt = np.linalg.svd(H)
    s[:, -1] *= np.sign(np.linalg.det(U) * np.linalg.det(Vt))
    msd = ((Xc ** 2).sum((1, 2)) + (yc ** 2).sum() - 2 * s.sum(1)) / X.shape[1]
    return np.sqrt(np.maximum(msd, 0))


def test_xtc_decoder_reproduces_fixture(golden_dir, G):
    d = xtc.read_xtc(os.path.join(golden_dir, "frame0.xtc"))
    assert d["xyz"].shape == (501, 22, 3)
    np.testing.assert_array_equal(d["xyz"], G["frame0_xyz"])
    # first atom of native.pdb (same system): 4.300 13.100 8.600 Angstrom
    np.testing.assert_allclose(d["xyz"][0, 0], [0.43, 1.31, 0.86], atol=1e-6)


def test_reference_known_answer_k3(G):
    """enspara/test/test_cluster.py:221-238 (assertAlmostEqual, 7 places)"""
    x = G["frame0_xyz"]
    inds, a, d = oc.kcenters(x, n_clusters=3)
    mean_ref, std_ref = G["frame0_ref_k3_mean_std"]
    assert len(np.unique(a)) == 3
    assert round(abs(np.average(d) - mean_ref), 7) == 0
    assert round(abs(np.std(d) - std_ref), 7) == 0


def test_reference_known_answer_cutoff(G):
    """enspara/test/test_cluster.py:199-218 (17 clusters; 5 places)"""
    x = G["frame0_xyz"]
    inds, a, d = oc.kcenters(x, dist_cutoff=0.1)
    n_ref, mean_ref, std_ref = G["frame0_ref_cut_n_mean_std"]
    assert len(np.unique(a)) == int(n_ref)
    assert round(abs(np.average(d) - mean_ref), 5) == 0
    assert round(abs(np.std(d) - std_ref), 5) == 0
    assert d.max() < 0.1


@pytest.mark.parametrize("n,A,seed", [(400, 22, 0), (300, 100, 1),
                                      (200, 301, 2), (50, 3, 3)])
def test_rmsd_vs_float64_kabsch(n, A, seed):
    x = synth.synth(n, A, 7, seed=seed)
    P = qcp.Prepared(x)
    for c in (0, n // 3, n - 1):
        got = P.rmsd_to_frame(c).astype(np.float64)
        want = kabsch(x, x[c])
        scale = np.sqrt((P.G + P.G[c]) / A)     # natural scale of the msd
        assert np.all(np.abs(got - want) <= 1e-5 * scale + 1e-7)
        assert got[c] < 1e-6


def test_rmsd_invariant_under_rigid_motion():
    x = synth.synth(64, 40, 4, seed=9)
    rng = np.random.RandomState(0)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    R = synth._quat_to_rot(q[None])[0]
    y = (x @ R.T + rng.uniform(-3, 3, size=3)).astype(np.float32)
    d0 = qcp.rmsd(x, x[5]).astype(np.float64)
    d1 = qcp.rmsd(y, x[5]).astype(np.float64)
    # msd = (Gx + Gy - 2 lambda) / A cancels: the error scale is (Gx+Gy)/A, so
    # compare squared distances (a rotated copy of the center itself comes out
    # at ~4e-4 nm instead of 0, like mdtraj's ~1e-4 self distances that
    # enspara tolerates at kmedoids.py:186,197)
    P = qcp.Prepared(x)
    scale = (P.G + P.G[5]) / x.shape[1]
    assert np.all(np.abs(d0 ** 2 - d1 ** 2) <= 1e-5 * scale)


def test_tiled_equals_scalar_bitwise():
    x = synth.synth(1000, 37, 5, seed=2)
    P = qcp.Prepared(x)
    for c in (0, 511, 999):
        np.testing.assert_array_equal(P.rmsd_to_frame(c),
                                      P.rmsd_to_frame_tiled(c))


def test_fused_step_equals_separate_passes():
    x = synth.synth(1500, 20, 6, seed=4)
    P = qcp.Prepared(x)
    dist = np.full(P.n, np.inf, dtype=np.float32)
    assign = np.full(P.n, -1, dtype=np.int32)
    d2 = dist.copy()
    a2 = assign.copy()
    nxt = 0
    for label in range(12):
        c = nxt
        mx, nxt = P.kcenters_step(P.c[c], P.G[c], label, dist, assign)
        nd = P.rmsd_to_frame(c)
        upd = nd < d2
        d2[upd] = nd[upd]
        a2[upd] = label
        assert nxt == int(np.argmax(d2)) and mx == d2.max()
    np.testing.assert_array_equal(dist, d2)
    np.testing.assert_array_equal(assign, a2)


# ---- control flow against the real reference's outputs ---------------------
def _X(args):
    n, A, T, seed = [int(v) for v in args[:4]]
    return synth.synth(n, A, T, seed=seed)


def test_kcenters_matches_reference(G):
    X = _X(G["kc_X_args"])
    i, a, d = oc.kcenters(X, n_clusters=20)
    np.testing.assert_array_equal(i, G["kc_n20_idx"])
    np.testing.assert_array_equal(a, G["kc_n20_assign"])
    np.testing.assert_array_equal(d, G["kc_n20_dist"])
    i, a, d = oc.kcenters(X, dist_cutoff=float(G["kc_cut_cutoff"]))
    np.testing.assert_array_equal(i, G["kc_cut_idx"])
    np.testing.assert_array_equal(a, G["kc_cut_assign"])
    np.testing.assert_array_equal(d, G["kc_cut_dist"])
    i, a, d = oc.kcenters(X, n_clusters=6, dist_cutoff=0.05)
    np.testing.assert_array_equal(i, G["kc_both_idx"])
    np.testing.assert_array_equal(a, G["kc_both_assign"])
    init = [X[j] for j in G["kc_init_frames"]]
    i, a, d = oc.kcenters(X, n_clusters=9, init_centers=init)
    np.testing.assert_array_equal(i, G["kc_init_idx"])
    np.testing.assert_array_equal(a, G["kc_init_assign"])
    np.testing.assert_array_equal(d, G["kc_init_dist"])


def test_config1_matches_reference(G):
    """BASELINE.json configs[0]: 1k frames x 100 atoms, 10 clusters"""
    X = _X(G["c1_args"])
    i, a, d = oc.kcenters(X, n_clusters=10)
    np.testing.assert_array_equal(i, G["c1_idx"])
    np.testing.assert_array_equal(a, G["c1_assign"])
    np.testing.assert_array_equal(d, G["c1_dist"])


def test_assign_matches_reference(G):
    X = _X(G["kc_X_args"])
    n, A, T, seed, k = [int(v) for v in G["asg_centers_args"]]
    Y = synth.synth(n, A, T, seed=seed)
    a, d = oc.assign_to_nearest_center(X, [c for c in Y[:k]])
    np.testing.assert_array_equal(a, G["asg_assign"])
    np.testing.assert_array_equal(d, G["asg_dist"])
    np.testing.assert_array_equal(oc.find_cluster_centers(a, d), G["asg_fcc"])
    a2, d2 = oc.assign_to_nearest_center(X[:40], [c for c in Y[:120]])
    np.testing.assert_array_equal(a2, G["asg2_assign"])
    np.testing.assert_array_equal(d2, G["asg2_dist"])
    # C implementation of the same thing
    P = qcp.Prepared(X)
    cc, Gc = qcp.center_and_trace(Y[:k])
    a3, d3 = qcp.assign_nearest(P.c, P.G, cc, Gc)
    np.testing.assert_array_equal(a3, G["asg_assign"])
    np.testing.assert_array_equal(d3.astype(np.float64), G["asg_dist"])


def test_hybrid_matches_reference(G):
    X = _X(G["hy_X_args"])
    i, a, d = oc.hybrid(X, n_clusters=12, n_iters=2,
                        random_state=np.random.RandomState(0))
    np.testing.assert_array_equal(i, G["hy_rs_idx"])
    np.testing.assert_array_equal(a, G["hy_rs_assign"])
    np.testing.assert_array_equal(d, G["hy_rs_dist"])
    i, a, d = oc.hybrid(X, n_clusters=12, n_iters=2, random_state=0)
    np.testing.assert_array_equal(i, G["hy_int_idx"])
    np.testing.assert_array_equal(a, G["hy_int_assign"])
    np.testing.assert_array_equal(d, G["hy_int_dist"])
    i, a, d = oc.hybrid(X, n_clusters=12, n_iters=0)
    np.testing.assert_array_equal(i, G["hy_0_idx"])
    np.testing.assert_array_equal(a, G["hy_0_assign"])


def test_pam_with_proposals_matches_reference(G):
    X = _X(G["hy_X_args"])
    i, a, d = oc.kcenters(X, n_clusters=8)
    mi, dd, aa = oc.pam_update(X, i, a, d, proposals=G["pam_props"],
                               random_state=0)
    np.testing.assert_array_equal(mi, G["pam_idx"])
    np.testing.assert_array_equal(aa, G["pam_assign"])
    np.testing.assert_array_equal(dd, G["pam_dist"])


# --- two more mdtraj-produced known answers the reference's tests hold ----------
# (integer-exact: medoid indices after one PAM sweep on test/data/frame0.h5,
# decoded here by enspara_amd.h5lite through util.load)

def _frame0_h5(golden_dir):
    from enspara_amd.util.load import load_as_concatenated
    lengths, x = load_as_concatenated([os.path.join(golden_dir, "frame0.h5")])
    assert list(lengths) == [501] and x.shape == (501, 22, 3)
    return x


def test_reference_known_answer_pam_k3(golden_dir):
    """enspara/test/test_cluster.py:533-554: k-centers with 3 clusters, one PAM
    sweep with random_state=0 -> medoids [298, 44, 341]; the state it leaves is
    the nearest-medoid assignment."""
    x = _frame0_h5(golden_dir)
    inds, a, d = oc.kcenters(x, n_clusters=3)
    mi, dd, aa = oc.pam_update(x, inds, a, d, random_state=0)
    assert [int(i) for i in mi] == [298, 44, 341]
    ea, ed = oc.assign_to_nearest_center(x, [x[i] for i in mi])
    np.testing.assert_array_equal(np.unique(aa), np.arange(3))
    np.testing.assert_array_equal(aa, ea)
    np.testing.assert_allclose(dd, ed, atol=1e-6)


def test_reference_known_answer_pam_k10_first_members(golden_dir):
    """enspara/test/test_cluster.py:379-412 at MPI size 1: 10 clusters, every
    cluster proposes its first member -> medoids
    [0, 37, 400, 105, 12, 327, 242, 346, 42, 3]."""
    x = _frame0_h5(golden_dir)
    inds, a, d = oc.kcenters(x, n_clusters=10)
    props = [int(np.where(a == c)[0][0]) for c in range(10)]
    mi, dd, aa = oc.pam_update(x, inds, a, d, proposals=props, random_state=0)
    assert [int(i) for i in mi] == [0, 37, 400, 105, 12, 327, 242, 346, 42, 3]


@pytest.mark.parametrize("n,A,K,T,seed", [(3000, 25, 40, 30, 1),
                                          (5000, 10, 200, 50, 2),
                                          (2000, 3, 30, 5, 3),
                                          (700, 22, 400, 10, 4)])
def test_pam_trial_in_c_equals_the_numpy_loop(n, A, K, T, seed):
    """oracle.cluster.pam_update builds every proposal's trial state in
    qcp_oracle.c (eko_pam_trial); pam_update_numpy is the line-by-line numpy
    form of kmedoids.py:637-683 that the goldens of the real reference were
    first pinned with.  Same medoids, labels, distances -- random and explicit
    proposals, a RandomState carried over two sweeps and an int seed."""
    x = synth.synth(n, A, T, seed=seed)
    P = qcp.Prepared(x)
    inds, a, d = oc.kcenters(P, n_clusters=K)
    for rs in (np.random.RandomState(seed), 7):
        rs2 = np.random.RandomState(seed) if not isinstance(rs, int) else rs
        r1 = oc.pam_update(P, inds, a.copy(), d.copy(), random_state=rs)
        r2 = oc.pam_update_numpy(P, inds, a.copy(), d.copy(), random_state=rs2)
        for _ in range(2):
            assert [int(i) for i in r1[0]] == [int(i) for i in r2[0]]
            np.testing.assert_array_equal(r1[1], r2[1])
            np.testing.assert_array_equal(r1[2], r2[2])
            r1 = oc.pam_update(P, r1[0], r1[2], r1[1], random_state=rs)
            r2 = oc.pam_update_numpy(P, r2[0], r2[2], r2[1], random_state=rs2)
    props = [int(np.flatnonzero(a == c)[-1]) for c in range(K)]
    r1 = oc.pam_update(P, inds, a, d, proposals=props)
    r2 = oc.pam_update_numpy(P, inds, a, d, proposals=props)
    assert [int(i) for i in r1[0]] == [int(i) for i in r2[0]]
    np.testing.assert_array_equal(r1[1], r2[1])
    np.testing.assert_array_equal(r1[2], r2[2])
