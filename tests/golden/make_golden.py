#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (bowman-lab/enspara
at /root/reference) in this container.

Run here only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

How the reference is made importable without touching it: its package is
copied to a temporary directory, its three Cython extensions are built there
(setup.py build_ext --inplace), and import-only stub modules stand in for
`mdtraj` and `tables` (imported at module top by the reference but absent
here; nothing on the paths exercised below calls into them).  The reference's
control flow -- kcenters(), assign_to_nearest_center(), find_cluster_centers(),
_kmedoids_pam_update(), hybrid(), ClusterResult.partition(), RaggedArray,
assigns_to_counts(), builders.normalize(), eigenspectrum() -- then runs
unmodified.  The metric plugged into it is the oracle's QCP RMSD
(oracle/qcp.py), which is itself pinned against the reference's
mdtraj-produced known answers (tests/test_oracle.py).  Only inputs and
outputs are stored; no reference source is copied into the repository.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def import_reference():
    tmp = os.path.join(tempfile.gettempdir(), "enspara_ref_build")
    if not os.path.exists(os.path.join(tmp, "enspara", "geometry")) or not any(
            f.startswith("libdist") and f.endswith(".so")
            for f in os.listdir(os.path.join(tmp, "enspara", "geometry"))):
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        shutil.copytree(os.path.join(REF, "enspara"),
                        os.path.join(tmp, "enspara"))
        for f in ("setup.py", "pyproject.toml"):
            shutil.copy(os.path.join(REF, f), tmp)
        subprocess.check_call([sys.executable, "setup.py", "build_ext",
                               "--inplace"], cwd=tmp,
                              stdout=subprocess.DEVNULL)
    md = types.ModuleType("mdtraj")
    md.io = types.ModuleType("mdtraj.io")
    md.Trajectory = type("Trajectory", (), {})
    sys.modules["mdtraj"] = md
    sys.modules["mdtraj.io"] = md.io
    sys.modules["tables"] = types.ModuleType("tables")
    sys.path.insert(0, tmp)
    import enspara  # noqa: F401
    return tmp


class FakeTraj:
    """Just enough of md.Trajectory for util.assign_to_nearest_center's
    per-frame branch (util.py:193-197): .xyz, len(), indexing, iteration."""

    def __init__(self, xyz):
        self.xyz = np.asarray(xyz, dtype=np.float32)
        if self.xyz.ndim == 2:
            self.xyz = self.xyz[None]

    def __len__(self):
        return len(self.xyz)

    def __getitem__(self, i):
        return FakeTraj(self.xyz[i])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


def make_trim_golden(rtm, rbuilders):
    """Ergodic trimming (transition_matrices.py:236-301) and MSM(trim=True):
    a count matrix with a large component, a closed small one, a source-only
    and a sink-only state."""
    import scipy.sparse
    from enspara.msm.msm import MSM as RMSM
    out = {}
    rng = np.random.RandomState(21)
    K = 40
    big, small = list(range(0, 30)), list(range(30, 36))
    rows = []
    for _ in range(30):
        comp = big if rng.rand() < 0.8 else small
        s = comp[rng.randint(len(comp))]
        t = []
        for _ in range(200):
            t.append(s)
            s = comp[(comp.index(s) + rng.choice([-1, 0, 1, 2])) % len(comp)]
        rows.append(t)
    rows[0] = [36] + rows[0][1:]             # 36: left once, never entered
    rows[1] = rows[1][:-1] + [37]            # 37: entered once, never left
    rows[2] = rows[2][:100] + [38, 38, 39, 39, 38] + rows[2][105:]
    assigns = np.array(rows, dtype=np.int64)
    out["assigns"] = assigns
    C = rtm.assigns_to_counts(assigns, lag_time=1, max_n_states=K)
    out["counts"] = np.asarray(C.todense())
    for thr in (1, 2, 5):
        for ren in (True, False):
            m, tc = rtm.trim_disconnected(C, threshold=thr,
                                          renumber_states=ren)
            key = "thr%d_ren%d" % (thr, int(ren))
            out[key + "_counts"] = np.asarray(tc.todense())
            out[key + "_map"] = np.array(sorted(m.to_original.items()))
            assert type(tc) is type(C)
    m = RMSM(lag_time=1, method=rbuilders.normalize, trim=True,
             max_n_states=K)
    m.fit(assigns)
    out["msm_tcounts"] = np.asarray(m.tcounts_.todense())
    out["msm_tprobs"] = np.asarray(m.tprobs_.todense())
    out["msm_eq"] = np.asarray(m.eq_probs_)
    out["msm_map"] = np.array(sorted(m.mapping_.to_original.items()))
    from enspara.msm import timescales as rts
    out["implied_lags"] = np.array([1, 2, 4])
    out["implied_times_trim"] = rts.implied_timescales(
        assigns, [1, 2, 4], rbuilders.normalize, n_times=3, trim=True)
    np.savez_compressed(os.path.join(HERE, "trim_golden.npz"), **out)
    print("trim_golden.npz", os.path.getsize(
        os.path.join(HERE, "trim_golden.npz")) // 1024, "KiB")


def main():
    if "--only-trim" in sys.argv:
        import_reference()
        from enspara.msm import builders as rbuilders
        from enspara.msm import transition_matrices as rtm
        make_trim_golden(rtm, rbuilders)
        return
    import_reference()
    import logging
    logging.disable(logging.CRITICAL)
    from enspara.cluster import kcenters as rkc, util as rutil, hybrid as rhy
    from enspara.cluster import kmedoids as rkm
    from enspara import ra as rra
    from enspara.msm import builders as rbuilders
    from enspara.msm import transition_matrices as rtm
    from oracle import qcp, xtc
    from enspara_amd import synth

    metric = qcp.rmsd
    out = {}

    # ---- 1. k-centers on synthetic frames ---------------------------------
    X = synth.synth(2000, 50, 10, seed=0)
    r = rkc.kcenters(X, metric, n_clusters=20)
    out["kc_X_args"] = np.array([2000, 50, 10, 0])
    out["kc_n20_idx"] = np.array(r.center_indices)
    out["kc_n20_assign"] = r.assignments
    out["kc_n20_dist"] = r.distances
    r = rkc.kcenters(X, metric, dist_cutoff=0.35)
    out["kc_cut_cutoff"] = np.array(0.35)
    out["kc_cut_idx"] = np.array(r.center_indices)
    out["kc_cut_assign"] = r.assignments
    out["kc_cut_dist"] = r.distances
    r = rkc.kcenters(X, metric, n_clusters=6, dist_cutoff=0.05)
    out["kc_both_idx"] = np.array(r.center_indices)
    out["kc_both_assign"] = r.assignments
    # warm start from three existing frames
    init = [X[5], X[100], X[700]]
    r = rkc.kcenters(X, metric, n_clusters=9, init_centers=init)
    out["kc_init_frames"] = np.array([5, 100, 700])
    out["kc_init_idx"] = np.array(r.center_indices)
    out["kc_init_assign"] = r.assignments
    out["kc_init_dist"] = r.distances

    # ---- 2. BASELINE.json configs[0]: 1k x 100, 10 clusters ----------------
    X1 = synth.synth(1000, 100, 10, seed=0)
    r = rkc.kcenters(X1, metric, n_clusters=10)
    out["c1_args"] = np.array([1000, 100, 10, 0])
    out["c1_idx"] = np.array(r.center_indices)
    out["c1_assign"] = r.assignments
    out["c1_dist"] = r.distances

    # ---- 3. assign_to_nearest_center, both branches; find_cluster_centers --
    Y = synth.synth(300, 50, 300, seed=5)          # "centers" from other data
    a, d = rutil.assign_to_nearest_center(X, [c for c in Y[:17]], metric)
    out["asg_centers_args"] = np.array([300, 50, 300, 5, 17])
    out["asg_assign"] = a
    out["asg_dist"] = d
    out["asg_fcc"] = rutil.find_cluster_centers(a, d)
    # more centers than frames + Trajectory-like centers -> per-frame branch
    few = FakeTraj(X[:40])
    many = FakeTraj(Y[:120])
    a2, d2 = rutil.assign_to_nearest_center(few, many, metric)
    out["asg2_assign"] = a2
    out["asg2_dist"] = d2
    # the same through the center-major branch must agree
    a3, d3 = rutil.assign_to_nearest_center(X[:40], [c for c in Y[:120]],
                                            metric)
    assert np.array_equal(a2, a3) and np.array_equal(d2, d3)

    # ---- 4. k-hybrid / PAM ---------------------------------------------------
    Xh = synth.synth(1200, 40, 8, seed=3)
    r = rhy.hybrid(Xh, metric, n_iters=2, n_clusters=12,
                   random_state=np.random.RandomState(0))
    out["hy_X_args"] = np.array([1200, 40, 8, 3])
    out["hy_rs_idx"] = np.array(r.center_indices)
    out["hy_rs_assign"] = r.assignments
    out["hy_rs_dist"] = r.distances
    r = rhy.hybrid(Xh, metric, n_iters=2, n_clusters=12, random_state=0)
    out["hy_int_idx"] = np.array(r.center_indices)
    out["hy_int_assign"] = r.assignments
    out["hy_int_dist"] = r.distances
    r = rhy.hybrid(Xh, metric, n_iters=0, n_clusters=12)
    out["hy_0_idx"] = np.array(r.center_indices)
    out["hy_0_assign"] = r.assignments
    # one PAM sweep with fixed proposals
    r = rkc.kcenters(Xh, metric, n_clusters=8)
    props = np.array([17, 400, 33, 801, 1100, 250, 999, 64])
    # proposals must belong to... any frame is legal (kmedoids.py:622-628)
    mi, dd, aa, cc = rkm._kmedoids_pam_update(
        Xh, metric, list(r.center_indices), r.assignments.copy(),
        r.distances.copy(), proposals=props, random_state=0)
    out["pam_props"] = props
    out["pam_idx"] = np.array(mi)
    out["pam_assign"] = aa
    out["pam_dist"] = dd

    # ---- 5. estimator surface: fit / predict / partition ----------------------
    est = rkc.KCenters(metric, n_clusters=7).fit(Xh[:900])
    p = est.predict(Xh[900:])
    out["pred_idx"] = np.array(est.center_indices_)
    out["pred_assign"] = p.assignments
    out["pred_dist"] = p.distances
    out["pred_center_indices"] = np.array(p.center_indices)
    res = est.result_
    part = res.partition([300, 300, 300])
    out["part_eq_assign"] = part.assignments
    out["part_eq_ci"] = np.array(part.center_indices)
    part = res.partition([100, 500, 300])
    out["part_rag_lengths"] = np.array([100, 500, 300])
    out["part_rag_assign_data"] = part.assignments._data
    out["part_rag_assign_lengths"] = part.assignments.lengths
    out["part_rag_ci"] = np.array(part.center_indices)

    # ---- 6. the reference's own fixture: frame0.xtc ---------------------------
    fx = xtc.read_xtc(os.path.join(REF, "enspara/test/data/frame0.xtc"))["xyz"]
    out["frame0_xyz"] = fx
    r = rkc.kcenters(fx, metric, n_clusters=3)
    out["frame0_k3_idx"] = np.array(r.center_indices)
    out["frame0_k3_assign"] = r.assignments
    out["frame0_k3_dist"] = r.distances
    r = rkc.kcenters(fx, metric, dist_cutoff=0.1)
    out["frame0_cut_idx"] = np.array(r.center_indices)
    out["frame0_cut_assign"] = r.assignments
    out["frame0_cut_dist"] = r.distances
    # mdtraj-produced constants asserted by the reference
    # (enspara/test/test_cluster.py:209-218 and :231-238)
    out["frame0_ref_k3_mean_std"] = np.array([0.10387578309920734,
                                              0.018355072790569946])
    out["frame0_ref_cut_n_mean_std"] = np.array([17, 0.074690734158752686,
                                                 0.018754008455304401])

    np.savez_compressed(os.path.join(HERE, "cluster_golden.npz"), **out)

    # ---- 7. RaggedArray behaviours (reference ra.py; test_ra.py) ---------------
    rag = {}
    rows = [np.arange(5), np.arange(3) + 10, np.arange(7) + 20, np.arange(2)]
    a = rra.RaggedArray(rows)
    rag["rows_lengths"] = a.lengths
    rag["data"] = a._data
    rag["starts"] = a.starts
    rag["slice_1_3_lengths"] = a[1:3].lengths
    rag["slice_1_3_data"] = a[1:3]._data
    rag["cols_0_2_data"] = a[:, 0:2]._data
    rag["cols_0_2_lengths"] = a[:, 0:2].lengths
    rag["cols_neg_data"] = a[:, :-1]._data
    rag["cols_neg_lengths"] = a[:, :-1].lengths
    rag["cols_step_data"] = a[:, 1:6:2]._data
    rag["cols_step_lengths"] = a[:, 1:6:2].lengths
    rag["fancy"] = a[([0, 2, 2], [1, 0, 6])]
    rag["gt_data"] = (a > 11)._data
    w = rra.where(a > 11)
    rag["where_rows"] = w[0]
    rag["where_cols"] = w[1]
    rag["mask_get"] = a[a > 11]
    rag["add_data"] = (a + 1)._data
    rag["partition_indices"] = np.array(
        rra.partition_indices([0, 4, 5, 7, 8, 15, 16], [5, 3, 7, 2]))
    np.savez_compressed(os.path.join(HERE, "ra_golden.npz"), **rag)

    # ---- 8. MSM: counts / normalise / eigenspectrum ----------------------------
    msm = {}
    rng = np.random.RandomState(7)
    K, n_trj, L = 60, 12, 4000
    assigns = np.empty((n_trj, L), dtype=np.int64)
    for t in range(n_trj):
        s = rng.randint(K)
        for i in range(L):
            assigns[t, i] = s
            s = (s + rng.choice([-2, -1, 0, 0, 1, 2, 3])) % K
    assigns[rng.rand(n_trj, L) < 0.001] = -1
    msm["assigns"] = assigns
    for lag in (1, 5):
        for sw in (True, False):
            C = rtm.assigns_to_counts(assigns, lag_time=lag,
                                      max_n_states=K, sliding_window=sw)
            msm["counts_lag%d_sw%d" % (lag, int(sw))] = np.asarray(
                C.todense())
    C = rtm.assigns_to_counts(assigns, lag_time=1, max_n_states=K)
    _, T, eq = rbuilders.normalize(C, calculate_eq_probs=True)
    msm["norm_T"] = np.asarray(T.todense())
    msm["norm_eq"] = eq
    _, Tt, eqt = rbuilders.transpose(C, calculate_eq_probs=True)
    msm["transpose_T"] = np.asarray(Tt.todense())
    msm["transpose_eq"] = eqt
    vals, vecs = rtm.eigenspectrum(T, n_eigs=5)
    msm["eig_vals"] = vals
    msm["eig_vecs"] = vecs
    # ragged input
    lens = [1000, 2500, 400, 3100]
    flat = np.concatenate([assigns[i, :n] for i, n in enumerate(lens)])
    ragA = rra.RaggedArray(flat, lengths=lens)
    C = rtm.assigns_to_counts(ragA, lag_time=3, max_n_states=K)
    msm["rag_lengths"] = np.array(lens)
    msm["rag_counts_lag3"] = np.asarray(C.todense())
    from enspara.msm import timescales as rts
    msm["implied_lags"] = np.array([1, 2, 5, 10])
    msm["implied_times"] = rts.implied_timescales(
        assigns, [1, 2, 5, 10], rbuilders.normalize, n_times=4)
    np.savez_compressed(os.path.join(HERE, "msm_golden.npz"), **msm)
    make_trim_golden(rtm, rbuilders)
    # ---- 9. feature-space metrics: the reference's own native kernels ----------
    from enspara.geometry import libdist as rlib
    feat = {}
    rng = np.random.RandomState(11)
    Xd = rng.normal(scale=3.0, size=(700, 37))
    yd = rng.normal(scale=3.0, size=37)
    for name, X_, y_ in (("f32", Xd.astype(np.float32), yd.astype(np.float32)),
                         ("f64", Xd, yd),
                         ("i64", (Xd * 10).astype(np.int64),
                          (yd * 10).astype(np.int64))):
        feat["X_" + name] = X_
        feat["y_" + name] = y_
        feat["euclidean_" + name] = rlib.euclidean(X_, y_)
        feat["manhattan_" + name] = rlib.manhattan(X_, y_)
    Xi = rng.randint(0, 4, size=(300, 21)).astype(np.int32)
    yi = rng.randint(0, 4, size=21).astype(np.int32)
    feat["X_ham"] = Xi
    feat["y_ham"] = yi
    feat["hamming"] = rlib.hamming(Xi, yi)
    # clustering in feature space through the string metric
    Xc = np.concatenate([c + rng.normal(size=(80, 6))
                         for c in rng.uniform(-15, 15, size=(6, 6))]
                        ).astype(np.float32)
    r = rkc.kcenters(Xc, "euclidean", n_clusters=6)
    feat["kc_X"] = Xc
    feat["kc_idx"] = np.array(r.center_indices)
    feat["kc_assign"] = r.assignments
    feat["kc_dist"] = r.distances
    r = rhy.hybrid(Xc, "euclidean", n_clusters=6, n_iters=2,
                   random_state=np.random.RandomState(3))
    feat["hy_idx"] = np.array(r.center_indices)
    feat["hy_assign"] = r.assignments
    feat["hy_dist"] = r.distances
    np.savez_compressed(os.path.join(HERE, "features_golden.npz"), **feat)

    for f in ("cluster_golden.npz", "ra_golden.npz", "msm_golden.npz",
              "features_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
