"""Scratch: randomized k-hybrid runs large enough for the restricted PAM prefetch
and the medoid pruning (n >= 16384), against the oracle."""
import os, sys, time
# (the oracle's OpenMP team: a box shows 256 CPUs and grants 16)
os.environ.setdefault("OMP_NUM_THREADS", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from enspara_amd import synth
from enspara_amd.cluster import kmedoids as km
from enspara_amd.device import FrameStore
from oracle import cluster as oc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0; t0 = time.time(); tot_r = tot_f = tot_sw = tot_se = 0
for case in range(cases):
    if case and case % 5 == 0:
        print("... %d cases, %d mismatches, %.0f s" % (case, bad, time.time() - t0), flush=True)
    rng = np.random.RandomState(base * 7919 + case)
    n = int(rng.randint(16384, 42000)); A = int(rng.choice([3, 4, 8, 20, 40]))
    nt = int(rng.choice([2, 10, 100, 400])); K = int(rng.choice([150, 300, 600, 900]))
    sweeps = int(rng.choice([1, 2]))
    x = synth.synth(n, A, nt, seed=int(rng.randint(1 << 30)))
    if rng.rand() < 0.25:       # a time-ordered walk: neighbouring medoids within reach
        x = synth.walk(n, A, seed=int(rng.randint(1 << 30)))
    if rng.rand() < 0.2:
        x = np.concatenate([x[: n // 2], x[: n - n // 2]])
    inds, a, d = oc.kcenters(x, n_clusters=K)
    props = None
    if rng.rand() < 0.3:
        props = [int(p) for p in rng.randint(0, n, size=len(inds))]
    rs = np.random.RandomState(case)
    wi, wd, wa = list(inds), d.copy(), a.copy()
    err = None
    try:
        for _ in range(sweeps):
            wi, wd, wa = oc.pam_update(x, wi, wa, wd, proposals=props, random_state=rs)
    except ValueError as e:
        err = e
    with FrameStore.from_array(x) as st:
        st.reset_state()
        idx, _, _ = st.kcenters_run(0, K, 0.0)
        try:
            r = km._kmedoids_iterations_device(x, st, sweeps, [int(i) for i in idx], props,
                                               np.random.RandomState(case))
            gerr = None
        except ValueError as e:
            gerr = e
        rr, ff = st.pam_prefetch_passes(); tot_r += rr; tot_f += ff
        sw, se = st.pam_sparse_stats(); tot_sw += sw; tot_se += se
    if err is None and gerr is None:
        ok = (list(r.center_indices) == [int(i) for i in wi] and np.array_equal(r.assignments, wa)
              and np.array_equal(r.distances, wd))
    else:
        ok = (err is None) == (gerr is None)
    if not ok:
        bad += 1
        print("MISMATCH: case %d n=%d A=%d nt=%d K=%d sweeps=%d props=%s" % (case, n, A, nt, K, sweeps, props is not None), flush=True)
print("fuzz3: %d cases, %d mismatches, %.0f s; prefetch passes restricted %d full %d; "
      "windows in one workgroup %d, of them ended early %d"
      % (cases, bad, time.time() - t0, tot_r, tot_f, tot_sw, tot_se))
