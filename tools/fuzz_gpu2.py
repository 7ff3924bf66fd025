"""Scratch: randomized comparison with the oracle, part 2: nearest-center
assignment (both kernels), the multi-rank drivers on a single DeviceShard, and
warm-started k-centers.  usage: fuzz_gpu2.py [n_cases] [seed]"""
import os, sys, time
# (the oracle's OpenMP team: a box shows 256 CPUs and grants 16)
os.environ.setdefault("OMP_NUM_THREADS", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from enspara_amd import synth, sharded
from enspara_amd.cluster import kcenters as kc
from enspara_amd.device import FrameStore
from oracle import cluster as oc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
ts = torch.cuda.Stream(device=0)
for case in range(cases):
    if case and case % 200 == 0:
        print("... %d cases, %d mismatches, %.0f s" % (case, bad, time.time() - t0), flush=True)
    rng = np.random.RandomState(base * 100003 + case)
    n = int(rng.choice([1, 3, 63, 64, 65, 255, 257, 1000, 1025, 4099]))
    A = int(rng.choice([1, 2, 3, 5, 16, 33, 100]))
    nt = int(rng.choice([1, 3, 40]))
    x = synth.synth(n, A, nt, seed=int(rng.randint(1 << 30)))
    kind = rng.choice(["assign", "sharded", "warm"])
    tag = "case %d %s n=%d A=%d nt=%d" % (case, kind, n, A, nt)
    ok = True
    try:
        if kind == "assign":
            K = int(rng.choice([1, 2, 23, 24, 25, 64, 70, 200]))
            src = rng.randint(0, n, size=K)
            ctr = x[src] if rng.rand() < 0.5 else synth.synth(K, A, max(K, 1), seed=case)
            a, d = oc.assign_to_nearest_center(x, ctr)
            for variant in (1, 2, 3):
                with FrameStore.from_array(x) as st:
                    st.set_option(2, variant)
                    st.assign_nearest(ctr)
                    dd, aa = st.download_state()
                ok = ok and np.array_equal(aa, a) and np.array_equal(dd.astype(np.float64), d)
        elif kind == "sharded":
            K = int(rng.choice([1, 3, 9, 40]))
            sweeps = int(rng.choice([0, 1, 2]))
            inds, a, d = oc.kcenters(x, n_clusters=K)
            rs = np.random.RandomState(case)
            wi, wd, wa = list(inds), d.copy(), a.copy()
            try:
                for _ in range(sweeps):
                    wi, wd, wa = oc.pam_update(x, wi, wa, wd, random_state=rs)
                want_err = None
            except ValueError as e:
                want_err = e
            with FrameStore(n, A, device=0, stream=ts.cuda_stream) as st:
                st.load(x); st.reset_state()
                sh = sharded.DeviceShard(st)
                try:
                    with torch.cuda.stream(ts):
                        med = sharded.khybrid_sharded(sh, K, 0.0, sweeps, random_state=case)
                    got_err = None
                except ValueError as e:
                    got_err = e
                dd, aa = st.download_state()
            if want_err is None and got_err is None:
                ok = (list(med) == [int(i) for i in wi] and np.array_equal(aa, wa)
                      and np.array_equal(dd.astype(np.float64), wd))
            else:
                ok = (want_err is None) == (got_err is None)
        else:
            K0 = int(rng.choice([1, 2, 5])); K = K0 + int(rng.choice([0, 1, 7, 30]))
            init = [x[int(i)] for i in rng.randint(0, n, size=K0)]
            inds, a, d = oc.kcenters(x, n_clusters=K, init_centers=init)
            r = kc.kcenters(x, "rmsd", n_clusters=K, init_centers=init)
            ok = (list(r.center_indices) == [int(i) for i in inds]
                  and np.array_equal(r.assignments, a) and np.array_equal(r.distances, d))
    except Exception as e:
        ok = False
        tag += " EXC %r" % (e,)
    if not ok:
        bad += 1
        print("MISMATCH:", tag, flush=True)
print("fuzz2: %d cases, %d mismatches, %.0f s" % (cases, bad, time.time() - t0))
