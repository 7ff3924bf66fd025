import os, sys, time
# (the oracle's OpenMP team: a box shows 256 CPUs and grants 16)
os.environ.setdefault("OMP_NUM_THREADS", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from enspara_amd import synth
from enspara_amd.cluster import hybrid as hy
from oracle import cluster as oc
n, A, nt, K, sweeps = [int(v) for v in sys.argv[1:6]]
cutoff = float(sys.argv[6]) if len(sys.argv) > 6 else None
x = synth.synth(n, A, nt, seed=5)
t = time.time()
try:
    inds, a, d = oc.kcenters(x, n_clusters=(K if K > 0 else None), dist_cutoff=cutoff)
    print("oracle kcenters %.2fs centers %d" % (time.time() - t, len(inds)), flush=True)
    t = time.time(); rs = np.random.RandomState(1)
    for _ in range(sweeps):
        inds, d, a = oc.pam_update(x, inds, a, d, random_state=rs)
    print("oracle pam %.2fs" % (time.time() - t), flush=True)
except Exception as e:
    print("oracle raised %r after %.2fs" % (e, time.time() - t), flush=True)
t = time.time()
try:
    r = hy.hybrid(x, "rmsd", n_iters=sweeps, n_clusters=(K if K > 0 else np.inf), dist_cutoff=(cutoff or 0), random_state=np.random.RandomState(1))
    print("device hybrid %.2fs centers %d" % (time.time() - t, len(r.center_indices)), flush=True)
except Exception as e:
    print("device raised %r after %.2fs" % (e, time.time() - t), flush=True)
