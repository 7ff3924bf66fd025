"""Leading eigenpairs of the bench's MSM matrix on the device (measurement only):
the restarted iteration on the matrix itself (FILTER = 0, round 4's) and on a Chebyshev
polynomial of it (FILTER = 1), against ARPACK.

  eig_probe.py [trajectories of 10 000 frames, default 1000]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse
import scipy.sparse.linalg

from enspara_amd.msm import transition_matrices as tm

K, L, lag = 5000, 10000, 1
n_trj = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.RandomState(11)
steps = rng.choice(np.array([-3, -2, -1, 0, 0, 1, 2, 3], dtype=np.int8), size=(n_trj, L))
inblock = (rng.randint(100, size=(n_trj, 1)) + np.cumsum(steps, axis=1, dtype=np.int32)) % 100
hops = np.cumsum(rng.rand(n_trj, L) < 0.002, axis=1, dtype=np.int32)
block = (rng.randint(K // 100, size=(n_trj, 1)) + hops * 7) % (K // 100)
A = (block * 100 + inblock).astype(np.int32)
A[rng.rand(n_trj, L) < 0.001] = -1
rows, cols = [], []
for a in A:
    a = a[a != -1]
    rows.append(a[:-lag])
    cols.append(a[lag:])
C = scipy.sparse.coo_matrix((np.ones(sum(len(r) for r in rows)),
                             (np.concatenate(rows), np.concatenate(cols))), shape=(K, K)).tocsr()
T = scipy.sparse.diags(1.0 / np.asarray(C.sum(axis=1)).ravel()) @ C
t = time.perf_counter()
want = np.sort(scipy.sparse.linalg.eigs(T.T.tocsr(), k=20, which="LR", tol=1e-12,
                                        return_eigenvectors=False).real)[::-1]
t_arpack = time.perf_counter() - t
print("%d states, %d non-zeros; ARPACK (host) %.3f s" % (K, T.nnz, t_arpack))
tm.eigenspectrum(T, n_eigs=20)          # (the library and the device awake)
for f in (0, 1, 0, 1):
    tm.FILTER = f
    best = None
    for _ in range(3):
        t = time.perf_counter()
        vals, vecs = tm.eigenspectrum(T, n_eigs=20)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    print("FILTER %d: %.4f s  max |eigenvalue - ARPACK's| %.1e  %s"
          % (f, best, np.abs(vals - want).max(), tm.LAST_RUN), flush=True)
