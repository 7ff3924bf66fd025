"""Transition counts at scale (measurement only): 10^7 frames in 100
trajectories over 5000 and 20 000 states, lag 10, against a scipy
construction: labels from host arrays, and labels resident in a FrameStore
(the state a fit leaves in HBM)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse

from enspara_amd.device import FrameStore
from enspara_amd.msm import assigns_to_counts

for K in (5000, 20000):
    rng = np.random.RandomState(3)
    n_trj, L, lag = 100, 100000, 10
    A = (rng.randint(K, size=(n_trj, 1)) +
         np.cumsum(rng.randint(-20, 21, size=(n_trj, L)), axis=1)) % K
    A[rng.rand(n_trj, L) < 0.001] = -1
    A = A.astype(np.int32)
    best = None
    for rep in range(3):
        t = time.perf_counter()
        C = assigns_to_counts(A, lag_time=lag, max_n_states=K)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    rows, cols = [], []
    for a in A:
        a = a[a != -1]
        rows.append(a[:-lag])
        cols.append(a[lag:])
    t = time.perf_counter()
    ref = scipy.sparse.coo_matrix(
        (np.ones(sum(len(r) for r in rows), dtype=np.int64),
         (np.concatenate(rows), np.concatenate(cols))), shape=(K, K)).tocsr()
    t_ref = time.perf_counter() - t
    print("%d states, %d frames: %.2f ms (host arrays in, COO out; scipy coo->csr "
          "%.0f ms), %d entries, equal: %s"
          % (K, A.size, best * 1e3, t_ref * 1e3, C.nnz,
             (C.tocsr() != ref).nnz == 0), flush=True)
    # the same labels resident in a store (one-atom frames: only the state matters)
    st = FrameStore(A.size, 1, device=0)
    st.load(np.zeros((A.size, 1, 3), dtype=np.float32))
    st.upload_state(np.zeros(A.size, dtype=np.float32), A.reshape(-1))
    lengths = [L] * n_trj
    best = None
    for rep in range(4):
        st.sync()
        t = time.perf_counter()
        r, c, v = st.msm_counts(lengths, lag, K)
        dt = time.perf_counter() - t
        if rep == 0:
            first = dt
        else:
            best = dt if best is None else min(best, dt)
    Cr = scipy.sparse.coo_matrix((v, (r, c)), shape=(K, K)).tocsr()
    print("%d states: resident labels -> COO on the host %.2f ms (first call, "
          "buffers allocated: %.2f ms), equal: %s; %.2e transitions/s, %.1f GB/s "
          "against 8 B per transition"
          % (K, best * 1e3, first * 1e3, (Cr != ref).nnz == 0,
             A.size / best, 8 * A.size / best / 1e9), flush=True)
    st.close()
