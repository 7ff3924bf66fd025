"""Measurement only: feature-space k-centers, whole loop resident on the device
(kcenters(X, 'euclidean')) vs the reference-shaped host loop around the device
metric (a wrapped callable of the same metric).  feat_probe.py [n] [features] [centers]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from enspara_amd.cluster.kcenters import kcenters  # noqa: E402
from enspara_amd.geometry import libdist  # noqa: E402

n, F, K = [int(v) for v in (sys.argv[1:] + ["1000000", "64", "200"][len(sys.argv) - 1:])]
X = np.random.RandomState(0).normal(size=(n, F)).astype(np.float32)
for name, metric in (("resident loop", "euclidean"),
                     ("host loop + device metric", libdist.euclidean.bind(X))):
    kcenters(X[:4096], "euclidean", n_clusters=3)           # warm the library
    t0 = time.perf_counter()
    r = kcenters(X, metric if isinstance(metric, str) else
                 (lambda A, y, f=metric: f(A, y)), n_clusters=K)
    dt = time.perf_counter() - t0
    print("%-28s %d x %d, %d centers: %.3f s  (%.3f ms per center)  last center %d"
          % (name, n, F, K, dt, dt / K * 1e3, r.center_indices[-1]), flush=True)
