"""A PAM sweep in feature space: resident on the device (ek_feat_pam_sweep) against
the reference-shaped host loop around the device metric (measurement only).

  feat_pam_probe.py [n] [features] [medoids] [--no-host] [--clustered] [--explicit]
  (--no-host: the device sweep alone)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from enspara_amd.cluster import kmedoids as km
from enspara_amd.cluster.kcenters import kcenters

args = [v for v in sys.argv[1:] if not v.startswith('--')]
n = int(args[0]) if len(args) > 0 else 200000
F = int(args[1]) if len(args) > 1 else 16
K = int(args[2]) if len(args) > 2 else 400
rs0 = np.random.RandomState(0)
if '--clustered' in sys.argv:       # K tight clusters (what clustering is run on) instead of noise
    X = (rs0.normal(size=(K, F))[rs0.randint(0, K, size=n)] +
         0.05 * rs0.normal(size=(n, F))).astype(np.float32)
else:
    X = rs0.normal(size=(n, F)).astype(np.float32)
explicit = None
if '--explicit' in sys.argv:        # given proposals instead of drawn ones
    explicit = [int(v) for v in rs0.randint(0, n, size=K)]
r = kcenters(X, "euclidean", n_clusters=K)
out = {}
for dev in ((1,) if '--no-host' in sys.argv else (1, 0)):
    km.PAM_FEATURE_DEVICE = dev
    inds = [int(i) for i in r.center_indices]
    t = time.perf_counter()
    inds, d, a, _ = km._kmedoids_pam_update(X, "euclidean", inds, r.assignments.copy(),
                                            r.distances.copy(), proposals=explicit,
                                            random_state=np.random.RandomState(1))
    out[dev] = (time.perf_counter() - t, list(inds), d, a)
if 0 not in out:
    print("%d x %d, %d medoids, euclidean: device-resident sweep %.3f s (%.1f us per proposal)"
          % (n, F, K, out[1][0], out[1][0] / K * 1e6))
    sys.exit(0)
same = (out[1][1] == out[0][1] and np.array_equal(out[1][2], out[0][2]) and
        np.array_equal(out[1][3], out[0][3]))
print("%d x %d, %d medoids, euclidean: device-resident sweep %.3f s (%.1f us per proposal), "
      "host loop around the device metric %.3f s (%.1f us); same results: %s"
      % (n, F, K, out[1][0], out[1][0] / K * 1e6, out[0][0], out[0][0] / K * 1e6, same))
