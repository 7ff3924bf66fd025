"""One-workgroup PAM windows against the three-launch form (measurement and a
quick equality check, not a test): the same sweeps with ek_set_option key 12 =
1 and 0, medoids + final state compared, time per sweep.

  sp_check.py [--big | --only-big]   (--big: 10^6 x 300, 5000 medoids as well;
                                      --only-big: that case alone, one-workgroup form only)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from enspara_amd import synth
from enspara_amd.cluster import kmedoids as km
from enspara_amd.device import FrameStore

CASES = [  # n, atoms, K, templates (0: a time-ordered walk), sweeps
    (20000, 10, 200, 200, 2),
    (24000, 8, 90, 6, 2),
    (50000, 30, 700, 0, 2),
    (100003, 20, 1500, 1500, 2),
    (70001, 12, 300, 0, 2),
    (16500, 5, 64, 64, 3),
    (30000, 8, 600, 2, 2),      # two templates, 600 medoids: everything within reach
    (41000, 20, 900, 10, 2),
]
if "--big" in sys.argv:
    CASES.append((1_000_000, 300, 5000, 5000, 2))
if "--only-big" in sys.argv:        # (for a kernel trace: one process, one case)
    CASES = [(1_000_000, 300, 5000, 5000, 1)]
MODES = (1,) if "--only-big" in sys.argv else (1, 0)

bad = 0
for n, A, K, nt, sweeps in CASES:
    x = synth.synth(n, A, nt, seed=n % 97) if nt else synth.walk(n, A, seed=n % 97)
    out = {}
    for mode in MODES:
        with FrameStore.from_array(x) as st:
            st.set_option(12, mode)
            st.reset_state()
            idx, _, _ = st.kcenters_run(0, K, 0.0)
            med = [int(i) for i in idx]
            rs = np.random.RandomState(5)
            st.sync()
            t = time.perf_counter()
            for _ in range(sweeps):
                med = km._pam_sweep_device(st, med, None, rs)
            st.sync()
            dt = (time.perf_counter() - t) / sweeps
            d, a = st.download_state()
            out[mode] = (list(med), d.copy(), a.copy(), dt, st.pam_sparse_stats(),
                         st.pam_prefetch_passes())
    if 0 not in out:
        out[0] = out[1]
    same = (out[1][0] == out[0][0] and np.array_equal(out[1][1], out[0][1]) and
            np.array_equal(out[1][2], out[0][2]))
    bad += not same
    print("%8d x %3d, %5d medoids, %s: %s; one workgroup %.2f ms per sweep "
          "(%.1f us per proposal; %d windows, %d ended early; prefetches "
          "restricted %d full %d), three launches %.2f ms (%.1f us)"
          % (n, A, K, "templates" if nt else "walk", "same" if same else "DIFFERENT",
             out[1][3] * 1e3, out[1][3] / K * 1e6, out[1][4][0], out[1][4][1],
             out[1][5][0], out[1][5][1], out[0][3] * 1e3, out[0][3] / K * 1e6),
          flush=True)
print("sp_check: %d cases differ" % bad)
sys.exit(1 if bad else 0)
