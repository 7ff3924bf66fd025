"""Which path do the windows of a PAM sweep take with the tables as bounds
(option key 16) and exact?  (debug, round 5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from enspara_amd import synth
from enspara_amd.cluster import kmedoids as km
from enspara_amd.device import FrameStore
for n, A, K, nt, seed in ((20000, 10, 200, 200, 3), (100000, 30, 1000, 1000, 4)):
    X = synth.synth(n, A, nt, seed=seed)
    for opt in (1, 0):
        with FrameStore.from_array(X) as st:
            st.set_option(16, opt)
            st.reset_state()
            idx, _, _ = st.kcenters_run(0, K, 0.0)
            r = km._kmedoids_iterations_device(X, st, 2, [int(i) for i in idx], None,
                                               np.random.RandomState(7))
            print(n, A, K, "bounds" if opt else "exact", "restricted/full",
                  st.pam_prefetch_passes(), "sparse windows/ended early",
                  st.pam_sparse_stats(), "prefetch hits/misses", st.pam_prefetch_stats(),
                  flush=True)
