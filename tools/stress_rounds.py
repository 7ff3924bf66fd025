"""Measurement/parity stress: the three-launch round (last-workgroup tails,
coherent hand-over without fences) against the one-launch-per-step form, many
repetitions at several shard sizes -- a race would show as a checksum that
differs from run to run.  stress_rounds.py [reps]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from enspara_amd import synth  # noqa: E402
from enspara_amd.device import FrameStore  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for n, A, K, tmpl in ((700, 9, 300, 5), (5000, 20, 900, 40), (60000, 50, 1500, 300),
                      (300000, 100, 1200, 2000), (1000000, 300, 600, 5000)):
    x = synth.synth(n, A, tmpl, seed=n % 97)
    with FrameStore.from_array(x) as st:
        sums = {}
        for fused, count in ((0, 2), (1, reps)):
            st.set_option(10, fused)
            for r in range(count):
                st.reset_state()
                idx, cd, mx = st.kcenters_run(0, K, 0.0)
                d, a = st.download_state()
                h = hashlib.sha256(idx.tobytes() + cd.tobytes() + d.tobytes() +
                                   a.tobytes() + np.float32(mx).tobytes()).hexdigest()[:12]
                sums[h] = sums.get(h, 0) + 1
        ok = len(sums) == 1
        bad += 0 if ok else 1
        print("%8d x %3d, %4d centers: %s %s" % (n, A, K, "identical" if ok else "DIFFER",
                                                 sums), flush=True)
print("stress: %d sizes with differing runs" % bad)
