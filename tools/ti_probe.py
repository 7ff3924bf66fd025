"""Measurement only: k-centers with the triangle-inequality tile skip
(kcenters(..., use_triangle_inequality=True) = FrameStore option 11) on frames
stored in blocks of one template each (what a set of trajectories in time
order looks like), next to the default run on the same frames.
  ti_probe.py [n_templates] [frames_per_template] [atoms] [centers]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from enspara_amd import synth  # noqa: E402
from enspara_amd.device import FrameStore  # noqa: E402

T, per, A, K = [int(v) for v in (sys.argv[1:] + ["2000", "500", "300", "3000"][len(sys.argv) - 1:])]
rng = np.random.default_rng(5)
tmpl = synth.templates(T, A, 4)
x = np.empty((T * per, A, 3), dtype=np.float32)
for t in range(T):
    x[t * per:(t + 1) * per] = tmpl[t] + rng.normal(scale=0.05, size=(per, A, 3))
with FrameStore.from_array(x) as st:
    res = {}
    for name, tri in (("default", 0), ("triangle", 1), ("one center per pass", 2),
                      ("triangle, one center per pass (round 4)", 3)):
        st.set_option(11, 1 if tri in (1, 3) else 0)
        st.set_option(4, 1 if tri >= 2 else -1)
        st.reset_state()
        st.sync()
        t0 = time.perf_counter()
        idx, cd, mx = st.kcenters_run(0, K, 0.0)
        dt = time.perf_counter() - t0
        d, a = st.download_state()
        res[name] = (idx, d, a)
        extra = ""
        if tri in (1, 3):
            tiles, skipped = st.ti_stats()
            extra = "  tiles %d, skipped %d (%.1f %%)" % (tiles, skipped,
                                                          100.0 * skipped / max(tiles, 1))
        print("%-40s %d frames x %d atoms, %d centers: %.3f s  %.3e pairs/s%s"
              % (name, len(x), A, K, dt, len(x) * K / dt, extra), flush=True)
    a0 = res["default"]
    for k, v in res.items():
        assert np.array_equal(v[0], a0[0]) and np.array_equal(v[1], a0[1]) \
            and np.array_equal(v[2], a0[2]), k
    print("identical centers, labels, distances")
