"""Measurement only: how long does a distances-only pass over the frames take
with 16 candidate centers (a build with -DEK_MAX_CANDS=16) next to 8?
usage: ENSPARA_HIP_LIB=enspara_amd/libek_c16.so python tools/pass16_probe.py [n A]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from enspara_amd import synth  # noqa: E402
from enspara_amd.device import FrameStore  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
A = int(sys.argv[2]) if len(sys.argv) > 2 else 300
x = synth.synth(n, A, 5000, 1)
st = FrameStore.from_array(x)
st.reset_state()
idx, cd, mx = st.kcenters_run(0, 64, 0.0)
st.pam_begin([int(i) for i in idx])
rng = np.random.RandomState(0)
frames = rng.choice(n, size=16, replace=False)
for count in (8, 16, 8, 16):
    st.pam_prefetch(frames[:count]); st.sync()
    t = time.time()
    reps = 20
    for _ in range(reps):
        st.pam_prefetch(frames[:count])
    st.sync()
    dt = (time.time() - t) / reps
    print("count %2d: %.3f ms per pass, %.4f ms per candidate, %.2f TB/s of frame bytes"
          % (count, dt * 1e3, dt * 1e3 / count, n * (12 * A + 8) / dt / 1e12), flush=True)
