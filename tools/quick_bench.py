"""Scratch benchmark: k-centers iterations/s per frames-per-lane variant."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from enspara_amd import synth
from enspara_amd.device import FrameStore

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
A = int(sys.argv[2]) if len(sys.argv) > 2 else 300
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
t = time.time()
tm = synth.templates(100, A, 1)
x = np.concatenate([synth.synth_chunk(c, min(synth.CHUNK, n - c * synth.CHUNK), tm, 1)
                    for c in range((n + synth.CHUNK - 1) // synth.CHUNK)])
print("synth %.1fs" % (time.time() - t), x.shape, flush=True)
t = time.time()
st = FrameStore.from_array(x)
st.sync()
print("upload+prepare %.2fs" % (time.time() - t), flush=True)
bpp = 12 * A + 20
import itertools
for nt, fpl in [(1, 2), (4, 0), (8, 0)]:
    st.set_frames_per_lane(fpl)
    st.set_option(4, nt)
    for rep in range(3):
        st.reset_state()
        t = time.time()
        idx, cd, mx = st.kcenters_run(0, K, 0.0)
        wall = time.time() - t
        ms, k = st.last_run_timing()
        k = len(idx)
    pairs = n * k / (ms * 1e-3)
    print("cands=%d passes=%d " % (nt, k if nt == 1 else st.last_run_timing()[1]), end="")
    print("fpl=%d  k=%d  dev %.2f ms (%.3f ms/iter)  wall %.2f ms  %.3e pairs/s  %.0f GB/s (%.1f%% of 8TB/s)"
          % (fpl, k, ms, ms / k, wall * 1e3, pairs, pairs * bpp / 1e9, pairs * bpp / 8e12 * 100), flush=True)
print("centers", idx[:8], "max", mx)
