import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from enspara_amd import synth
from enspara_amd.device import FrameStore
n, A, K, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tm = synth.templates(5000, A, 1)
x = np.concatenate([synth.synth_chunk(c, min(synth.CHUNK, n - c * synth.CHUNK), tm, 1)
                    for c in range((n + synth.CHUNK - 1) // synth.CHUNK)])
st = FrameStore.from_array(x)
st.set_option(4, T)
st.reset_state()
idx, cd, mx = st.kcenters_run(0, K, 0.0)
ms, k = st.last_run_timing()
print("T=%d centers=%d rounds=%d %.2f ms  %.3f ms/center" % (T, len(idx), k, ms, ms / len(idx)))
