"""Scratch: time the distances-only multi-candidate pass (ek_pass_kernel<8,false>)
with experimental builds.  usage: ablate_pass.py <lib.so> [count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ENSPARA_HIP_LIB"] = os.path.abspath(sys.argv[1])
import numpy as np
from enspara_amd import synth
from enspara_amd.device import FrameStore
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, A = 1000000, 300
x = synth.synth(n, A, 5000, 1)
st = FrameStore.from_array(x)
st.reset_state()
st.kcenters_run(0, 16, 0.0)
st.pam_begin(list(range(0, 16 * 1000, 1000)))
fr = list(range(5, 5 + 977 * cnt, 977))
for rep in range(3):
    st.pam_prefetch(fr); st.sync()
    t = time.time()
    for i in range(100):
        st.pam_prefetch(fr)
    st.sync()
    dt = (time.time() - t) / 100
    print("%s count=%d  %.4f ms/pass  %.0f GB/s" % (os.path.basename(sys.argv[1]), cnt, dt * 1e3, n * (12 * A + 4 + 4 * cnt) / dt / 1e9), flush=True)
