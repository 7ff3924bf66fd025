"""Scratch: bisect the slow PAM sweep seen inside bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2: os.environ["ENSPARA_HIP_LIB"] = os.path.abspath(sys.argv[2])
import torch
import numpy as np
from enspara_amd import synth
from enspara_amd.device import FrameStore
from enspara_amd.cluster import kmedoids as km
mode = sys.argv[1]
n, A, K = 1000000, 300, 2000
x = synth.synth(n, A, 5000, 1)
torch.cuda.set_device(0)
st = FrameStore(n, A, device=0, global_offset=0, stream=None)
st.load(x); st.sync()
st.set_frames_per_lane(0); st.set_option(4, -1)
st.reset_state()
if mode == "two":
    w, _, _ = st.kcenters_run(0, 20, 0.0)
    idx, _, _ = st.kcenters_run(20, K - 20, 0.0)
    idx = np.concatenate([w, idx])
elif mode == "timing":
    w, _, _ = st.kcenters_run(0, 20, 0.0)
    st.timing_begin(sample_every=8, max_samples=512)
    torch.cuda.synchronize()
    idx, _, _ = st.kcenters_run(20, K - 20, 0.0)
    torch.cuda.synchronize()
    print(st.timing_end(), st.spec_rounds())
    idx = np.concatenate([w, idx])
else:
    idx, _, _ = st.kcenters_run(0, K, 0.0)
t = time.time()
med = km._pam_sweep_device(st, [int(i) for i in idx[:K]], None, np.random.RandomState(1))
torch.cuda.synchronize()
dt = time.time() - t
print("%s: sweep of %d: %.3fs  %.3f ms/proposal %s" % (mode, K, dt, dt / K * 1e3, st.pam_prefetch_stats()), flush=True)
