"""Host -> HBM rate of FrameStore.load (measurement only): context creation,
first load (allocates the pinned and the staging buffers), later loads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from enspara_amd.device import FrameStore
n, A = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 300
x = np.random.RandomState(0).standard_normal((n, A, 3)).astype(np.float32)
for thr, mb in (("8", "64"), ("8", "16"), ("8", "32"), ("8", "128"), ("8", "256"),
                ("4", "32"), ("1", "64")):
    os.environ["EK_UPLOAD_THREADS"] = thr
    os.environ["EK_UPLOAD_CHUNK_MB"] = mb
    t = time.perf_counter(); st = FrameStore(n, A, device=0); st.sync(); t_create = time.perf_counter() - t
    out = []
    for rep in range(3):
        t = time.perf_counter(); st.load(x); t_ret = time.perf_counter() - t; st.sync(); out.append((t_ret, time.perf_counter() - t))
    st.close()
    print("chunk %s MB, threads %s: create %.3f s; loads (returned / done) %s -> %.1f GB/s"
          % (mb, thr, t_create, " ".join("%.3f/%.3f" % o for o in out), x.nbytes / out[-1][1] / 1e9), flush=True)
