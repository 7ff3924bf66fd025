"""A/B two builds of the library on one box: alternate child processes, each
timing a k-centers run on the same frames (ENSPARA_HIP_LIB picks the build).
usage: ab_libs.py <libA.so> <libB.so> [more.so ...] [n A K reps]"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)

if sys.argv[1] == "--child":
    import numpy as np
    from enspara_amd.device import FrameStore
    x = np.load(sys.argv[2], mmap_mode="r")
    K = int(sys.argv[3])
    st = FrameStore.from_array(np.ascontiguousarray(x))
    out = []
    for rep in range(3):
        st.reset_state()
        t = time.time()
        idx, cd, mx = st.kcenters_run(0, K, 0.0)
        out.append(time.time() - t)
    print("%s  %s  last center %d" % (os.path.basename(os.environ["ENSPARA_HIP_LIB"]),
                                      " ".join("%.4f" % v for v in out), idx[-1]),
          flush=True)
    sys.exit(0)

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
nums = [a for a in sys.argv[1:] if not a.endswith(".so")]
n, A, K, reps = [int(v) for v in (nums + ["1000000", "300", "2000", "2"][len(nums):])]
from enspara_amd import synth  # noqa: E402
import numpy as np  # noqa: E402
path = "/tmp/ab_frames.npy"
np.save(path, synth.synth(n, A, 5000, 1))
for rep in range(reps):
    for lib in libs:
        env = dict(os.environ, ENSPARA_HIP_LIB=os.path.abspath(lib))
        subprocess.check_call([sys.executable, __file__, "--child", path, str(K)],
                              env=env)
