"""A/B of pass-kernel forms and builds on one box (measurement only).

  lab_pass.py [lib.so ...] [--n N] [--atoms A] [--centers K]

For every library given (default: the in-tree build) a child process loads the
same frames and times a k-centers run of K centers for each combination of
  form   (ignored: round 1's LDS form of the pass kernel is retired)
  adapt  0 = always the pinned / widest form, 1 = 1/8/16 by measured rate
(LAB_CONFIGS="form,adapt,cands[,fused[,fine[,sweep]]];..." picks the combinations) printing seconds per run, the mean pass-kernel time (HIP events), the passes
by candidates per pass, and a checksum of centers + final state (all
combinations must agree: the forms are bit-identical by construction).
Variants are built with enspara_amd.build.build(out=..., tag=..., extra_flags=[...]).
"""
import hashlib
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)


def child(path, K):
    import numpy as np
    from enspara_amd.device import FrameStore
    x = np.load(path, mmap_mode="r")
    st = FrameStore.from_array(np.ascontiguousarray(x))
    name = os.path.basename(os.environ.get("ENSPARA_HIP_LIB", "default"))
    sums = set()
    configs = os.environ.get("LAB_CONFIGS")
    configs = ([tuple(int(v) for v in c.split(",")) for c in configs.split(";")]
               if configs else [(1, 0, 8), (1, 0, 16), (1, 1, -1), (1, 0, 4),
                                (1, 0, 1)])
    for cfg in configs:
        form, adapt, cands = cfg[:3]
        fused = cfg[3] if len(cfg) > 3 else 1
        fine = cfg[4] if len(cfg) > 4 else 1
        sweep = cfg[5] if len(cfg) > 5 else 1   # 0 never, 1 by size, 2 always
        st.set_option("pass_sweep", sweep)  # per-prefix maxima taken by the pass (round 6)
        st.set_option(15, fine)         # maxima per 64 frames for the pick
        st.set_option(10, fused)
        st.set_option(8, adapt)
        st.set_option(4, cands)
        best = None
        every = []
        for rep in range(int(os.environ.get("LAB_REPS", "2"))):
            st.reset_state()
            st.sync()
            st.timing_begin(sample_every=1, max_samples=1024)
            t = time.perf_counter()
            idx, cd, mx = st.kcenters_run(0, K, 0.0)
            dt = time.perf_counter() - t
            kms, ns = st.timing_end()
            every.append("%.4f/%.4f" % (dt, kms))
            if best is None or dt < best[0]:
                best = (dt, kms, ns)
        d, a = st.download_state()
        h = hashlib.sha256(idx.tobytes() + d.tobytes() + a.tobytes()).hexdigest()[:12]
        sums.add(h)
        stats = st.run_stats() if cands != 1 else {}
        print("%-28s form %d adapt %d cands %2d fused %d fine %d sweep %d: %.4f s  %.4f ms/center  pass %.4f ms "
              "(%d samples)  %s  sum %s%s"
              % (name, form, adapt, cands, fused, fine, sweep, best[0], best[0] / K * 1e3, best[1], best[2],
                 {T: pc for T, pc in stats.items() if pc[0]}, h,
                 "  runs (s/pass ms): " + " ".join(every) if len(every) > 2 else ""),
              flush=True)
    print("%-28s checksums agree: %s" % (name, len(sums) == 1), flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["--child"]:
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    args = sys.argv[1:]
    libs = [a for a in args if a.endswith(".so")]

    def opt(name, default):
        return int(args[args.index(name) + 1]) if name in args else default
    n, A, K = opt("--n", 1000000), opt("--atoms", 300), opt("--centers", 2000)
    import numpy as np
    from enspara_amd import synth
    path = "/tmp/lab_frames_%d_%d.npy" % (n, A)
    if not os.path.exists(path):
        np.save(path, synth.synth(n, A, 5000, 1))
    for lib in (libs or [None]):
        env = dict(os.environ)
        if lib:
            env["ENSPARA_HIP_LIB"] = os.path.abspath(lib)
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", path,
                         str(K)], env=env)
