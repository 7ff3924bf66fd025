"""PAM sweep timing on one box (measurement only).

  lab_pam.py [lib.so ...] [--n N] [--atoms A] [--centers K] [--reps R] [--sweeps S] [--walk 1]

For every library given (default: the in-tree build) a child process loads the
same frames, runs k-centers to K centers and then S PAM sweeps
(kmedoids._pam_sweep_device, RandomState(0)), R times from the same start.
Prints seconds per sweep, the wall time spent inside each FrameStore call of
the sweep, and a checksum of medoids + final state (all libraries must agree).
"""
import hashlib
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)

PHASES = ("pam_begin", "pam_count_members_batch", "pam_select_members_batch",
          "pam_prefetch", "pam_window_run", "pam_count_members",
          "pam_propose_member", "pam_commit", "pam_sweep_run")


def child(path, K, reps, sweeps):
    import numpy as np
    if os.environ.get("LAB_TORCH"):         # as in bench.py: torch's runtime alongside
        import torch
        torch.cuda.set_device(0)
        torch.cuda.synchronize()
    from enspara_amd.device import FrameStore
    from enspara_amd.cluster import kmedoids as km
    x = np.load(path, mmap_mode="r")
    st = FrameStore.from_array(np.ascontiguousarray(x))
    name = os.path.basename(os.environ.get("ENSPARA_HIP_LIB", "default"))
    spent = {}

    def timed(fn_name):
        fn = getattr(st, fn_name, None)
        if fn is None:
            return

        def wrap(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                c = spent.setdefault(fn_name, [0.0, 0])
                c[0] += time.perf_counter() - t
                c[1] += 1
        setattr(st, fn_name, wrap)
    for p in PHASES:
        timed(p)
    sums = set()
    # LAB_PAM_OPTS="key=value,..;key=value,.." : a run per option set (ek_set_option),
    # e.g. "16=1;16=0" = the window tables as bounds / exact
    optsets = [dict((int(kv.split("=")[0]), int(kv.split("=")[1]))
                    for kv in o.split(",") if kv)
               for o in os.environ.get("LAB_PAM_OPTS", "").split(";")]
    for rep in range(reps * len(optsets)):
        opts = optsets[rep % len(optsets)]
        for k_, v_ in opts.items():
            st.set_option(k_, v_)
        name = os.path.basename(os.environ.get("ENSPARA_HIP_LIB", "default")) + \
            (" " + str(opts) if opts else "")
        st.reset_state()
        idx, cd, mx = st.kcenters_run(0, K, 0.0)
        med = [int(i) for i in idx]
        med0 = list(med)
        rs = np.random.RandomState(0)
        spent.clear()
        st.sync()
        t = time.perf_counter()
        for _ in range(sweeps):
            med = km._pam_sweep_device(st, med, None, rs)
        st.sync()
        dt = (time.perf_counter() - t) / sweeps
        d, a = st.download_state()
        h = hashlib.sha256(np.asarray(med, dtype=np.int64).tobytes() + d.tobytes() +
                           a.tobytes() + rs.get_state()[1].tobytes()).hexdigest()[:12]
        sums.add(h)
        parts = "  ".join("%s %.3f/%d" % (k.replace("pam_", ""), v[0] / sweeps,
                                          v[1] // sweeps)
                          for k, v in spent.items())
        moved = sum(1 for u, w in zip(med0, med) if u != w)
        ahead = st.pam_ahead_stats() if hasattr(st, "pam_ahead_stats") else 0
        print("%-24s sweep %.4f s  %.1f us/proposal  [%s]  %d of %d medoids moved  "
              "windows %s  slots taken over as evaluated ahead (so far) %d  sum %s"
              % (name, dt, dt / K * 1e6, parts, moved, K, st.pam_sparse_stats(), ahead, h),
              flush=True)
    print("%-24s checksums agree: %s" % (name, len(sums) == 1), flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["--child"]:
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    args = sys.argv[1:]
    libs = [a for a in args if a.endswith(".so")]

    def opt(name, default):
        return int(args[args.index(name) + 1]) if name in args else default
    n, A, K = opt("--n", 1000000), opt("--atoms", 300), opt("--centers", 5000)
    reps, sweeps = opt("--reps", 3), opt("--sweeps", 1)
    import numpy as np
    from enspara_amd import synth
    walk = opt("--walk", 0)                 # one time-ordered trajectory instead of templates
    path = "/tmp/lab_frames_%d_%d%s.npy" % (n, A, "_walk" if walk else "")
    if not os.path.exists(path):
        np.save(path, synth.walk(n, A, seed=1) if walk else synth.synth(n, A, 5000, 1))
    for lib in (libs or [None]):
        env = dict(os.environ)
        if lib:
            env["ENSPARA_HIP_LIB"] = os.path.abspath(lib)
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", path,
                         str(K), str(reps), str(sweeps)], env=env)
