"""Scratch: run bench.main() with the PAM section under cProfile."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--no-cpu-baseline", "--pam-sweeps", "1"] + sys.argv[1:]
import bench
from enspara_amd.cluster import kmedoids as km
orig = km._pam_sweep_device
def wrapped(*a, **k):
    pr = cProfile.Profile(); pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
        sys.stderr.write(s.getvalue())
km._pam_sweep_device = wrapped
bench.main()
