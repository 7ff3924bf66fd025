"""Randomized comparison of the device paths with the oracle over many small
shapes: odd sizes, 1-3 atoms, duplicates, more clusters than frames, cut-offs,
warm starts, cost ties.  (tools/fuzz_gpu.py and fuzz_gpu2.py are the long-running
versions; they found a non-exact shortcut in degenerate geometry and the PAM
cost-tie dependence on the summation order, both fixed.)"""
import numpy as np
import pytest

from enspara_amd import synth

pytestmark = pytest.mark.gpu


def _same(r, inds, a, d):
    return (list(r.center_indices) == [int(i) for i in inds]
            and np.array_equal(r.assignments, a)
            and np.array_equal(r.distances, d))


def test_khybrid_random_small_cases():
    from enspara_amd.cluster import hybrid as hy
    from oracle import cluster as oc
    bad = []
    for case in range(160):
        rng = np.random.RandomState(77_000 + case)
        n = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 255, 256, 257, 1000, 1023]))
        A = int(rng.choice([1, 2, 3, 4, 7, 16, 33]))
        nt = int(rng.choice([1, 2, 5, 40]))
        K = int(rng.choice([1, 2, 3, 8, 9, 17, 40, 90]))
        mode = rng.choice(["count", "cutoff", "both", "dups"])
        x = synth.synth(n, A, nt, seed=int(rng.randint(1 << 30)))
        if mode == "dups" and n > 4:
            x = np.concatenate([x[: max(2, n // 3)]] * 3)[:n]
        cutoff, nclu = None, K
        if mode == "cutoff":
            nclu, cutoff = None, float(rng.uniform(0.05, 0.6))
        elif mode == "both":
            cutoff = float(rng.uniform(0.02, 0.3))
        sweeps = int(rng.choice([0, 1, 2]))
        seed = int(rng.randint(1000))
        inds, a, d = oc.kcenters(x, n_clusters=nclu, dist_cutoff=cutoff)
        if len(inds) > 200:
            sweeps = 0
        kw = dict(n_iters=sweeps,
                  n_clusters=(nclu if nclu is not None else np.inf),
                  dist_cutoff=(cutoff if cutoff is not None else 0))
        try:
            rs = np.random.RandomState(seed)
            wi, wd, wa = list(inds), d.copy(), a.copy()
            for _ in range(sweeps):
                wi, wd, wa = oc.pam_update(x, wi, wa, wd, random_state=rs)
        except ValueError:          # an empty cluster: both sides must raise
            with pytest.raises(ValueError):
                hy.hybrid(x, "rmsd", random_state=np.random.RandomState(seed),
                          **kw)
            continue
        r = hy.hybrid(x, "rmsd", random_state=np.random.RandomState(seed), **kw)
        if not _same(r, wi, wa, wd):
            bad.append((case, n, A, nt, nclu, cutoff, mode, sweeps))
    assert not bad, bad


def test_assign_and_warm_start_random_small_cases():
    from enspara_amd.cluster import kcenters as kc
    from enspara_amd.device import FrameStore
    from oracle import cluster as oc
    bad = []
    for case in range(160):
        rng = np.random.RandomState(91_000 + case)
        n = int(rng.choice([1, 3, 63, 64, 65, 255, 257, 1000, 1025]))
        A = int(rng.choice([1, 2, 3, 5, 16, 33]))
        x = synth.synth(n, A, int(rng.choice([1, 3, 40])),
                        seed=int(rng.randint(1 << 30)))
        if case % 2 == 0:
            K = int(rng.choice([1, 2, 23, 24, 25, 64, 70, 200]))
            ctr = (x[rng.randint(0, n, size=K)] if rng.rand() < 0.5
                   else synth.synth(K, A, K, seed=case))
            a, d = oc.assign_to_nearest_center(x, ctr)
            for variant in (1, 2, 3):
                with FrameStore.from_array(x) as st:
                    st.set_option(2, variant)
                    st.assign_nearest(ctr)
                    dd, aa = st.download_state()
                if not (np.array_equal(aa, a)
                        and np.array_equal(dd.astype(np.float64), d)):
                    bad.append(("assign", case, n, A, K, variant))
        else:
            K0 = int(rng.choice([1, 2, 5]))
            K = K0 + int(rng.choice([0, 1, 7, 30]))
            init = [x[int(i)] for i in rng.randint(0, n, size=K0)]
            inds, a, d = oc.kcenters(x, n_clusters=K, init_centers=init)
            r = kc.kcenters(x, "rmsd", n_clusters=K, init_centers=init)
            if not _same(r, inds, a, d):
                bad.append(("warm", case, n, A, K0, K))
    assert not bad, bad
