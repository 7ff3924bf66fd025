"""The multi-rank k-centers driver (enspara_amd/sharded.py) with world_size 2
on CPU: backend gloo, checker-backed shards (tests/_host_shard.py).  The
sharded result must equal the single-process result exactly -- the property
the reference's own MPI tests assert (enspara/test/test_cluster.py:241-275,
:278-314)."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, A, seed, n_clusters, cutoff, outdir,
            cands=1, chain=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2" if world <= 3 else "1"
    from enspara_amd import sharded, synth
    from _host_shard import (HostShard, HostShardRounds, HostShardChain,
                             HostShardMs)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    x = synth.synth(n, A, 9, seed=seed)
    lo, cnt = sharded.shard_bounds(n, world, rank)
    shard = (HostShard(x[lo:lo + cnt], lo) if cands == 1 else
             (HostShardMs if chain == "ms" else
              HostShardChain if chain else HostShardRounds)(
                 x[lo:lo + cnt], lo, cands))
    max_new = n_clusters if n_clusters else n
    idx, cd = sharded.kcenters_sharded(shard, 0, max_new, cutoff,
                                       check_every=4)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), idx=idx, cd=cd, lo=lo,
             dist=shard.dist, assign=shard.assign)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n, A, seed, n_clusters, cutoff, cands=1, chain=False):
    from oracle import cluster as oc
    from enspara_amd import synth
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), n, A, seed, n_clusters,
                                cutoff, d, cands, chain), nprocs=world,
                 join=True)
        parts = [np.load(os.path.join(d, "r%d.npz" % r)) for r in range(world)]
    x = synth.synth(n, A, 9, seed=seed)
    inds, a, dd = oc.kcenters(x, n_clusters=n_clusters,
                              dist_cutoff=cutoff if cutoff else None)
    for p in parts:                      # every rank reports the same centers
        np.testing.assert_array_equal(p["idx"], np.array(inds))
    got_a = np.concatenate([p["assign"] for p in parts])
    got_d = np.concatenate([p["dist"] for p in parts])
    np.testing.assert_array_equal(got_a, a)
    np.testing.assert_array_equal(got_d.astype(np.float64), dd)
    return inds


def test_two_ranks_fixed_count():
    inds = _run(2, 1500, 20, 5, 11, 0.0)
    assert len(inds) == 11


def test_two_ranks_cutoff():
    inds = _run(2, 1200, 15, 6, None, 0.45)
    assert len(inds) > 2


def test_three_ranks_one_empty():
    # 2 tiles over 3 ranks: the last rank owns no frames
    inds = _run(3, 500, 10, 7, 6, 0.0)
    assert len(inds) == 6


def test_two_ranks_multi_candidate_rounds():
    inds = _run(2, 1500, 20, 5, 23, 0.0, cands=4)
    assert len(inds) == 23


def test_two_ranks_multi_candidate_rounds_cutoff():
    inds = _run(2, 1200, 15, 6, None, 0.45, cands=8)
    assert len(inds) > 2


def test_three_ranks_one_empty_multi_candidate():
    inds = _run(3, 500, 10, 7, 9, 0.0, cands=4)
    assert len(inds) == 9


def test_two_ranks_chained_rounds():
    inds = _run(2, 1500, 20, 5, 23, 0.0, cands=8, chain=True)
    assert len(inds) == 23


def test_two_ranks_chained_rounds_cutoff():
    inds = _run(2, 1200, 15, 6, None, 0.45, cands=8, chain=True)
    assert len(inds) > 2


def test_three_ranks_one_empty_chained_rounds():
    inds = _run(3, 500, 10, 7, 9, 0.0, cands=4, chain=True)
    assert len(inds) == 9


@pytest.mark.parametrize("world,n,K", [
    (4, 2600, 40),      # 4 x 8 = 32 records on offer per round
    (8, 4200, 60),      # 8 x 8 = 64: exactly the plan's table
    (10, 5200, 50),     # more ranks than 64 / 8: every rank offers 6 records
])
def test_larger_groups_chained_rounds(world, n, K):
    """BASELINE.json configs[3] runs 8 ranks; the round protocol with 8
    candidates per pass there sits at the plan's limit of 64 records on offer,
    and larger groups offer fewer per rank instead of leaving the protocol."""
    inds = _run(world, n, 12, 31 + world, K, 0.0, cands=8, chain=True)
    assert len(inds) == K


@pytest.mark.parametrize("world,n,K,cutoff,cands", [
    (2, 1500, 40, 0.0, 8), (2, 1200, None, 0.45, 8), (3, 300, 12, 0.0, 16),
    (8, 2600, 50, 0.0, 16),
    # rounds of 32 candidates: two passes behind one exchange (round 5)
    (2, 1500, 70, 0.0, 32), (2, 1200, None, 0.45, 32), (3, 300, 40, 0.0, 32),
    (8, 2600, 90, 0.0, 32)])
def test_one_exchange_per_round(world, n, K, cutoff, cands):
    """the round protocol of csrc/ek_mshard.hip (one message per shard and
    round: per-prefix maxima + speculative offers, a broken chain re-offered)
    through the product's driver loop, gather transport over gloo; world 3:
    one shard is empty"""
    inds = _run(world, n, 20 if world < 8 else 10, 5, K, cutoff, cands=cands,
                chain="ms")
    if K:
        assert len(inds) == K


def test_shard_bounds():
    from enspara_amd.sharded import shard_bounds
    for n, w in [(1000, 2), (1_000_000, 8), (255, 4), (256 * 7 + 3, 3)]:
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0
        assert sum(c for _, c in spans) == n
        for (lo, c), (lo2, _) in zip(spans, spans[1:]):
            assert lo + c == lo2
            assert lo2 % 256 == 0 or lo2 == n


def test_single_process_without_process_group():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _host_shard import (HostShard, HostShardRounds, HostShardChain,
                             HostShardMs)
    from enspara_amd import sharded, synth
    from oracle import cluster as oc
    x = synth.synth(700, 12, 5, seed=2)
    inds, a, d = oc.kcenters(x, n_clusters=8)
    for sh in (HostShard(x, 0), HostShardRounds(x, 0, 4),
               HostShardChain(x, 0, 8)):
        idx, cd = sharded.kcenters_sharded(sh, 0, 8, 0.0)
        np.testing.assert_array_equal(idx, np.array(inds))
        np.testing.assert_array_equal(sh.assign, a)
        np.testing.assert_array_equal(sh.dist.astype(np.float64), d)


@pytest.mark.parametrize("cls_name,cands", [("HostShardRounds", 4),
                                            ("HostShardChain", 8)])
def test_rounds_only_compare_kept_distances(cls_name, cands):
    """The device pass keeps +inf instead of a candidate's distance wherever that
    distance is not below the frame's current one (ek_rmsd_from_S_below stops
    the quartic early).  The round protocol only ever asks "is it smaller": the
    same run with such vectors gives the same centers, labels and distances."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _host_shard
    from enspara_amd import sharded, synth
    from oracle import cluster as oc
    for seed, n, K in ((2, 700, 40), (5, 1500, 120)):
        x = synth.synth(n, 12, 7, seed=seed)
        inds, a, d = oc.kcenters(x, n_clusters=K)
        sh = getattr(_host_shard, cls_name)(x, 0, cands)
        sh.far_as_inf = True
        idx, cd = sharded.kcenters_sharded(sh, 0, K, 0.0)
        np.testing.assert_array_equal(idx, np.array(inds))
        np.testing.assert_array_equal(sh.assign, a)
        np.testing.assert_array_equal(sh.dist.astype(np.float64), d)


# ---- PAM sweeps / k-hybrid across ranks -------------------------------------
def _hybrid_worker(rank, world, port, n, A, seed, K, n_iters, outdir, cands,
                   explicit, prefetch):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    from enspara_amd import sharded, synth
    from _host_shard import HostShard, HostShardRounds
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    x = synth.synth(n, A, 9, seed=seed)
    lo, cnt = sharded.shard_bounds(n, world, rank)
    shard = (HostShard(x[lo:lo + cnt], lo) if cands == 1 else
             HostShardRounds(x[lo:lo + cnt], lo, cands))
    if explicit:
        idx, _ = sharded.kcenters_sharded(shard, 0, K, 0.0)
        props = np.random.RandomState(seed).randint(0, n, size=K)
        med = sharded.pam_sweep_sharded(shard, list(idx), proposals=props,
                                        prefetch=prefetch)
    else:
        med = sharded.khybrid_sharded(shard, K, 0.0, n_iters, random_state=4)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), med=np.array(med), lo=lo,
             dist=shard.dist, assign=shard.assign)
    dist.barrier()
    dist.destroy_process_group()


def _run_hybrid(world, n, A, seed, K, n_iters, cands=1, explicit=False,
                prefetch=8):
    from oracle import cluster as oc
    from enspara_amd import synth
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_hybrid_worker,
                 args=(world, _free_port(), n, A, seed, K, n_iters, d, cands,
                       explicit, prefetch), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, "r%d.npz" % r)) for r in range(world)]
    x = synth.synth(n, A, 9, seed=seed)
    inds, a, dd = oc.kcenters(x, n_clusters=K)
    if explicit:
        props = np.random.RandomState(seed).randint(0, n, size=K)
        inds, dd, a = oc.pam_update(x, inds, a, dd, proposals=props)
    else:
        rs = np.random.RandomState(4)
        for _ in range(n_iters):
            inds, dd, a = oc.pam_update(x, inds, a, dd, random_state=rs)
    for p in parts:
        np.testing.assert_array_equal(p["med"], np.array(inds))
    np.testing.assert_array_equal(
        np.concatenate([p["assign"] for p in parts]), a)
    np.testing.assert_array_equal(
        np.concatenate([p["dist"] for p in parts]).astype(np.float64), dd)


def test_two_ranks_khybrid():
    _run_hybrid(2, 1500, 15, 5, 12, 2)


def test_three_ranks_one_empty_khybrid_rounds():
    _run_hybrid(3, 500, 10, 7, 6, 2, cands=4)


def test_two_ranks_pam_explicit_proposals():
    _run_hybrid(2, 1200, 12, 3, 20, 1, explicit=True, prefetch=3)


def test_pam_sweep_without_process_group():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _host_shard import HostShard
    from enspara_amd import sharded, synth
    from oracle import cluster as oc
    x = synth.synth(900, 12, 6, seed=12)
    inds, a, d = oc.kcenters(x, n_clusters=30)
    for width in (1, 8):
        sh = HostShard(x, 0)
        idx, _ = sharded.kcenters_sharded(sh, 0, 30, 0.0)
        rs_a, rs_b = np.random.RandomState(9), np.random.RandomState(9)
        med, wi, wd, wa = list(idx), list(inds), d.copy(), a.copy()
        for _ in range(2):
            med = sharded.pam_sweep_sharded(sh, med, random_state=rs_a,
                                            prefetch=width)
            wi, wd, wa = oc.pam_update(x, wi, wa, wd, random_state=rs_b)
        np.testing.assert_array_equal(med, wi)
        np.testing.assert_array_equal(sh.assign, wa)
        np.testing.assert_array_equal(sh.dist.astype(np.float64), wd)


# ---- warm start across ranks (reference kcenters.py:200-213 in MPI mode) --------------
def _warm_worker(rank, world, port, n, A, seed, n_clusters, init_frames, outdir,
                 cands, chain):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    from enspara_amd import sharded, synth
    from _host_shard import HostShard, HostShardMs
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    x = synth.synth(n, A, 9, seed=seed)
    init = _init_centers(x, init_frames)
    lo, cnt = sharded.shard_bounds(n, world, rank)
    shard = (HostShard(x[lo:lo + cnt], lo) if cands == 1 else
             HostShardMs(x[lo:lo + cnt], lo, cands))
    ctr = sharded.warm_start_sharded(shard, init)
    idx, cd = sharded.kcenters_sharded(shard, len(ctr), n_clusters - len(ctr), 0.0,
                                       check_every=4)
    np.savez(os.path.join(outdir, "r%d.npz" % rank),
             idx=np.array(ctr + [int(i) for i in idx]), lo=lo, dist=shard.dist,
             assign=shard.assign)
    dist.barrier()
    dist.destroy_process_group()


def _init_centers(x, init_frames):
    """frames of the data, one of them twice (the second copy attracts nothing:
    a label without members, kcenters.py:233), and a structure far from all"""
    init = [x[j] for j in init_frames] + [x[init_frames[0]]]
    init.append((x[0] * 7.0 + 3.0).astype(np.float32))
    return init


@pytest.mark.parametrize("world,cands", [(2, 1), (2, 8), (3, 16)])
def test_warm_start_across_ranks(world, cands):
    """every rank assigns its frames to the initial centers, the occupied
    labels' closest members are found over all ranks (first in the global
    order), the sharded loop continues from there: equal to the single-process
    oracle, including a label that attracts no frame"""
    from oracle import cluster as oc
    from enspara_amd import synth
    n, A, seed, K = 1800, 25, 21, 12
    init_frames = [5, 1000, 700]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_warm_worker, args=(world, _free_port(), n, A, seed, K, init_frames,
                                     d, cands, "ms"), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, "r%d.npz" % r)) for r in range(world)]
    x = synth.synth(n, A, 9, seed=seed)
    inds, a, dd = oc.kcenters(x, n_clusters=K, init_centers=_init_centers(x, init_frames))
    assert len(inds) == K
    for p in parts:
        np.testing.assert_array_equal(p["idx"], np.array(inds))
    np.testing.assert_array_equal(np.concatenate([p["assign"] for p in parts]), a)
    np.testing.assert_array_equal(
        np.concatenate([p["dist"] for p in parts]).astype(np.float64), dd)


# ---- connect_mailboxes is a collective every rank completes, whatever fails where -------
class _FakeStore:
    """stands for a FrameStore in connect_mailboxes: exports a handle, opens the
    peers' -- and fails where told to"""

    def __init__(self, rank, fail_export_on, fail_open_on):
        self.rank, self.fe, self.fo = rank, fail_export_on, fail_open_on
        self.setups, self.connected = 0, []

    def ms_setup(self, world, rank):
        self.setups += 1
        self.connected = []

    def ms_mailbox(self, ipc=False):
        if self.rank == self.fe:
            raise RuntimeError("no IPC handle on rank %d" % self.rank)
        return ("handle", self.rank)

    def ms_connect(self, p, ipc=None):
        if self.rank == self.fo and p != self.rank:
            raise RuntimeError("no peer access from rank %d to %d" % (self.rank, p))
        self.connected.append(p)


class _FakeShard:
    def __init__(self, store):
        self.store = store
        self.ms_connected = -1


def _connect_worker(rank, world, port, fail_export_on, fail_open_on, outdir):
    sys.path.insert(0, ROOT)
    from enspara_amd import sharded
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    sh = _FakeShard(_FakeStore(rank, fail_export_on, fail_open_on))
    ok = sharded.connect_mailboxes(sh)
    # the next collective every rank makes must still line up
    t = torch.tensor([rank])
    dist.all_reduce(t)
    np.savez(os.path.join(outdir, "c%d.npz" % rank), ok=ok, connected=sh.ms_connected,
             setups=sh.store.setups, n_conn=len(sh.store.connected),
             had_error=sh.ms_connect_error is not None, total=int(t.item()))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_export_on,fail_open_on", [(-1, -1), (1, -1), (-1, 2), (0, 1)])
def test_connect_mailboxes_is_collective_whatever_fails(fail_export_on, fail_open_on):
    """round-3 advisor: a rank whose hipIpc export / open failed skipped the
    trailing barrier and left the healthy ranks in it.  Now every rank makes the
    same two gathers, all ranks get the same verdict, a failed group has closed
    what it opened (ms_setup again) and stays on the all-gather transport."""
    world = 3
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_connect_worker, args=(world, _free_port(), fail_export_on,
                                        fail_open_on, d), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, "c%d.npz" % r)) for r in range(world)]
    healthy = fail_export_on < 0 and fail_open_on < 0
    for r, p in enumerate(parts):
        assert bool(p["ok"]) == healthy
        assert int(p["connected"]) == (world if healthy else 0)
        assert int(p["total"]) == 0 + 1 + 2
        if healthy:
            assert int(p["n_conn"]) == world and int(p["setups"]) == 1
        else:
            assert int(p["setups"]) >= 1 and int(p["n_conn"]) == 0
    if not healthy:
        bad = {fail_export_on, fail_open_on} - {-1}
        assert all(bool(parts[r]["had_error"]) for r in bad)


# ---- k-medoids with a warm start across ranks (reference kmedoids.py MPI mode) --------
_KM_LENGTHS = [300, 120, 260, 200, 180]


def _striped(x, lengths, world):
    """the reference's distribution (mpi/io.py:126): trajectory t on rank
    t % world, a rank's trajectories one after the other -> per rank the frame
    indices (into the concatenation of all trajectories) it holds, in order"""
    starts = np.concatenate([[0], np.cumsum(lengths)])
    return [np.concatenate([np.arange(starts[t], starts[t + 1])
                            for t in range(r, len(lengths), world)] or
                           [np.zeros(0, dtype=np.int64)]).astype(np.int64)
            for r in range(world)]


def _km_worker(rank, world, port, seed, K, n_iters, form, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    from enspara_amd import sharded, synth
    from _host_shard import HostShard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    inp = np.load(os.path.join(outdir, "in.npz"))
    x = synth.synth(sum(_KM_LENGTHS), 18, 11, seed=seed)
    held = _striped(x, _KM_LENGTHS, world)
    lo = sum(len(h) for h in held[:rank])
    mine = x[held[rank]]
    shard = HostShard(mine, lo)
    ctr = inp["flat"] if form == "flat" else [tuple(p) for p in inp["pairs"]]
    a = inp["a"][lo:lo + len(mine)]
    d = inp["d"][lo:lo + len(mine)]
    pairs, coords = sharded.kmedoids_sharded(
        shard, mine, n_iters=n_iters, assignments=a, distances=d,
        cluster_center_inds=list(ctr), X_lengths=_KM_LENGTHS,
        random_state=np.random.RandomState(4))
    np.savez(os.path.join(outdir, "r%d.npz" % rank), pairs=np.array(pairs),
             coords=coords, dist=shard.dist, assign=shard.assign, lo=lo)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,form", [(2, "flat"), (2, "pairs"), (3, "flat"),
                                        (3, "pairs")])
def test_kmedoids_warm_start_across_ranks(world, form):
    """kmedoids() in MPI mode (kmedoids.py:133-146, :264-283, :365-407): every
    rank passes its frames' assignments and distances and the cluster centers
    with respect to ALL data -- flat indices or (trajectory, frame) pairs with
    X_lengths -- the ranks holding the trajectories striped; two PAM sweeps;
    equal to the single-process oracle on the frames in the ranks' order"""
    from oracle import cluster as oc
    from enspara_amd import synth
    seed, K, n_iters = 33, 9, 2
    x = synth.synth(sum(_KM_LENGTHS), 18, 11, seed=seed)
    held = _striped(x, _KM_LENGTHS, world)
    perm = np.concatenate(held)             # rank order -> original index
    xp = x[perm]
    inds, a, d = oc.kcenters(xp, n_clusters=K)
    flat = perm[[int(i) for i in inds]]     # the centers, w.r.t. all data
    starts = np.concatenate([[0], np.cumsum(_KM_LENGTHS)])
    traj = np.searchsorted(starts, flat, side="right") - 1
    pairs = np.stack([traj, flat - starts[traj]], axis=1)
    rs = np.random.RandomState(4)
    wi, wd, wa = [int(i) for i in inds], d.copy(), a.copy()
    for _ in range(n_iters):
        wi, wd, wa = oc.pam_update(xp, wi, wa, wd, random_state=rs)
    with tempfile.TemporaryDirectory() as dd:
        np.savez(os.path.join(dd, "in.npz"), flat=flat, pairs=pairs, a=a, d=d)
        mp.spawn(_km_worker, args=(world, _free_port(), seed, K, n_iters, form, dd),
                 nprocs=world, join=True)
        parts = [np.load(os.path.join(dd, "r%d.npz" % r)) for r in range(world)]
    los = [int(p["lo"]) for p in parts]
    for p in parts:                         # (rank, local index) pairs
        got = [los[int(r)] + int(i) for r, i in p["pairs"]]
        assert got == [int(i) for i in wi]
        np.testing.assert_array_equal(p["coords"], xp[[int(i) for i in wi]])
    np.testing.assert_array_equal(np.concatenate([p["assign"] for p in parts]), wa)
    np.testing.assert_array_equal(
        np.concatenate([p["dist"] for p in parts]).astype(np.float64), wd)


def test_ctr_ids_mpi_is_the_references_mapping():
    """sharded.ctr_ids_mpi against the reference's own construction
    (kmedoids.py:365-407: ragged arrays of global and per-rank indices)"""
    from enspara_amd import sharded
    lengths = [5, 3, 7, 2, 6, 4]
    starts = np.concatenate([[0], np.cumsum(lengths)])
    for world in (1, 2, 3, 4):
        want = {}
        for t, L in enumerate(lengths):
            r = t % world
            owned = lengths[r::world]
            base = sum(owned[:t // world])
            for f in range(L):
                want[(t, f)] = (r, base + f)
        pairs = sorted(want)
        assert sharded.ctr_ids_mpi(pairs, lengths, world) == [want[p] for p in pairs]
        flat = [int(starts[t] + f) for t, f in pairs]
        assert sharded.ctr_ids_mpi(flat, lengths, world) == [want[p] for p in pairs]
    with pytest.raises(IndexError):
        sharded.ctr_ids_mpi([(1, 3)], lengths, 2)
    with pytest.raises(IndexError):
        sharded.ctr_ids_mpi([27], lengths, 2)


def test_kmedoids_mpi_mode_input_errors():
    """the half-supplied cases of _kmedoids_inputs_tree_mpi (kmedoids.py:277-281)
    and of kmedoids() itself (:156-166), without a process group"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _host_shard import HostShard
    from enspara_amd import sharded, synth
    from enspara_amd.exception import DataInvalid, ImproperlyConfigured
    from enspara_amd.cluster.kmedoids import kmedoids
    x = synth.synth(300, 8, 4, seed=2)
    a = np.zeros(300, dtype=np.int64)
    d = np.zeros(300)
    for kw in (dict(assignments=a), dict(distances=d), dict(cluster_center_inds=[0]),
               dict(assignments=a, distances=d),
               dict(assignments=a, cluster_center_inds=[0])):
        with pytest.raises(ImproperlyConfigured, match="can start from scratch"):
            sharded.kmedoids_sharded(HostShard(x, 0), x, X_lengths=[300], **kw)
    with pytest.raises(ImproperlyConfigured, match="X_lengths"):
        sharded.kmedoids_sharded(HostShard(x, 0), x, assignments=a, distances=d,
                                 cluster_center_inds=[0])
    with pytest.raises(DataInvalid, match="holds 300 frames"):
        sharded.kmedoids_sharded(HostShard(x, 0), x, assignments=a, distances=d,
                                 cluster_center_inds=[0], X_lengths=[100, 100])
    with pytest.raises(ImproperlyConfigured, match="in MPI mode"):
        sharded.kmedoids_sharded(HostShard(x, 0), x)
    with pytest.raises(ImproperlyConfigured, match="in MPI mode"):
        kmedoids(x, "rmsd", mpi_mode=True)
    with pytest.raises(ImproperlyConfigured, match="X_lengths also needs"):
        kmedoids(x, "rmsd", cluster_center_inds=[(0, 1)], mpi_mode=True)


# ---- the ranks agree on the width of their rounds (round-4 advisor finding) -----------
def _form_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    from enspara_amd import sharded, synth
    from _host_shard import HostShardMs
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)

    class Tight(HostShardMs):
        """a shard whose rank 1 has no room for the quad copy of its frames"""
        pinned = None
        pins = ()

        def quad_copy_ready(self):
            return rank != 1

        def candidates_option(self):
            return 16

        def pin_candidates(self, T):
            self.pins = self.pins + (T,)
            if self.pinned is None:
                self.pinned = T
            self.candidates = T

    x = synth.synth(1500, 20, 9, seed=5)
    lo, cnt = sharded.shard_bounds(len(x), world, rank)
    shard = Tight(x[lo:lo + cnt], lo, 16)
    idx, _ = sharded.kcenters_sharded(shard, 0, 30, 0.0)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), idx=idx,
             pinned=-1 if shard.pinned is None else shard.pinned,
             pins=np.array(shard.pins), cands=shard.candidates)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_the_width_of_their_rounds():
    """one rank without memory for the quad copy: EVERY rank pins its rounds to 8
    candidates before anything runs (sharded._agree_on_form), instead of that
    rank alone narrowing its rounds beside peers that plan sixteen"""
    from oracle import cluster as oc
    from enspara_amd import synth
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_form_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        parts = [np.load(os.path.join(d, "r%d.npz" % r)) for r in range(2)]
    x = synth.synth(1500, 20, 9, seed=5)
    inds, _, _ = oc.kcenters(x, n_clusters=30)
    for p in parts:
        # narrowed for the run, and the caller's own setting back after it
        assert int(p["pinned"]) == 8 and list(p["pins"]) == [8, 16]
        assert int(p["cands"]) == 16
        np.testing.assert_array_equal(p["idx"], np.array(inds))


def _km_none_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    from enspara_amd import sharded, synth
    from _host_shard import HostShard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    x = synth.synth(900, 18, 11, seed=8)
    lo, cnt = sharded.shard_bounds(len(x), world, rank)
    shard = HostShard(x[lo:lo + cnt], lo)
    np.random.seed(1000 + rank)         # (nothing shared by accident)
    pairs, coords = sharded.kmedoids_sharded(shard, x[lo:lo + cnt], n_clusters=7,
                                             n_iters=3, random_state=None)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), pairs=np.array(pairs),
             coords=coords, dist=shard.dist, assign=shard.assign, lo=lo)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_kmedoids_across_ranks_without_a_random_state(world):
    """random_state=None (what KMedoids.fit passes in MPI mode): the ranks must
    still draw the same proposals -- the reference agrees on every draw across
    ranks (mpi/ops.py randind).  Round-5 advisor finding: each rank seeded its
    own stream and the medoids diverged.  Here: identical medoids on every
    rank, owned rows consistent, every frame's distance the RMSD to the medoid
    its label names."""
    from oracle import qcp
    from enspara_amd import synth
    with tempfile.TemporaryDirectory() as dd:
        mp.spawn(_km_none_worker, args=(world, _free_port(), dd), nprocs=world,
                 join=True)
        parts = [np.load(os.path.join(dd, "r%d.npz" % r)) for r in range(world)]
    x = synth.synth(900, 18, 11, seed=8)
    los = [int(p["lo"]) for p in parts]
    med = [los[int(r)] + int(i) for r, i in parts[0]["pairs"]]
    assert len(set(med)) == 7
    for p in parts:
        assert [los[int(r)] + int(i) for r, i in p["pairs"]] == med
        np.testing.assert_array_equal(p["coords"], x[med])
    a = np.concatenate([p["assign"] for p in parts])
    d = np.concatenate([p["dist"] for p in parts])
    for k, g in enumerate(med):
        sel = np.flatnonzero(a == k)
        np.testing.assert_array_equal(
            d[sel].astype(np.float32),
            qcp.rmsd(x, x[int(g)])[sel].astype(np.float32))
