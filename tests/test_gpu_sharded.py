"""The product shard (enspara_amd.sharded.DeviceShard: torch tensors + the C
ABI) under the multi-rank drivers, on the one GPU of the test box: without a
process group, and as a 1-rank RCCL job in a child process.  Results must
equal the oracle's single-process k-centers / PAM exactly.  (world_size > 1 is
covered on CPU by tests/test_sharded_gloo.py with the same drivers.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from enspara_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expected(x, K, n_iters, seed, cutoff=None):
    from oracle import cluster as oc
    inds, a, d = oc.kcenters(x, n_clusters=K, dist_cutoff=cutoff)
    rs = np.random.RandomState(seed)
    for _ in range(n_iters):
        inds, d, a = oc.pam_update(x, inds, a, d, random_state=rs)
    return inds, a, d


@pytest.mark.parametrize("n,A,K,prefetch", [(3000, 20, 40, 8), (1500, 9, 6, 3),
                                            (2000, 33, 25, 1)])
def test_device_shard_khybrid_no_group(n, A, K, prefetch):
    import torch
    from enspara_amd import sharded
    from enspara_amd.device import FrameStore
    x = synth.synth(n, A, 7, seed=n)
    ts = torch.cuda.Stream(device=0)
    with FrameStore(n, A, device=0, stream=ts.cuda_stream) as st:
        st.load(x)
        st.reset_state()
        sh = sharded.DeviceShard(st)
        with torch.cuda.stream(ts):
            idx, _ = sharded.kcenters_sharded(sh, 0, K, 0.0)
            med = [int(i) for i in idx]
            rs = np.random.RandomState(3)
            for _ in range(2):
                med = sharded.pam_sweep_sharded(sh, med, random_state=rs,
                                                prefetch=prefetch)
        d, a = st.download_state()
    inds, wa, wd = _expected(x, K, 2, 3)
    np.testing.assert_array_equal(med, inds)
    np.testing.assert_array_equal(a, wa)
    np.testing.assert_array_equal(d.astype(np.float64), wd)


def test_device_shard_explicit_proposals_match_single_gpu_sweep():
    import torch
    from enspara_amd import sharded
    from enspara_amd.cluster import kmedoids as km
    from enspara_amd.device import FrameStore
    n, A, K = 2500, 16, 30
    x = synth.synth(n, A, 5, seed=77)
    props = [int(p) for p in np.random.RandomState(1).randint(0, n, size=K)]
    with FrameStore.from_array(x) as st:
        st.reset_state()
        idx, _, _ = st.kcenters_run(0, K, 0.0)
        want = km._pam_sweep_device(st, [int(i) for i in idx], props, None)
        wd, wa = st.download_state()
    ts = torch.cuda.Stream(device=0)
    with FrameStore(n, A, device=0, stream=ts.cuda_stream) as st:
        st.load(x)
        st.reset_state()
        sh = sharded.DeviceShard(st)
        with torch.cuda.stream(ts):
            idx2, _ = sharded.kcenters_sharded(sh, 0, K, 0.0)
            got = sharded.pam_sweep_sharded(sh, list(idx2), proposals=props)
        d, a = st.download_state()
    np.testing.assert_array_equal(idx2, idx)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(a, wa)
    np.testing.assert_array_equal(d, wd)


_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
from enspara_amd import sharded, synth
from enspara_amd.device import FrameStore
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[3])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
n, A, K = 3000, 20, 40
x = synth.synth(n, A, 7, seed=n)
ts = torch.cuda.Stream(device=0)
with FrameStore(n, A, device=0, stream=ts.cuda_stream) as st:
    st.load(x)
    st.reset_state()
    sh = sharded.DeviceShard(st)
    with torch.cuda.stream(ts):
        med = sharded.khybrid_sharded(sh, K, 0.0, 2, random_state=3)
    d, a = st.download_state()
np.savez(sys.argv[2], med=np.array(med), d=d, a=a)
dist.barrier()
dist.destroy_process_group()
"""


def test_device_shard_khybrid_one_rank_rccl(tmp_path):
    out = str(tmp_path / "r.npz")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", _CHILD, ROOT, out, "29613"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    r = np.load(out)
    x = synth.synth(3000, 20, 7, seed=3000)
    inds, wa, wd = _expected(x, 40, 2, 3)
    np.testing.assert_array_equal(r["med"], inds)
    np.testing.assert_array_equal(r["a"], wa)
    np.testing.assert_array_equal(r["d"].astype(np.float64), wd)


# Two ranks, each with its own FrameStore on the one GPU of the test box.  RCCL
# refuses two ranks on one device, so the collectives of this test go through
# gloo with the payload staged in host memory (patched in the child only); what
# is under test is the device side of the multi-shard protocol: candidate
# records and chained rounds across shards, the PAM table / proposal exchange,
# per-rank propose + the gathered decision.
_CHILD2 = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port,
                        rank=rank, world_size=world)
_agit, _ar = dist.all_gather_into_tensor, dist.all_reduce
def agit(out_t, in_t, group=None):
    torch.cuda.current_stream().synchronize()
    o = torch.empty(out_t.shape, dtype=out_t.dtype)
    _agit(o, in_t.cpu(), group=group)
    out_t.copy_(o)
def ar(t, op=dist.ReduceOp.SUM, group=None):
    torch.cuda.current_stream().synchronize()
    h = t.cpu()
    _ar(h, op=op, group=group)
    t.copy_(h)
dist.all_gather_into_tensor, dist.all_reduce = agit, ar
from enspara_amd import sharded, synth
from enspara_amd.device import FrameStore
n, A, K, tmpl, iters, cands = [int(v) for v in sys.argv[6:12]]
mailboxes = len(sys.argv) > 12 and sys.argv[12] == "ipc"
x = synth.synth(n, A, tmpl, seed=21)
lo, cnt = sharded.shard_bounds(n, world, rank)
torch.cuda.set_device(0)
ts = torch.cuda.Stream(device=0)
with FrameStore(cnt, A, device=0, global_offset=lo, stream=ts.cuda_stream) as st:
    st.load(x[lo:lo + cnt])
    st.set_option(4, cands)
    st.reset_state()
    sh = sharded.DeviceShard(st)
    if mailboxes:       # the k-centers rounds exchange on the device (hipIpc)
        sharded.connect_mailboxes(sh)
    with torch.cuda.stream(ts):
        med = sharded.khybrid_sharded(sh, K, 0.0, iters, random_state=5)
    d, a = st.download_state()
np.savez(out + ".%d.npz" % rank, med=np.array(med), d=d, a=a, lo=lo)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("n,A,K,tmpl,iters,cands,transport", [
    (6000, 14, 45, 9, 2, 8, "gather"),
    # the shape of BASELINE.json configs[3] as far as one GPU allows: 500 atoms,
    # two shards, ~200 centers -- rounds of 16 and of 8 candidates
    (40000, 500, 200, 300, 0, 16, "gather"),
    (40000, 500, 200, 300, 0, 8, "gather"),
    # the same two processes with their mailboxes mapped into each other
    # (hipIpc): the exchange of a round happens on the device
    (6000, 14, 45, 9, 2, 16, "ipc"),
    (40000, 500, 200, 300, 0, 16, "ipc"),
    # rounds of 32 candidates, both transports
    (6000, 14, 80, 9, 1, 32, "gather"),
    (6000, 14, 80, 9, 1, 32, "ipc"),
    (40000, 500, 200, 300, 0, 32, "ipc"),
])
def test_two_device_shards_on_one_gpu(tmp_path, n, A, K, tmpl, iters, cands,
                                      transport):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    out = str(tmp_path / "r")
    procs = [subprocess.Popen([sys.executable, "-c", _CHILD2, ROOT, str(r), "2",
                               port, out, str(n), str(A), str(K), str(tmpl),
                               str(iters), str(cands), transport],
                              stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    parts = [np.load(out + ".%d.npz" % r) for r in range(2)]
    x = synth.synth(n, A, tmpl, seed=21)
    inds, wa, wd = _expected(x, K, iters, 5)
    for p in parts:
        np.testing.assert_array_equal(p["med"], inds)
    np.testing.assert_array_equal(np.concatenate([p["a"] for p in parts]), wa)
    np.testing.assert_array_equal(
        np.concatenate([p["d"] for p in parts]).astype(np.float64), wd)


@pytest.mark.parametrize("world,n,A,K,tmpl,iters,cands", [
    (8, 60000, 30, 260, 400, 1, -1),    # the ladder of 8 / 16 / 32 candidates per round
    (8, 60000, 30, 260, 400, 0, 32),    # every round with 32
    (8, 3000, 12, 40, 9, 0, 16),        # shards of one or two tiles, some empty
    (8, 3000, 12, 60, 9, 1, -1),        # ... and the ladder's decisions with empty shards
])
def test_eight_processes_on_one_gpu(tmp_path, world, n, A, K, tmpl, iters, cands):
    """BASELINE.json configs[3]'s process layout -- EIGHT ranks, one process
    each, every rank's mailbox mapped into every other by hipIpc, the exchange
    of a round written and polled on the device (ek_ms_run) -- as far as one
    GPU allows: all eight contexts share it.  k-hybrid (the PAM exchanges go
    through gloo) against the single-process oracle."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    out = str(tmp_path / "r")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, "-c", _CHILD2, ROOT, str(r), str(world),
                               port, out, str(n), str(A), str(K), str(tmpl),
                               str(iters), str(cands), "ipc"],
                              env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = [p.communicate(timeout=1200)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    parts = [np.load(out + ".%d.npz" % r) for r in range(world)]
    x = synth.synth(n, A, tmpl, seed=21)
    inds, wa, wd = _expected(x, K, iters, 5)
    for p in parts:
        np.testing.assert_array_equal(p["med"], inds)
    np.testing.assert_array_equal(np.concatenate([p["a"] for p in parts]), wa)
    np.testing.assert_array_equal(
        np.concatenate([p["d"] for p in parts]).astype(np.float64), wd)


_CHILD_MBOX = r"""
import sys, threading
sys.path.insert(0, sys.argv[1])
import numpy as np
from enspara_amd import sharded, synth
from enspara_amd.device import FrameStore
from oracle import cluster as oc
shards, n, A, K, cands = [int(v) for v in sys.argv[2:7]]
cutoff = float(sys.argv[7])
x = synth.synth(n, A, 11, seed=n + shards)
inds, wa, wd = oc.kcenters(x, n_clusters=K or None, dist_cutoff=cutoff or None)
stores = []
for r in range(shards):
    lo, cnt = sharded.shard_bounds(n, shards, r)
    st = FrameStore(cnt, A, device=0, global_offset=lo)
    st.load(x[lo:lo + cnt])
    st.set_option(4, cands)
    st.ms_setup(shards, r)
    # (what the run needs is allocated before any shard enters it: an allocation
    # inside ms_run waits for the whole device -- for the other shards' kernels
    # too, which by then wait for this shard's message)
    st.reserve_centers(K if K else n)
    stores.append(st)
boxes = [st.ms_mailbox() for st in stores]
for st in stores:
    for p in range(shards):
        st.ms_connect(p, boxes[p][0], boxes[p][1])
out = [None] * shards
def work(r):
    out[r] = stores[r].ms_run(0, K if K else n, cutoff)
for rep in range(2):
    for st in stores:
        # (round 6: the per-prefix maxima of a round of 16 taken by the pass, then by
        # the chain kernel's sweep; a shard may choose for itself -- the messages are
        # the same --, so the odd shards keep the default)
        if st is stores[0] or len(stores) % 2 == 0:
            st.set_option("pass_sweep", 2 if rep == 0 else 0)
        # (the headers first and the offers of the state the chain really left, then the
        # speculating offers with their exchanges without a pass: every shard alike)
        st.set_option("ms_two_phase", 1 if rep == 0 else 0)
        st.reset_state()
        st.sync()
    th = [threading.Thread(target=work, args=(r,)) for r in range(shards)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for o in out:
        np.testing.assert_array_equal(o[0], np.array(inds))
    parts = [st.download_state() for st in stores]
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), wa)
    np.testing.assert_array_equal(
        np.concatenate([p[0] for p in parts]).astype(np.float64), wd)
    # distances.max() after the last update (kcenters.py:226)
    assert out[0][2] == np.float32(wd.max()), (out[0][2], wd.max())
if K > 4:
    # the same fit in two calls: the second continues from the first's labels
    k1 = K // 3
    for st in stores:
        st.reset_state()
        st.sync()
    def part(r, first, count):
        out[r] = stores[r].ms_run(first, count, cutoff)
    got = []
    for first, count in ((0, k1), (k1, K - k1)):
        th = [threading.Thread(target=part, args=(r, first, count))
              for r in range(shards)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        got += [int(i) for i in out[0][0]]
    assert got == [int(i) for i in inds], (got, inds)
    parts = [st.download_state() for st in stores]
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), wa)
print("ok", len(inds), stores[0].ms_state())
"""


@pytest.mark.parametrize("shards,n,A,K,cutoff,cands", [
    (2, 9000, 21, 70, 0.0, 16), (3, 7000, 10, 0, 0.3, 8),
    (3, 600, 5, 40, 0.0, 16),           # the third shard is empty
    (2, 30000, 33, 300, 0.0, 16), (8, 40000, 20, 400, 0.0, 16),
    (2, 9000, 21, 70, 0.0, 32), (3, 600, 5, 40, 0.0, 32),
    (2, 30000, 33, 300, 0.0, 32), (8, 40000, 20, 400, 0.0, 32),
    (4, 50000, 24, 20000, 0.0, -1),     # configs[3]'s center count, the ladder
    (1, 777, 3, 200, 0.0, 8), (2, 777, 3, 200, 0.0, 16)])   # one sweeping workgroup per shard
def test_mailbox_rounds_between_contexts(shards, n, A, K, cutoff, cands):
    """the rounds of csrc/ek_mshard.hip with the exchange on the device: the
    shards are contexts of ONE process on the one GPU, their mailboxes plain
    addresses, every shard's loop runs in its own host thread (ek_ms_run) --
    twice from the untouched state: the second start is simultaneous.  (In a
    child process with one hardware queue per stream: shards that wait for one
    another on the device must not share a queue, which one shard per GPU never
    does.)"""
    env = dict(os.environ)
    env["GPU_MAX_HW_QUEUES"] = "16"
    p = subprocess.run([sys.executable, "-c", _CHILD_MBOX, ROOT, str(shards),
                        str(n), str(A), str(K), str(cands), str(cutoff)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.strip().splitlines()[-1].startswith("ok")


@pytest.mark.parametrize("cases,seed,env_extra", [
    (120, 0, {}),
    # (the bench shapes' atom counts, three and more shards: mailboxes of 50 - 800 KB; 7 of
    # these 60 runs went wrong while uncached memory was still handed back to hipFree)
    (60, 50, {"FUZZ_ATOMS": "300,500", "FUZZ_SHARDS_MIN": "3"})])
def test_mailbox_rounds_in_a_process_that_has_seen_many_contexts(cases, seed, env_extra):
    """tools/fuzz_ms.py: randomized multi-shard runs in ONE process (1 .. 8 shards, ragged
    and empty shards, 1 .. 100 atoms or 300 / 500, center counts and cut-offs, the ladder and pinned round
    widths, the exchange in one step and in two) against the oracle.  What round 6 found with
    it: a shard of up to 4096 frames in rounds of 8 reduced its waves' maxima before all of
    them were written (one workgroup: no arrival, so no barrier); and hipFree of the
    mailboxes' uncached memory left the process with allocations that overlapped live ones
    -- an upload landing in another context's frames, wrong centers, faults -- once a few
    hundred contexts had come and gone (the blocks are kept for reuse now, ek_uncached_alloc).
    Neither shows in a process that runs one configuration."""
    env = dict(os.environ)
    env["GPU_MAX_HW_QUEUES"] = "16"
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ms.py"), str(cases),
                        str(seed)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "%d cases, 0 mismatches" % cases in p.stdout, p.stdout[-2000:]


_CHILD_TI = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port,
                        rank=rank, world_size=world)
_agit = dist.all_gather_into_tensor
def agit(out_t, in_t, group=None):
    torch.cuda.current_stream().synchronize()
    o = torch.empty(out_t.shape, dtype=out_t.dtype)
    _agit(o, in_t.cpu(), group=group)
    out_t.copy_(o)
dist.all_gather_into_tensor = agit
from enspara_amd import sharded, synth
from enspara_amd.device import FrameStore
rng = np.random.RandomState(3)
A, T, per, K = 40, 24, 512, 60
tmpl = synth.templates(T, A, 9)
x = np.concatenate([tmpl[t] + rng.normal(scale=0.05, size=(per, A, 3))
                    for t in range(T)]).astype(np.float32)      # time-ordered blocks
n = len(x)
lo, cnt = sharded.shard_bounds(n, world, rank)
torch.cuda.set_device(0)
ts = torch.cuda.Stream(device=0)
with FrameStore(cnt, A, device=0, global_offset=lo, stream=ts.cuda_stream) as st:
    st.load(x[lo:lo + cnt])
    st.reset_state()
    sh = sharded.DeviceShard(st)
    with torch.cuda.stream(ts):
        idx, cd = sharded.kcenters_sharded(sh, 0, K, 0.0, use_triangle_inequality=True)
    d, a = st.download_state()
    tiles, skipped = st.ti_stats()
np.savez(out + ".%d.npz" % rank, idx=idx, d=d, a=a, tiles=tiles, skipped=skipped)
dist.barrier()
dist.destroy_process_group()
"""


def test_triangle_inequality_in_the_sharded_iteration(tmp_path):
    """reference kcenters.py:351-364 (`use_triangle_inequality` in
    _kcenters_iteration_mpi): two shards on the one GPU (collectives staged
    through gloo), frames in blocks of one template each: same centers, labels
    and distances as the oracle's plain run, and most tiles are never read once
    the templates have centers"""
    import socket
    from oracle import cluster as oc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    out = str(tmp_path / "ti")
    procs = [subprocess.Popen([sys.executable, "-c", _CHILD_TI, ROOT, str(r), "2",
                               port, out], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    parts = [np.load(out + ".%d.npz" % r) for r in range(2)]
    rng = np.random.RandomState(3)
    A, T, per, K = 40, 24, 512, 60
    tmpl = synth.templates(T, A, 9)
    x = np.concatenate([tmpl[t] + rng.normal(scale=0.05, size=(per, A, 3))
                        for t in range(T)]).astype(np.float32)
    inds, wa, wd = oc.kcenters(x, n_clusters=K)
    for p in parts:
        np.testing.assert_array_equal(p["idx"], np.array(inds))
        assert int(p["tiles"]) == (K - 1) * ((len(x) // 2 + 255) // 256)
        assert int(p["skipped"]) > 0.5 * int(p["tiles"])
    np.testing.assert_array_equal(np.concatenate([p["a"] for p in parts]), wa)
    np.testing.assert_array_equal(
        np.concatenate([p["d"] for p in parts]).astype(np.float64), wd)


# ---- the estimators' mpi_mode=True (every rank passes its own frames) ------------
_CHILD3 = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
rank, world, port, out, backend = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
torch.cuda.set_device(0)
if backend == "nccl":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", port)
    dist.init_process_group("nccl", rank=rank, world_size=world,
                            device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port,
                            rank=rank, world_size=world)
    _agit, _ar = dist.all_gather_into_tensor, dist.all_reduce
    def agit(out_t, in_t, group=None):
        torch.cuda.current_stream().synchronize()
        o = torch.empty(out_t.shape, dtype=out_t.dtype)
        _agit(o, in_t.cpu(), group=group)
        out_t.copy_(o)
    def ar(t, op=dist.ReduceOp.SUM, group=None):
        torch.cuda.current_stream().synchronize()
        h = t.cpu()
        _ar(h, op=op, group=group)
        t.copy_(h)
    dist.all_gather_into_tensor, dist.all_reduce = agit, ar
from enspara_amd import sharded, synth
from enspara_amd.cluster import KCenters, KHybrid
n, A, K = int(sys.argv[7]), 12, int(sys.argv[8])
x = synth.synth(n, A, 7, seed=13)
lo, cnt = sharded.shard_bounds(n, world, rank)
mine = x[lo:lo + cnt]
kc = KCenters("rmsd", n_clusters=K, mpi_mode=True).fit(mine)
hy = KHybrid("rmsd", n_clusters=K, kmedoids_updates=2, random_state=3,
             mpi_mode=True).fit(mine)
# a warm start (kcenters.py:200-213): the same initial centers on every rank
init = [x[5], x[n // 2], x[7], x[5]]
ws = KCenters("rmsd", n_clusters=K, mpi_mode=True).fit(mine, init_centers=init)
wh = KHybrid("rmsd", n_clusters=K, kmedoids_updates=1, random_state=5,
             mpi_mode=True).fit(mine, init_centers=init)
np.savez(out + ".%d.npz" % rank, lo=lo,
         kc_ci=np.array(kc.center_indices_), kc_a=kc.labels_, kc_d=kc.distances_,
         kc_c=np.array(kc.centers_),
         ws_ci=np.array(ws.center_indices_), ws_a=ws.labels_, ws_d=ws.distances_,
         ws_c=np.array(ws.centers_),
         wh_ci=np.array(wh.center_indices_), wh_a=wh.labels_, wh_d=wh.distances_,
         wh_c=np.array(wh.centers_),
         hy_ci=np.array(hy.center_indices_), hy_a=hy.labels_, hy_d=hy.distances_,
         hy_c=np.array(hy.centers_))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,backend,n,K", [(1, "nccl", 5000, 30),
                                               (2, "gloo", 5000, 30),
                                               (3, "gloo", 500, 12)])
def test_estimators_in_mpi_mode(tmp_path, world, backend, n, K):
    # (3 ranks over 500 frames = 2 tiles: the last rank owns no frames)
    import socket
    from enspara_amd import sharded
    from oracle import cluster as oc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    out = str(tmp_path / "r")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, "-c", _CHILD3, ROOT, str(r),
                               str(world), port, out, backend, str(n), str(K)],
                              env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(world)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    parts = [np.load(out + ".%d.npz" % r) for r in range(world)]
    x = synth.synth(n, 12, 7, seed=13)
    starts = [sharded.shard_bounds(n, world, r)[0] for r in range(world)]
    inds, a, d = oc.kcenters(x, n_clusters=K)
    rs = np.random.RandomState(3)
    wi, wd, wa = list(inds), d.copy(), a.copy()
    for _ in range(2):
        wi, wd, wa = oc.pam_update(x, wi, wa, wd, random_state=rs)
    # the warm starts: the single-process oracle from the same initial centers
    init = [x[5], x[n // 2], x[7], x[5]]
    si, sa, sd = oc.kcenters(x, n_clusters=K, init_centers=init)
    assert len(si) == K
    hi, hd, ha = oc.pam_update(x, list(si), sa.copy(), sd.copy(),
                               random_state=np.random.RandomState(5))
    for key, want_i, want_a, want_d in (("kc", inds, a, d), ("hy", wi, wa, wd),
                                        ("ws", si, sa, sd), ("wh", hi, ha, hd)):
        for p in parts:             # (rank, local index) pairs, kcenters.py:375-376
            got = [starts[int(r)] + int(i) for r, i in p[key + "_ci"]]
            assert got == [int(i) for i in want_i]
            want_c = x[[int(i) for i in want_i]]
            if key == "ws":
                # k-centers alone: the caller's initial centers, then the new ones
                # (kcenters.py:200-240); the second copy of x[5] attracted nothing,
                # so three labels are occupied and K - 3 centers are new
                want_c = np.concatenate([np.array(init), want_c[3:]])
            np.testing.assert_array_equal(p[key + "_c"], want_c)
        np.testing.assert_array_equal(
            np.concatenate([p[key + "_a"] for p in parts]), want_a)
        np.testing.assert_array_equal(
            np.concatenate([p[key + "_d"] for p in parts]), want_d)


# ---- KMedoids / kmedoids() in MPI mode: a warm start w.r.t. all data ---------------
_CHILD4 = _CHILD3[:_CHILD3.index("from enspara_amd import sharded, synth")] + r"""
from enspara_amd import synth
from enspara_amd.cluster import KMedoids
from enspara_amd.cluster.kmedoids import kmedoids
lengths = [int(v) for v in sys.argv[7].split(",")]
inp = np.load(sys.argv[8])
x = synth.synth(sum(lengths), 14, 11, seed=33)
starts = np.concatenate([[0], np.cumsum(lengths)])
held = np.concatenate([np.arange(starts[t], starts[t + 1])
                       for t in range(rank, len(lengths), world)]
                      or [np.zeros(0, dtype=np.int64)]).astype(np.int64)
lo = int(inp["los"][rank])
mine = x[held]
a, d = inp["a"][lo:lo + len(mine)], inp["d"][lo:lo + len(mine)]
# the estimator decides by the group's size (the reference: mpi.size() > 1)
km = KMedoids("rmsd", n_iters=2, mpi_mode=True if world == 1 else None)
np.random.seed(0)
r1 = kmedoids(mine, "rmsd", n_iters=2, assignments=a, distances=d,
              cluster_center_inds=[int(i) for i in inp["flat"]], X_lengths=lengths,
              random_state=np.random.RandomState(4), mpi_mode=True)
r2 = kmedoids(mine, "rmsd", n_iters=2, assignments=a, distances=d,
              cluster_center_inds=[tuple(int(v) for v in p) for p in inp["pairs"]],
              X_lengths=lengths, random_state=np.random.RandomState(4),
              mpi_mode=True if world == 1 else None)
np.savez(out + ".%d.npz" % rank, lo=lo,
         r1_ci=np.array(r1.center_indices), r1_a=r1.assignments, r1_d=r1.distances,
         r1_c=np.array(r1.centers),
         r2_ci=np.array(r2.center_indices), r2_a=r2.assignments, r2_d=r2.distances,
         r2_c=np.array(r2.centers))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo"), (3, "gloo")])
def test_kmedoids_in_mpi_mode(tmp_path, world, backend):
    """kmedoids() with the reference's MPI-mode warm start (kmedoids.py:133-146,
    :264-283, :365-407): cluster centers as flat indices and as (trajectory,
    frame) pairs w.r.t. all data, the ranks holding the trajectories striped;
    two sweeps on the device; equal to the single-process oracle on the frames in
    the ranks' order"""
    import socket
    from oracle import cluster as oc
    lengths = [300, 120, 260, 200, 180]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    x = synth.synth(sum(lengths), 14, 11, seed=33)
    starts = np.concatenate([[0], np.cumsum(lengths)])
    held = [np.concatenate([np.arange(starts[t], starts[t + 1])
                            for t in range(r, len(lengths), world)]
                           or [np.zeros(0, dtype=np.int64)]).astype(np.int64)
            for r in range(world)]
    perm = np.concatenate(held)
    los = np.concatenate([[0], np.cumsum([len(h) for h in held])])[:-1]
    xp = x[perm]
    K = 9
    inds, a, d = oc.kcenters(xp, n_clusters=K)
    flat = perm[[int(i) for i in inds]]
    traj = np.searchsorted(starts, flat, side="right") - 1
    pairs = np.stack([traj, flat - starts[traj]], axis=1)
    rs = np.random.RandomState(4)
    wi, wd, wa = [int(i) for i in inds], d.copy(), a.copy()
    for _ in range(2):
        wi, wd, wa = oc.pam_update(xp, wi, wa, wd, random_state=rs)
    inp = str(tmp_path / "in.npz")
    np.savez(inp, flat=flat, pairs=pairs, a=a, d=d, los=los)
    out = str(tmp_path / "r")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, "-c", _CHILD4, ROOT, str(r),
                               str(world), port, out, backend,
                               ",".join(str(v) for v in lengths), inp],
                              env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    parts = [np.load(out + ".%d.npz" % r) for r in range(world)]
    for key in ("r1", "r2"):
        for p in parts:
            got = [int(los[int(r)]) + int(i) for r, i in p[key + "_ci"]]
            assert got == [int(i) for i in wi]
            np.testing.assert_array_equal(p[key + "_c"], xp[[int(i) for i in wi]])
        np.testing.assert_array_equal(
            np.concatenate([p[key + "_a"] for p in parts]), wa)
        np.testing.assert_array_equal(
            np.concatenate([p[key + "_d"] for p in parts]), wd)


def test_bench_two_ranks_on_one_device():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one
    process per rank), with EK_BENCH_ONE_DEVICE=1: both ranks on device 0 and a
    gloo rendezvous -- the N > 1 code path of the bench (mailboxes over hipIpc,
    the validation fit, barriers, the max-over-ranks time, rank 0's one JSON
    line), exercised where there is one GPU.  The fit's centers are checked by
    the sharded tests above; here the line has to come out whole."""
    import json
    env = dict(os.environ)
    env.update({"EK_BENCH_ONE_DEVICE": "1", "GPU_MAX_HW_QUEUES": "16",
                "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "1", "--frames", "120000", "--atoms", "60", "--centers", "600",
           "--templates", "600", "--pam-sweeps", "0", "--no-msm", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 1
    assert d["config"]["world_size"] == 2 and d["value"] > 0
    assert "mailbox" in d["config"]["sharding"]
    assert d["roofline"]["frac"] > 0 and d["config"]["frames_total"] == 120000
    # what every rank spent where (round 5): a scaling curve that explains itself
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1]
    for r in pr:
        assert r["transport"] == "mailbox" and r["frames"] > 0
        assert r["exchanges"] > 0 and r["exchanges_without_a_pass"] >= 0
        assert r["wait_for_peers_us_per_exchange"] >= r["wait_for_own_flag_us_per_exchange"] >= 0
        sr = r["sampled_round_us"]
        assert sr["rounds_sampled"] > 0 and sr["pass"] > 0
        assert sr["chain_with_exchange"] > 0 and sr["plan"] > 0
        assert sum(v["centers"] for v in r["rounds_by_candidates"].values()) == 600
        assert len(r["can_access_peer"]) >= 1
