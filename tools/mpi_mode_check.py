"""Scratch: KHybrid(mpi_mode=True) with WORLD ranks on one GPU (gloo-staged
collectives) vs the oracle.  usage: mpi_mode_check.py <rank> <world> <port> <outprefix>; then with 'check'."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
n, A, K = [int(v) for v in os.environ.get("MM_SHAPE", "9000,10,40").split(",")]
if sys.argv[1] == "check":
    from enspara_amd import sharded, synth
    from oracle import cluster as oc
    world, out = int(sys.argv[2]), sys.argv[3]
    x = synth.synth(n, A, 9, seed=2)
    parts = [np.load(out + ".%d.npz" % r) for r in range(world)]
    starts = [sharded.shard_bounds(n, world, r)[0] for r in range(world)]
    inds, a, d = oc.kcenters(x, n_clusters=K)
    rs = np.random.RandomState(3)
    for _ in range(2):
        inds, d, a = oc.pam_update(x, inds, a, d, random_state=rs)
    ok = True
    for p in parts:
        ok &= [starts[int(r)] + int(i) for r, i in p["ci"]] == [int(i) for i in inds]
    ok &= np.array_equal(np.concatenate([p["a"] for p in parts]), a)
    ok &= np.array_equal(np.concatenate([p["d"] for p in parts]), d)
    print("world %d: %s" % (world, "OK" if ok else "MISMATCH"))
    sys.exit(0)
import torch, torch.distributed as dist
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
_agit, _ar = dist.all_gather_into_tensor, dist.all_reduce
def agit(o_t, i_t, group=None):
    torch.cuda.current_stream().synchronize(); o = torch.empty(o_t.shape, dtype=o_t.dtype); _agit(o, i_t.cpu(), group=group); o_t.copy_(o)
def ar(t, op=dist.ReduceOp.SUM, group=None):
    torch.cuda.current_stream().synchronize(); h = t.cpu(); _ar(h, op=op, group=group); t.copy_(h)
dist.all_gather_into_tensor, dist.all_reduce = agit, ar
from enspara_amd import sharded, synth
from enspara_amd.cluster import KHybrid
x = synth.synth(n, A, 9, seed=2)
lo, cnt = sharded.shard_bounds(n, world, rank)
hy = KHybrid("rmsd", n_clusters=K, kmedoids_updates=2, random_state=3, mpi_mode=True).fit(x[lo:lo + cnt])
np.savez(out + ".%d.npz" % rank, ci=np.array(hy.center_indices_), a=hy.labels_, d=hy.distances_)
dist.barrier(); dist.destroy_process_group()
