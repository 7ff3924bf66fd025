"""Scratch: rounds needed by 2 device shards on one GPU (gloo-staged
collectives) vs one shard, same data.  usage: two_shard_rounds.py <rank> <world> <port>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
_agit = dist.all_gather_into_tensor
def agit(out_t, in_t, group=None):
    torch.cuda.current_stream().synchronize()
    o = torch.empty(out_t.shape, dtype=out_t.dtype)
    _agit(o, in_t.cpu(), group=group)
    out_t.copy_(o)
dist.all_gather_into_tensor = agit
from enspara_amd import sharded, synth
from enspara_amd.device import FrameStore
n, A, K = 400000, 100, 1500
x = synth.synth(n, A, 2000, seed=1)
lo, cnt = sharded.shard_bounds(n, world, rank)
torch.cuda.set_device(0)
ts = torch.cuda.Stream(device=0)
st = FrameStore(cnt, A, device=0, global_offset=lo, stream=ts.cuda_stream)
st.load(x[lo:lo + cnt]); st.reset_state()
sh = sharded.DeviceShard(st)
with torch.cuda.stream(ts):
    idx, _ = sharded.kcenters_sharded(sh, 0, K, 0.0)
if rank == 0:
    print("world %d: centers %d rounds %d  centers/round %.2f" % (world, len(idx), st.spec_rounds(), len(idx) / st.spec_rounds()), flush=True)
dist.barrier(); dist.destroy_process_group()
