"""Frame-sharded k-centers over the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI);
rank r owns the contiguous frame block [offset_r, offset_r + n_r) of the
concatenated data set.  Per iteration each rank contributes ONE candidate
record -- (local max distance, global index, trace, centred coordinates of
that frame; ``ek_record_bytes`` long, 3.6 KB at 300 atoms) -- to a single
all-gather; every rank then picks the same winner on the device (largest
distance, lowest rank on ties) inside the distance kernel's prologue.

This replaces the reference's MPI iteration, enspara/cluster/kcenters.py:
314-378: two pickled allgathers (:332-335) + owner arg-max (:337) + Bcast of
the frame (mpi/ops.py:169-212) + Barrier/allreduce(MAX) for the stop test
(mpi/ops.py:128-140).  Contiguous sharding + "lowest rank wins" reproduces
the single-process np.argmax first-index rule globally, so results are
identical to the 1-GPU run bit for bit, and center indices are plain global
frame numbers instead of the reference's (rank, local index) pairs.

The driver is written against a small shard protocol so that the same loop
runs on CPU with the "gloo" backend in tests (tests/test_sharded_gloo.py
plugs in a checker-backed shard); the product shard is :class:`DeviceShard`.
"""
import numpy as np


class DeviceShard:
    """Shard protocol on top of a :class:`enspara_amd.device.FrameStore`.

    Records live in torch CUDA tensors so that torch.distributed can move
    them.  The FrameStore must have been created on a non-default torch
    stream (``s = torch.cuda.Stream(); FrameStore(..., stream=s.cuda_stream)``)
    and the driver must run under ``with torch.cuda.stream(s)``: torch orders
    its collectives against the current stream, so kernels and the all-gather
    are then ordered on the device without host syncs.  (Handle 0, the legacy
    default stream, means "create your own" to ek_ctx_create.)
    """

    def __init__(self, store):
        import torch
        self.torch = torch
        self.store = store
        self.device = torch.device("cuda", store.device)

    @property
    def record_bytes(self):
        return self.store.record_bytes

    def new_buffer(self, nbytes):
        return self.torch.empty(nbytes, dtype=self.torch.uint8,
                                device=self.device)

    def local_candidate(self, rec):
        self.store.local_candidate(rec.data_ptr())

    def step(self, all_recs, n_recs, label, cutoff, own_rec):
        self.store.kcenters_step(all_recs.data_ptr(), n_recs, label, cutoff,
                                 own_rec.data_ptr())

    def progress(self):
        """-> (n_done, stopped) ; synchronises"""
        _, _, n_done = self.store.history(0, 0)
        return n_done

    def history(self, first, count):
        idx, cd, n_done = self.store.history(first, count)
        return idx, cd, n_done

    def reset_history(self):
        self.store.reset_history()


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def kcenters_sharded(shard, first_label, max_new, dist_cutoff=0.0, group=None,
                     check_every=16, fresh=True):
    """Run up to ``max_new`` k-centers iterations over all ranks' shards.

    Every rank calls this with its own shard.  Returns
    (global center indices int64 [k], their pre-update distances float32 [k]);
    identical on every rank.  The per-frame state stays on each shard.
    ``fresh=False`` continues a previous call (keeps the accepted-center
    history and the candidate record already held by the shard).
    """
    import torch.distributed as dist
    world, _ = _world(group)
    collective = dist.is_available() and dist.is_initialized()
    rb = shard.record_bytes
    mine = shard.new_buffer(rb)
    everyone = shard.new_buffer(rb * world) if collective else mine
    if fresh:
        shard.reset_history()
    shard.local_candidate(mine)
    open_loop = not (dist_cutoff > 0)
    issued = 0
    while issued < max_new:
        todo = (max_new - issued) if open_loop else min(check_every,
                                                        max_new - issued)
        for i in range(todo):
            if collective:
                dist.all_gather_into_tensor(everyone, mine, group=group)
            shard.step(everyone, world, first_label + issued + i,
                       float(dist_cutoff), mine)
        issued += todo
        if not open_loop:
            n_done = shard.progress()
            if n_done < first_label + issued:     # a step hit the stop rule
                break
    idx, cd, n_done = shard.history(first_label, max_new)
    k = max(0, n_done - first_label)
    return np.array(idx[:k], dtype=np.int64), np.array(cd[:k],
                                                      dtype=np.float32)


def shard_bounds(n_total, world, rank, align=256):
    """Contiguous, tile-aligned split of n_total frames over ``world`` ranks.
    -> (offset, count)"""
    tiles = (n_total + align - 1) // align
    base, extra = divmod(tiles, world)
    t0 = rank * base + min(rank, extra)
    t1 = t0 + base + (1 if rank < extra else 0)
    lo = min(n_total, t0 * align)
    hi = min(n_total, t1 * align)
    return lo, hi - lo
