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
        """-> number of centers so far; synchronises"""
        _, _, n_done = self.store.history(0, 0)
        return n_done

    # multi-candidate rounds
    @property
    def candidates(self):
        return self.store.candidates

    def spec_begin(self, first_label, limit, recs):
        self.store.spec_begin(first_label, limit, recs.data_ptr())

    def spec_round(self, recs_all, n_recs, cutoff):
        self.store.spec_round(recs_all.data_ptr(), n_recs, cutoff)

    def spec_localmax(self, hdr):
        self.store.spec_localmax(hdr.data_ptr())

    def spec_apply(self, hdrs_all, n_hdrs, cutoff):
        self.store.spec_apply(hdrs_all.data_ptr(), n_hdrs, cutoff)

    def spec_round_end(self, recs):
        self.store.spec_round_end(recs.data_ptr())

    def spec_progress(self):
        return self.store.spec_progress()

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
    T = getattr(shard, "candidates", 1)
    if T > 1 and world * T <= 64:
        return _kcenters_sharded_rounds(shard, first_label, max_new,
                                        dist_cutoff, group, fresh, world, T,
                                        collective)
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


def _kcenters_sharded_rounds(shard, first_label, max_new, dist_cutoff, group,
                             fresh, world, T, collective):
    """Multi-candidate rounds (csrc/ek_spec.hip) across ranks.  Messages:
    per round one all-gather of T candidate records per rank; per accepted
    center one all-gather of a 16-byte (max distance, global index) header per
    rank -- the same information the reference exchanges per iteration
    (kcenters.py:332-335), while the frame Bcast (:345) is paid once per round
    instead of once per center."""
    import torch.distributed as dist
    rb = shard.record_bytes
    recs_mine = shard.new_buffer(rb * T)
    recs_all = shard.new_buffer(rb * T * world) if collective else recs_mine
    hdr_mine = shard.new_buffer(16)
    hdr_all = shard.new_buffer(16 * world) if collective else hdr_mine
    limit = first_label + max_new
    if fresh:
        shard.reset_history()
    shard.spec_begin(first_label, limit, recs_mine)
    n_done = first_label
    per_round = 0.6 * T
    while max_new > 0:
        rounds = max(2, min(256, int((limit - n_done) / per_round) + 1))
        before = n_done
        for _ in range(rounds):
            if collective:
                dist.all_gather_into_tensor(recs_all, recs_mine, group=group)
            shard.spec_round(recs_all, world * T, float(dist_cutoff))
            for _j in range(1, T):
                shard.spec_localmax(hdr_mine)
                if collective:
                    dist.all_gather_into_tensor(hdr_all, hdr_mine, group=group)
                shard.spec_apply(hdr_all, world, float(dist_cutoff))
            shard.spec_round_end(recs_mine)
        n_done, stopped = shard.spec_progress()
        if stopped or n_done >= limit:
            break
        per_round = max(1.0, (n_done - before) / rounds)
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
