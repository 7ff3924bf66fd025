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

    def set_triangle_inequality(self, on):
        self.store.set_option("triangle", 1 if on else 0)

    def quad_copy_ready(self):
        return self.store.quad_copy_ready()

    def pin_candidates(self, T):
        self.store.set_option("candidates", int(T))

    def reserve_centers(self, n_centers):
        self.store.reserve_centers(n_centers)

    def candidates_option(self):
        """what the option holds (-1: automatic), to put it back afterwards"""
        return self.store.get_option("candidates")

    def set_small_shards(self, on):
        self.store.set_option("small_shards", 1 if on else 0)

    def assign_nearest(self, centers_xyz):
        """every local frame against the given centers (float32 [K, A, 3]):
        the state becomes (nearest center, distance), util.py:199-203"""
        self.store.assign_nearest(centers_xyz)

    def state(self):
        """-> (distances float32 [n_local], labels int32 [n_local]) on the host"""
        return self.store.download_state()

    def set_state(self, distances, assignments):
        """the caller's per-frame state (a warm start, kmedoids.py:133-146)"""
        self.store.upload_state(distances, assignments)

    def set_exact(self, on):
        """every frame's distance IS the one to the medoid its label names
        (option key 7: lets the PAM search use the triangle inequality)"""
        self.store.set_option("state_exact", 1 if on else 0)

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

    # chained cheap steps (csrc/ek_chain.hip)
    def spec_chain_bytes(self):
        return self.store.spec_chain_bytes()

    def spec_chain_rows(self, rows):
        self.store.spec_chain_rows(rows.data_ptr())

    def spec_chain_max(self, rows_all, n_shards, hdrs):
        self.store.spec_chain_max(rows_all.data_ptr(), n_shards,
                                  hdrs.data_ptr())

    def spec_chain_apply(self, hdrs_all, n_shards, cutoff):
        self.store.spec_chain_apply(hdrs_all.data_ptr(), n_shards, cutoff)

    def spec_round_end(self, recs):
        self.store.spec_round_end(recs.data_ptr())

    def spec_progress(self):
        return self.store.spec_progress()

    # rounds with one exchange each (csrc/ek_mshard.hip)
    def ms_setup(self, world, rank):
        return self.store.ms_setup(world, rank)

    def ms_begin(self, first_label, limit):
        self.store.ms_begin(first_label, limit)

    def ms_local(self, cutoff, msg):
        self.store.ms_local(cutoff, msg.data_ptr())

    def ms_global(self, cutoff, msgs_all):
        self.store.ms_global(cutoff, msgs_all.data_ptr())

    def ms_end(self):
        self.store.ms_end()

    ms_connected = 0        # shards whose mailboxes are connected (connect_mailboxes)

    def ms_run(self, first_label, max_new, cutoff):
        idx, cd, _ = self.store.ms_run(first_label, max_new, cutoff)
        return idx, cd

    def history(self, first, count):
        idx, cd, n_done = self.store.history(first, count)
        return idx, cd, n_done

    def reset_history(self):
        self.store.reset_history()

    # PAM across shards: centers travel as centred coordinates + trace
    @property
    def n_atoms(self):
        return self.store.A

    @property
    def n_local(self):
        return self.store.n

    @property
    def offset(self):
        return self.store.global_offset

    def host_to_buffer(self, arr):
        """int64 numpy array -> tensor a collective can move"""
        return self.torch.from_numpy(np.ascontiguousarray(
            arr, dtype=np.int64)).to(self.device)

    def to_host(self, t):
        """Small device tensor -> numpy, waiting by polling an event: a
        blocking synchronize can cost far more than the proposal it guards
        when the host thread is put to sleep."""
        torch = self.torch
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        while not ev.query():
            pass
        return h.numpy()

    def new_table(self, rows):
        """-> (coords float32 [rows, 3A], meta int64 [2 * rows]), zeroed;
        meta[:rows] holds the traces' float64 bit patterns, meta[rows:] is
        free for the caller (global frame indices)."""
        t = self.torch
        return (t.zeros((rows, 3 * self.n_atoms), dtype=t.float32,
                        device=self.device),
                t.zeros(2 * rows, dtype=t.int64, device=self.device))

    def fill_rows(self, local_frames, rows, coords, meta):
        if len(local_frames):
            self.store.centered_frames(local_frames, rows, coords.data_ptr(),
                                       meta.data_ptr())

    def pam_begin_table(self, coords, meta, n_medoids):
        self.store.pam_begin_table(coords.data_ptr(), meta.data_ptr(),
                                   n_medoids)

    def pam_count_batch(self, cid0, count):
        return self.store.pam_count_members_batch(cid0, count)

    def pam_select_batch(self, cid0, js):
        return self.store.pam_select_members_batch(cid0, js)

    def pam_count(self, cid):
        return self.store.pam_count_members(cid)

    def pam_select(self, cid, j):
        return self.store.pam_select_member(cid, j)

    def pam_prefetch_centers(self, coords, meta, count, win_lo=0, win_count=0):
        self.store.pam_prefetch_centers(coords.data_ptr(), meta.data_ptr(),
                                        count, win_lo, win_count)

    def pam_propose_center(self, cid, slot, coords, meta, row, n_members_local,
                           win_lo, win_count, out):
        self.store.pam_propose_center(
            cid, slot, coords.data_ptr() + 12 * self.n_atoms * row,
            meta.data_ptr() + 8 * row, n_members_local, win_lo, win_count,
            out.data_ptr())

    def pam_commit(self, accept):
        self.store.pam_commit(accept)


def _to_host(shard, t):
    f = getattr(shard, "to_host", None)
    return f(t) if f is not None else t.cpu().numpy()


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def kcenters_sharded(shard, first_label, max_new, dist_cutoff=0.0, group=None,
                     check_every=16, fresh=True, use_triangle_inequality=False):
    """Run up to ``max_new`` k-centers iterations over all ranks' shards.

    Every rank calls this with its own shard.  Returns
    (global center indices int64 [k], their pre-update distances float32 [k]);
    identical on every rank.  The per-frame state stays on each shard.
    ``fresh=False`` continues a previous call (keeps the accepted-center
    history and the candidate record already held by the shard).
    ``use_triangle_inequality`` (reference kcenters.py:351-364): one center per
    pass, every shard keeps the accepted centers and does not read the tiles of
    256 frames none of which can come closer to the new center than to its own
    (csrc/ek_kcenters.hip ek_ti_center_tab_kernel); same results, fewer bytes
    on time-ordered data.
    """
    import torch.distributed as dist
    world, _ = _world(group)
    collective = dist.is_available() and dist.is_initialized()
    T = getattr(shard, "candidates", 1)
    if hasattr(shard, "set_triangle_inequality"):
        shard.set_triangle_inequality(bool(use_triangle_inequality))
    if use_triangle_inequality:
        T = 1               # (the test is per center: one-center passes)
    if T > 8 and hasattr(shard, "quad_copy_ready"):
        # (a group that has to narrow its rounds does so for THIS run only: the
        # caller's own setting of the option comes back afterwards)
        before = (shard.candidates_option()
                  if hasattr(shard, "candidates_option") else None)
        T0, T = T, _agree_on_form(shard, T, group, world, collective)
        if T != T0 and before is not None:
            try:
                return _kcenters_sharded_ms(shard, first_label, max_new,
                                            dist_cutoff, group, fresh, world,
                                            T, collective)
            finally:
                shard.pin_candidates(before)
    if T > 1 and world <= MAX_ROUND_RECORDS and hasattr(shard, "ms_local"):
        return _kcenters_sharded_ms(shard, first_label, max_new, dist_cutoff,
                                    group, fresh, world, T, collective)
    if T > 1 and world <= MAX_ROUND_RECORDS:
        return _kcenters_sharded_rounds(shard, first_label, max_new,
                                        dist_cutoff, group, fresh, world, T,
                                        collective)
    if T > 1:
        import logging
        logging.getLogger(__name__).warning(
            "%d ranks: more than the %d candidate records a round can choose "
            "from; falling back to one exchange per center", world,
            MAX_ROUND_RECORDS)
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


# records a round's plan can choose its candidates from (the device keeps a
# 64 x 64 table of their pairwise distances, csrc/ek_spec.hip)
MAX_ROUND_RECORDS = 64


def _agree_on_form(shard, T, group, world, collective):
    """Every shard of a group has to run rounds of the same width: they plan
    the same candidates from the same messages.  Rounds of 16 / 32 need a third
    copy of the frames on the device; a rank without room for it would, on its
    own, run rounds of 8 beside peers running 16 -- diverging plans, a mailbox
    time-out.  So the ranks ask first (the copy is made here if it can be) and
    if ANY has no room, ALL pin their rounds to 8 candidates (option key 4)."""
    try:
        ok = 1 if shard.quad_copy_ready() else 0
    except Exception:           # reported by the run itself; here: narrow rounds
        ok = 0
    everyone = _gather_i64(shard, [ok, shard.n_local], group, world, collective)
    # (the ladder of the rounds across shards: earlier to 16 candidates where the
    # shards are small -- the same answer on every rank, from the same numbers)
    if hasattr(shard, "set_small_shards"):
        shard.set_small_shards(int(everyone[:, 1].max()) < 300000)
    if int(everyone[:, 0].min()) == 1:
        return T
    import logging
    logging.getLogger(__name__).warning(
        "rank(s) %s have no memory for the quad copy of their frames: every "
        "rank runs rounds of 8 candidates",
        [int(r) for r in np.flatnonzero(everyone[:, 0] == 0)])
    shard.pin_candidates(8)
    return 8


def _kcenters_sharded_ms(shard, first_label, max_new, dist_cutoff, group, fresh,
                         world, T, collective):
    """Rounds of up to T candidates with ONE exchange each (csrc/
    ek_mshard.hip): per round every rank all-gathers one message -- its (max
    distance, global index) in the state every prefix of the round's chain
    would leave, and its farthest frames of the state the whole chain would
    leave -- where the reference moves two allgathers, a frame broadcast and
    an allreduce per CENTER (kcenters.py:332-348).  Every rank takes the same
    decisions from the same messages.  If the shards are connected by peer
    mailboxes (`shard.ms_connected`), the exchange happens on the device and
    the whole loop inside the library (`ms_run`): nothing here per round."""
    import torch.distributed as dist
    rank = dist.get_rank(group) if collective else 0
    if fresh:
        shard.reset_history()
    if getattr(shard, "ms_connected", 0) == world:
        # (nothing is allocated inside the run: a hipMalloc / hipFree waits for the
        # whole device, and the peers may be polling for this shard's message by then)
        if hasattr(shard, "reserve_centers"):
            shard.reserve_centers(first_label + max_new)
        if collective:      # (the shards wait for one another on the device from here on)
            dist.barrier(group=group)
        idx, cd = shard.ms_run(first_label, max_new, float(dist_cutoff))
        return (np.array(idx, dtype=np.int64), np.array(cd, dtype=np.float32))
    key = (world, rank)
    if getattr(shard, "_ms_key", None) != key:
        shard._ms_bytes = shard.ms_setup(world, rank)
        shard._ms_key = key
    mb = shard._ms_bytes
    mine = shard.new_buffer(mb)
    everyone = shard.new_buffer(mb * world) if collective else mine
    limit = first_label + max_new
    shard.ms_begin(first_label, limit)
    n_done = first_label
    per_round = 0.6 * T
    while max_new > 0:
        # (the same count on every rank: n_done is)
        rounds = max(2, min(256, int((limit - n_done) / per_round) + 2))
        before = n_done
        for _ in range(rounds):
            shard.ms_local(float(dist_cutoff), mine)
            if collective:
                dist.all_gather_into_tensor(everyone, mine, group=group)
            shard.ms_global(float(dist_cutoff), everyone)
        n_done, stopped = shard.spec_progress()
        if stopped or n_done >= limit:
            break
        per_round = max(1.0, (n_done - before) / rounds)
    shard.ms_end()
    idx, cd, n_done = shard.history(first_label, max_new)
    k = max(0, min(max_new, n_done - first_label))
    return np.array(idx[:k], dtype=np.int64), np.array(cd[:k],
                                                      dtype=np.float32)


def connect_mailboxes(shard, group=None):
    """Peer mailboxes for the shards of a torch.distributed group of processes
    on one node (one GPU each): every rank publishes the hipIpc handles of its
    mailbox, opens the others', and from then on `kcenters_sharded` runs its
    rounds with the exchange on the device (DeviceShard.ms_run).  Collective:
    every rank calls it, and every rank makes the SAME sequence of collectives
    whatever fails where (a rank that cannot export or open a handle -- no
    peer access to one GPU -- still takes part in both gathers).  Returns True
    with ``shard.ms_connected = world`` only if every rank connected every
    peer; otherwise every rank closes what it opened, ``ms_connected`` is 0,
    ``shard.ms_connect_error`` holds this rank's exception (if it had one) and
    the caller stays on the all-gather transport."""
    import torch.distributed as dist
    world, rank = _world(group)
    st = shard.store
    why, mine = None, None
    try:
        st.ms_setup(world, rank)
        mine = st.ms_mailbox(ipc=True) if world > 1 else None
    except Exception as e:          # reported below, after the collectives
        why = e
    handles = [None] * world
    if world > 1:
        dist.all_gather_object(handles, mine, group=group)
    if why is None:
        try:
            for p in range(world):
                if p == rank:
                    st.ms_connect(p)
                elif handles[p] is None:
                    raise RuntimeError("rank %d published no mailbox" % p)
                else:
                    st.ms_connect(p, ipc=handles[p])
        except Exception as e:
            why = e
    # one more gather: the verdict, and the point no rank passes before every
    # rank has mapped every mailbox
    verdicts = [why is None]
    if world > 1:
        verdicts = [None] * world
        dist.all_gather_object(verdicts, why is None, group=group)
    shard.ms_connect_error = why
    if not all(verdicts):
        try:                        # closes the mappings, ms_peers = 0
            st.ms_setup(world, rank)
        except Exception:
            pass
        shard.ms_connected = 0
        return False
    shard.ms_connected = world
    return True


def _kcenters_sharded_rounds(shard, first_label, max_new, dist_cutoff, group,
                             fresh, world, T, collective):
    """Multi-candidate rounds (csrc/ek_spec.hip) across ranks.  Messages per
    round: one all-gather of T candidate records per rank, then -- chained
    cheap steps, csrc/ek_chain.hip -- one all-gather of each rank's view of
    the candidate frames it owns (320 B) and one of its per-prefix (max
    distance, global index) headers (128 B): three exchanges per round of
    ~6 centers, against the reference's two allgathers + Bcast + allreduce per
    center (kcenters.py:332-348).  A shard without the chained entry points
    falls back to one 16-byte header all-gather per accepted center."""
    import torch.distributed as dist
    rb = shard.record_bytes
    # every rank offers its first `offer` records (they are ordered: record 0 is
    # its farthest point): all T while world * T fits the plan's table, fewer
    # per rank in larger groups -- the round still runs T candidates
    offer = min(T, max(1, MAX_ROUND_RECORDS // world))
    recs_mine = shard.new_buffer(rb * T)
    recs_all = shard.new_buffer(rb * offer * world) if collective else recs_mine
    hdr_mine = shard.new_buffer(16)
    hdr_all = shard.new_buffer(16 * world) if collective else hdr_mine
    limit = first_label + max_new
    chained = hasattr(shard, "spec_chain_rows")
    if chained:
        rows_b, hdrs_b = shard.spec_chain_bytes()
        rows_mine = shard.new_buffer(rows_b)
        rows_all = shard.new_buffer(rows_b * world) if collective else rows_mine
        chdr_mine = shard.new_buffer(hdrs_b)
        chdr_all = shard.new_buffer(hdrs_b * world) if collective else chdr_mine
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
                dist.all_gather_into_tensor(recs_all, recs_mine[:rb * offer],
                                            group=group)
            shard.spec_round(recs_all, world * offer if collective else T,
                             float(dist_cutoff))
            if chained:
                shard.spec_chain_rows(rows_mine)
                if collective:
                    dist.all_gather_into_tensor(rows_all, rows_mine,
                                                group=group)
                shard.spec_chain_max(rows_all, world, chdr_mine)
                if collective:
                    dist.all_gather_into_tensor(chdr_all, chdr_mine,
                                                group=group)
                shard.spec_chain_apply(chdr_all, world, float(dist_cutoff))
            else:
                for _j in range(1, T):
                    shard.spec_localmax(hdr_mine)
                    if collective:
                        dist.all_gather_into_tensor(hdr_all, hdr_mine,
                                                    group=group)
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


def warm_start_sharded(shard, init_centers, group=None):
    """The warm start of the reference's k-centers (kcenters.py:200-206) over
    all ranks' shards: every frame goes to its nearest initial center
    (util.assign_to_nearest_center: strict <, the lower index wins ties) and
    every OCCUPIED label is represented by its member closest to that center
    (util.find_cluster_centers: the first such member) -- first in the GLOBAL
    order, rank r's frames following rank r-1's, so that the result is the
    single-process one.  (The reference's MPI mode runs find_cluster_centers on
    each rank's local arrays, kcenters.py:205: indices that mean different
    frames on different ranks.  Not reproduced.)

    Every rank passes the same ``init_centers``.  Returns the list of global
    frame indices, one per occupied label in label order: the k-centers loop
    continues from ``first_label = len(result)`` (kcenters.py:306)."""
    import torch.distributed as dist
    world, rank = _world(group)
    collective = dist.is_available() and dist.is_initialized()
    from .cluster.util import _stack_centers
    centers = _stack_centers([c for c in init_centers])
    K0 = int(centers.shape[0])
    shard.assign_nearest(centers)
    d, a = shard.state()
    d = np.asarray(d, dtype=np.float64)
    a = np.asarray(a, dtype=np.int64)
    best_d = np.full(K0, np.inf, dtype=np.float64)
    best_g = np.full(K0, -1, dtype=np.int64)
    if len(a):
        # per label: smallest distance, then smallest index
        order = np.lexsort((np.arange(len(a)), d, a))
        lab = a[order]
        first = np.flatnonzero(np.r_[True, lab[1:] != lab[:-1]])
        best_d[lab[first]] = d[order[first]]
        best_g[lab[first]] = shard.offset + order[first]
    tables = [(best_d, best_g)]
    if collective and world > 1:
        tables = [None] * world
        dist.all_gather_object(tables, (best_d, best_g), group=group)
    ctr = []
    for lab in range(K0):
        win_d, win_g = np.inf, -1
        for bd, bg in tables:               # rank order = global frame order
            if bg[lab] >= 0 and (win_g < 0 or bd[lab] < win_d):
                win_d, win_g = bd[lab], int(bg[lab])
        if win_g >= 0:
            ctr.append(win_g)
    return ctr


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


# ---------------------------------------------------------------------------
# PAM (k-medoids) sweeps over sharded frames
# ---------------------------------------------------------------------------
PAM_OUT = np.dtype([("sum_old", "<f8"), ("sum_new", "<f8"),
                    ("n_frames", "<i8"), ("n_amb", "<u4"), ("moved", "<u4")])
assert PAM_OUT.itemsize == 32


def _gather_i64(shard, values, group, world, collective):
    """Every rank's small int64 vector -> ndarray [world, len(values)]."""
    import torch
    import torch.distributed as dist
    values = np.ascontiguousarray(values, dtype=np.int64)
    if not collective:
        return values[None, :].copy()
    mine = shard.host_to_buffer(values)
    everyone = torch.empty(world * len(values), dtype=torch.int64,
                           device=mine.device)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    return _to_host(shard, everyone).reshape(world, len(values)).copy()


def _share_rows(coords, meta, group, collective):
    """Each row was filled by exactly one rank and is zero elsewhere: an
    integer sum over ranks of the raw bit patterns reproduces it exactly on
    every rank (x + 0 + ... + 0 in two's complement), for the float32
    coordinates, the float64 traces and the int64 indices alike."""
    if not collective:
        return
    import torch
    import torch.distributed as dist
    dist.all_reduce(coords.view(torch.int32), op=dist.ReduceOp.SUM,
                    group=group)
    dist.all_reduce(meta, op=dist.ReduceOp.SUM, group=group)


class _ShardWindow:
    __slots__ = ("lo", "hi", "m", "m_local", "before", "j", "gidx", "stale")


def pam_sweep_sharded(shard, medoids, proposals=None, random_state=None,
                      group=None, prefetch=8):
    """One PAM sweep (reference kmedoids.py:575-699, MPI branch) over all
    ranks' shards.  ``medoids``: global frame index of every cluster's medoid,
    the same list on every rank; ``random_state`` must also be the same on
    every rank (an int seed or identically seeded RandomState).  Returns the
    updated list -- identical on every rank and identical to what the
    single-GPU sweep returns for the concatenated frames; the per-frame state
    stays on the shards.

    Messages: per window of ``prefetch`` clusters one all-gather of member
    counts and one exchange of the proposed frames' coordinates (3.6 KB each
    at 300 atoms); per proposal one all-gather of a 32-byte record per rank
    (cost sums, moved-cluster mask) -- against the reference's pickled
    allgather + Bcast of the frame + two allreduces per proposal
    (kmedoids.py:482-517, mpi/ops.py:143-212)."""
    import torch.distributed as dist
    from .cluster.kcenters import check_random_state
    world, rank = _world(group)
    collective = dist.is_available() and dist.is_initialized()
    if collective and world > 1 and random_state is None and proposals is None:
        raise ValueError("pam_sweep_sharded needs a random_state shared by "
                         "all ranks")
    rs = check_random_state(random_state)
    medoids = [int(g) for g in medoids]
    K = len(medoids)
    if proposals is not None and len(proposals) != K:
        raise ValueError("Length of 'proposals' didn't match length of "
                         "'medoids' (%d != %d)." % (len(proposals), K))
    width = max(1, min(int(prefetch), 8))
    lay = _gather_i64(shard, [shard.offset, shard.n_local], group, world,
                      collective)
    n_total = int(lay[:, 1].sum())
    lo_mine, hi_mine = shard.offset, shard.offset + shard.n_local

    def owned(g):
        return lo_mine <= g < hi_mine

    # medoid table, assembled from the owners' rows
    coords, meta = shard.new_table(K)
    rows = [r for r, g in enumerate(medoids) if owned(g)]
    shard.fill_rows([medoids[r] - lo_mine for r in rows], rows, coords, meta)
    _share_rows(coords, meta, group, collective)
    shard.pam_begin_table(coords, meta, K)

    pc, pm = shard.new_table(width)            # window proposals
    oc, om = shard.new_table(1)                # a proposal outside the window
    out_mine = shard.new_buffer(PAM_OUT.itemsize)
    out_all = (shard.new_buffer(PAM_OUT.itemsize * world) if collective
               else out_mine)
    win = None
    acceptances = 0
    for cid in range(K):
        if win is None or cid >= win.hi:
            win = _ShardWindow()
            win.lo, win.hi, win.stale = cid, min(K, cid + width), 0
            cnt = win.hi - win.lo
            mine = shard.pam_count_batch(win.lo, cnt)
            allc = _gather_i64(shard, mine, group, world, collective)
            win.m = [int(x) for x in allc.sum(axis=0)]
            win.m_local = [int(x) for x in mine]
            win.before = [int(x) for x in allc[:rank].sum(axis=0)]
            pc.zero_()
            pm.zero_()
            if proposals is None:
                ahead = np.random.RandomState()
                ahead.set_state(rs.get_state())
                win.j = []
                for m in win.m:
                    if m <= 0:
                        break
                    win.j.append(int(ahead.choice(m)))
                jl = [j - b for j, b in zip(win.j, win.before)]
                jl = [x if 0 <= x < ml else -1
                      for x, ml in zip(jl, win.m_local)]
                slots = [s for s, x in enumerate(jl) if x >= 0]
                if slots:
                    fl = shard.pam_select_batch(win.lo, jl)
                    shard.fill_rows([int(fl[s]) for s in slots], slots, pc, pm)
                    idx = np.zeros(width, dtype=np.int64)
                    for s in slots:
                        idx[s] = lo_mine + int(fl[s])
                    pm[width:] = shard.host_to_buffer(idx)
                n_guess = len(win.j)
            else:
                win.j = None
                want = [int(g) for g in proposals[win.lo:win.hi]]
                slots = [s for s, g in enumerate(want) if owned(g)]
                shard.fill_rows([want[s] - lo_mine for s in slots], slots, pc,
                                pm)
                idx = np.zeros(width, dtype=np.int64)
                for s in slots:
                    idx[s] = want[s]
                pm[width:] = shard.host_to_buffer(idx)
                n_guess = cnt
            _share_rows(pc, pm, group, collective)
            win.gidx = [int(g) for g in
                        _to_host(shard, pm[width:width + n_guess])]
            shard.pam_prefetch_centers(pc, pm, n_guess, win.lo,
                                       win.hi - win.lo)
        slot = cid - win.lo
        exact = not ((win.stale >> slot) & 1)
        if exact:
            m, m_local, before = win.m[slot], win.m_local[slot], win.before[slot]
            counted = False
        else:
            m_local = int(shard.pam_count(cid))
            allc = _gather_i64(shard, [m_local], group, world, collective)
            m, before = int(allc.sum()), int(allc[:rank].sum())
            counted = True
        use_slot = -1
        if proposals is None:
            j = int(rs.choice(m))                            # kmedoids.py:514
            if exact and slot < len(win.j) and j == win.j[slot]:
                use_slot, g = slot, win.gidx[slot]
        else:
            use_slot, g = slot, win.gidx[slot]
        if use_slot < 0:
            oc.zero_()
            om.zero_()
            jl = j - before
            if 0 <= jl < m_local:
                if not counted:
                    shard.pam_count(cid)         # leaves the scan for select
                f = int(shard.pam_select(cid, jl))
                shard.fill_rows([f], [0], oc, om)
                om[1:] = shard.host_to_buffer([lo_mine + f])
            _share_rows(oc, om, group, collective)
            g = int(_to_host(shard, om[1:])[0])
            shard.pam_propose_center(cid, -1, oc, om, 0, m_local, win.lo,
                                     win.hi - win.lo, out_mine)
        else:
            shard.pam_propose_center(cid, use_slot, pc, pm, use_slot, m_local,
                                     win.lo, win.hi - win.lo, out_mine)
        if collective:
            dist.all_gather_into_tensor(out_all, out_mine, group=group)
        recs = _to_host(shard, out_all).view(PAM_OUT)
        if int(recs[rank]["n_amb"]) > m_local:
            raise RuntimeError("PAM proposal for cluster %d: %d ambiguous "
                               "members on this shard, %d declared"
                               % (cid, int(recs[rank]["n_amb"]), m_local))
        s_old = s_new = 0.0
        moved = 0
        for r in range(len(recs)):               # fixed order: deterministic
            s_old += float(recs[r]["sum_old"])
            s_new += float(recs[r]["sum_new"])
            moved |= int(recs[r]["moved"])
        accept = (s_new / n_total) < (s_old / n_total)       # kmedoids.py:683
        shard.pam_commit(accept)
        if accept:
            medoids[cid] = g
            acceptances += 1
            win.stale |= moved
    return medoids


def khybrid_sharded(shard, n_clusters, dist_cutoff=0.0, n_iters=5,
                    random_state=None, group=None):
    """k-centers then ``n_iters`` PAM sweeps over all ranks' shards (reference
    hybrid.py:112-162 in MPI mode).  ``random_state`` is wrapped once, so an
    int seed gives one stream across the sweeps -- what the KHybrid estimator
    does (hybrid.py:78,103).  Returns the medoids' global frame indices;
    labels and distances stay on the shards."""
    from .cluster.kcenters import check_random_state
    import torch.distributed as dist
    rs = check_random_state(random_state)
    if n_clusters is None or np.isinf(n_clusters):
        # cut-off only: same cap as the single-GPU path (cluster/kcenters.py)
        world, _ = _world(group)
        lay = _gather_i64(shard, [shard.n_local], group, world,
                          dist.is_available() and dist.is_initialized())
        n_clusters = 2 * int(lay.sum()) + 16
    idx, _ = kcenters_sharded(shard, 0, int(n_clusters), dist_cutoff,
                              group=group)
    medoids = [int(i) for i in idx]
    for _ in range(int(n_iters)):
        medoids = pam_sweep_sharded(shard, medoids, random_state=rs,
                                    group=group)
    return medoids


# ---------------------------------------------------------------------------
# k-medoids across shards with a warm start (reference kmedoids.py, MPI mode)
# ---------------------------------------------------------------------------
def ctr_ids_mpi(cluster_center_inds, lengths, world):
    """The reference's ``ctr_ids_mpi`` (kmedoids.py:365-407): cluster centers
    given with respect to ALL data -- flat frame indices into the
    concatenation of all trajectories, or (global trajectory, frame) pairs --
    -> (rank, index into that rank's concatenated frames) pairs, for the
    reference's distribution of trajectories over ranks: trajectory t lives on
    rank ``t % world`` (mpi/io.py:126, load_trajectory_as_striped), a rank's
    frames are its trajectories t = rank, rank + world, .. one after the
    other.  A pure function of ``world`` (the reference reads mpi.size())."""
    lengths = [int(v) for v in lengths]
    starts = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    if len(cluster_center_inds) == 0:
        return []
    pairs = cluster_center_inds
    if not hasattr(cluster_center_inds[0], "__len__"):
        # [global frame index, ...] -> [[global trajectory, frame], ...]
        pairs = []
        for c in cluster_center_inds:
            c = int(c)
            if c < 0 or c >= starts[-1]:
                raise IndexError("cluster center index %d outside the %d frames "
                                 "X_lengths describes" % (c, int(starts[-1])))
            t = int(np.searchsorted(starts, c, side="right") - 1)
            pairs.append((t, c - int(starts[t])))
    out = []
    for t, f in pairs:
        t, f = int(t), int(f)
        if t < 0 or t >= len(lengths) or f < 0 or f >= lengths[t]:
            raise IndexError("cluster center (%d, %d) outside X_lengths" % (t, f))
        r = t % world
        owned = lengths[r::world]
        out.append((r, int(sum(owned[:t // world])) + f))
    return out


def _gather_rows_f32(shard, K, width, rows, values, group, collective):
    """A float32 table [K, width] of which this rank knows ``rows`` (their
    ``values``): every row is filled by exactly one rank, the integer sum of
    the bit patterns over ranks reproduces it everywhere (cf. _share_rows)."""
    import torch
    import torch.distributed as dist
    like = shard.new_buffer(4)
    tab = torch.zeros((max(K, 1), width), dtype=torch.float32, device=like.device)
    if len(rows):
        src = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float32)
                               .reshape(len(rows), width))
        tab[torch.as_tensor(list(rows), device=like.device)] = src.to(like.device)
    if collective:
        dist.all_reduce(tab.view(torch.int32), op=dist.ReduceOp.SUM, group=group)
    return np.array(_to_host(shard, tab))[:K]


def kmedoids_sharded(shard, xyz_local, n_iters=5, assignments=None,
                     distances=None, cluster_center_inds=None, X_lengths=None,
                     proposals=None, random_state=None, n_clusters=None,
                     group=None):
    """k-medoids over all ranks' shards, the reference's MPI mode
    (kmedoids.py:133-146 dispatch, :225-283 ``_kmedoids_inputs_tree_mpi``,
    :365-407 ``ctr_ids_mpi``, sweeps :410-476 / :575-699).

    Warm start (what the reference supports there): every rank passes the
    ``assignments`` / ``distances`` of ITS frames and the same
    ``cluster_center_inds`` with respect to all data -- flat indices or
    (trajectory, frame) pairs -- with ``X_lengths`` = the lengths of ALL
    trajectories; the ranks hold the trajectories striped (t % world).
    Supplying only some of the three raises the reference's
    ImproperlyConfigured.  Cold start (none of the three; the reference's own
    branch, :247-262, cannot run -- ``np.arange(X)``, ``.append`` on None):
    ``n_clusters`` distinct frames drawn over ALL frames from
    ``np.random.default_rng(random_state)`` like the single-process path
    (kmedoids.py:345-352), the same on every rank (rank 0's seed when
    ``random_state`` is None), then every frame to its nearest.

    ``proposals``: (rank, local index) pairs, one per cluster (:581-623).
    Returns (medoids as (rank, local index) pairs, centers float32 [K, A, 3]);
    labels and distances stay on the shard.  Collective: every rank calls it."""
    import torch.distributed as dist
    from .cluster.kcenters import check_random_state
    from .exception import DataInvalid, ImproperlyConfigured
    world, rank = _world(group)
    collective = dist.is_available() and dist.is_initialized()
    lay = _gather_i64(shard, [shard.offset, shard.n_local], group, world, collective)
    offs = [int(v) for v in lay[:, 0]]
    counts = [int(v) for v in lay[:, 1]]
    n_total = int(sum(counts))
    A = shard.n_atoms
    xyz_local = np.asarray(xyz_local, dtype=np.float32).reshape(-1, A, 3)
    given = [assignments is not None, distances is not None,
             cluster_center_inds is not None]
    if all(given):
        if X_lengths is None:
            raise ImproperlyConfigured(
                "If cluster_center_inds is given with respect to all data then "
                "X_lengths (the lengths of ALL trajectories) also needs to be "
                "supplied")
        if sum(int(v) for v in X_lengths[rank::world]) != counts[rank]:
            raise DataInvalid(
                "rank %d holds %d frames, but the trajectories X_lengths gives it "
                "(t %% %d == %d) have %d" % (rank, counts[rank], world, rank,
                                            sum(int(v) for v in X_lengths[rank::world])))
        pairs = ctr_ids_mpi(cluster_center_inds, X_lengths, world)
        med = [offs[r] + i for r, i in pairs]
        shard.set_state(np.asarray(distances), np.asarray(assignments))
        mine = [i for r, i in pairs if r == rank]
        # the medoids must sit at (numerically) zero distance (kmedoids.py:185-187)
        if not np.all(np.asarray(distances)[mine] < 0.001):
            raise DataInvalid(
                "cluster_center_inds name frames that are not at distance 0 "
                "from their cluster's center (max %g)"
                % float(np.max(np.asarray(distances)[mine])))
    elif not any(given):
        if n_clusters is None:
            raise ImproperlyConfigured(
                "Must provide n_clusters or cluster_center_inds, assignments,"
                "and distances for KMedoids in MPI mode.")
        seed = random_state
        if seed is None or not isinstance(seed, (int, np.integer)):
            draw = np.random.SeedSequence().entropy % (2 ** 62) if seed is None \
                else int(check_random_state(seed).randint(0, 2 ** 31 - 1))
            seed = int(_gather_i64(shard, [draw], group, world, collective)[0, 0])
        rng = np.random.default_rng(seed=int(seed))
        med = np.array([])
        while len(np.unique(med)) < int(n_clusters):
            med = rng.integers(0, n_total, int(n_clusters))
        # (global index in the ranks' order: rank r's frames after rank r-1's)
        bounds = np.cumsum([0] + counts)
        med = [offs[int(np.searchsorted(bounds, g, side="right") - 1)] +
               int(g - bounds[int(np.searchsorted(bounds, g, side="right") - 1)])
               for g in med]
        rows = [k for k, g in enumerate(med)
                if offs[rank] <= g < offs[rank] + counts[rank]]
        ctr = _gather_rows_f32(shard, len(med), 3 * A, rows,
                               xyz_local[[med[k] - offs[rank] for k in rows]],
                               group, collective)
        shard.assign_nearest(ctr.reshape(len(med), A, 3))
        if hasattr(shard, "set_exact"):
            shard.set_exact(True)
    else:
        raise ImproperlyConfigured(
            "For KMedoids, MPI mode can start from scratch without "
            "assignments, distances, or cluster_center_inds. "
            "Or, it requires that all are supplied.")
    prop = None
    if proposals is not None:
        if len(proposals) != len(med):
            raise DataInvalid(
                "Length of 'proposals' didn't match length of 'medoid_inds' "
                "({} != {}).".format(len(proposals), len(med)))
        if not hasattr(proposals[0], "__len__"):
            raise DataInvalid(
                "Depth of 'proposals' didn't match 'medoid_inds' "
                "(proposals[0] == {}, whereas medoid_inds[0] == {})".format(
                    proposals[0], (0, 0)))
        prop = [offs[int(r)] + int(i) for r, i in proposals]
    if random_state is None and prop is None:
        # every rank must draw the same member ranks (the reference agrees on
        # each draw across ranks, mpi/ops.py randind): rank 0's seed for all
        draw = int(np.random.SeedSequence().entropy % (2 ** 31 - 1))
        random_state = int(_gather_i64(shard, [draw], group, world,
                                       collective)[0, 0])
    rs = check_random_state(random_state)
    for _ in range(int(n_iters)):
        med = pam_sweep_sharded(shard, med, proposals=prop, random_state=rs,
                                group=group)
    pairs = []
    for g in med:
        r = max(q for q in range(world) if offs[q] <= g and counts[q] > 0
                and g < offs[q] + counts[q])
        pairs.append((r, int(g - offs[r])))
    rows = [k for k, (r, _) in enumerate(pairs) if r == rank]
    ctr = _gather_rows_f32(shard, len(med), 3 * A, rows,
                           xyz_local[[pairs[k][1] for k in rows]], group, collective)
    return pairs, ctr.reshape(len(med), A, 3)


# ---------------------------------------------------------------------------
# estimator-level entry: the reference's mpi_mode
# ---------------------------------------------------------------------------
def kmedoids_fit_sharded(traj, n_clusters=None, n_iters=5, assignments=None,
                         distances=None, cluster_center_inds=None,
                         X_lengths=None, proposals=None, random_state=None,
                         group=None):
    """``kmedoids()`` / ``KMedoids.fit`` when the process group has more than one
    rank (the reference decides by ``mpi.size() > 1``, kmedoids.py:172): every
    rank passes ITS OWN frames (the trajectories t % world == rank, one after
    the other), see ``kmedoids_sharded``.  One process per GPU; the current CUDA
    device holds the shard.  Returns a ClusterResult like the reference's MPI
    mode: ``center_indices`` as (rank, local frame index) pairs, this rank's
    ``assignments`` / ``distances``, the medoids' coordinates on every rank."""
    import torch
    import torch.distributed as dist
    from .cluster import util
    from .device import FrameStore, as_xyz
    from .exception import ImproperlyConfigured
    if not (dist.is_available() and dist.is_initialized()):
        raise ImproperlyConfigured(
            "mpi_mode needs an initialised torch.distributed process group "
            "(one process per GPU, e.g. under torchrun)")
    xyz = as_xyz(traj)
    n_local, A = int(xyz.shape[0]), int(xyz.shape[1])
    world, rank = _world(group)
    device = torch.cuda.current_device()
    tstream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(tstream):
        mine = torch.tensor([n_local], dtype=torch.int64, device="cuda")
        everyone = torch.empty(world, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(everyone, mine, group=group)
        counts = [int(c) for c in everyone.cpu().numpy()]
    starts = np.concatenate([[0], np.cumsum(counts)])
    with FrameStore(n_local, A, device=device, global_offset=int(starts[rank]),
                    stream=tstream.cuda_stream) as store:
        store.load(xyz)
        store.reset_state()
        shard = DeviceShard(store)
        with torch.cuda.stream(tstream):
            pairs, ctr = kmedoids_sharded(
                shard, xyz, n_iters=n_iters, assignments=assignments,
                distances=distances, cluster_center_inds=cluster_center_inds,
                X_lengths=X_lengths, proposals=proposals,
                random_state=random_state, n_clusters=n_clusters, group=group)
        d, a = store.download_state()
    return util.ClusterResult(
        center_indices=pairs, assignments=a.astype(np.int64),
        distances=d.astype(np.float64),
        centers=[ctr[k].copy() for k in range(len(pairs))])


def fit_sharded(traj, n_clusters=None, dist_cutoff=0.0, n_iters=0,
                random_state=None, group=None, use_triangle_inequality=False,
                init_centers=None):
    """k-centers (+ ``n_iters`` PAM sweeps) where every rank of the initialised
    ``torch.distributed`` group passes ITS OWN frames -- what ``mpi_mode=True``
    means in the reference (kcenters.py:314-378, kmedoids.py MPI branch): rank
    r's frames follow rank r-1's in the global order, ties go to the lowest
    rank.  One process per GPU; the current CUDA device holds the shard.

    Returns a ClusterResult like the reference's MPI mode: ``center_indices``
    as (rank, local frame index) pairs (kcenters.py:375-376), ``assignments`` /
    ``distances`` for this rank's frames, ``centers`` = the centers'
    coordinates (float32 [A, 3] each), the same list on every rank.
    ``random_state`` and ``init_centers`` (a warm start, kcenters.py:200-206:
    see ``warm_start_sharded``) must be the same on every rank."""
    import torch
    import torch.distributed as dist
    from .cluster import util
    from .cluster.kcenters import check_random_state
    from .device import FrameStore, as_xyz
    from .exception import ImproperlyConfigured
    if not (dist.is_available() and dist.is_initialized()):
        raise ImproperlyConfigured(
            "mpi_mode needs an initialised torch.distributed process group "
            "(one process per GPU, e.g. under torchrun)")
    xyz = as_xyz(traj)
    n_local, A = int(xyz.shape[0]), int(xyz.shape[1])
    world, rank = _world(group)
    device = torch.cuda.current_device()
    tstream = torch.cuda.Stream(device=device)
    rs = check_random_state(random_state)
    with torch.cuda.stream(tstream):
        # where this rank's block starts in the global order
        mine = torch.tensor([n_local], dtype=torch.int64, device="cuda")
        everyone = torch.empty(world, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(everyone, mine, group=group)
        counts = [int(c) for c in everyone.cpu().numpy()]
    starts = np.concatenate([[0], np.cumsum(counts)])
    n_total = int(starts[-1])
    if n_total == 0:
        raise ValueError("cannot cluster an empty trajectory")
    if n_clusters is None or np.isinf(n_clusters):
        if not dist_cutoff:
            raise ImproperlyConfigured("Either n_clusters or cluster_radius "
                                       "is required for KHybrid clustering")
        max_new = 2 * n_total + 16
    else:
        max_new = int(n_clusters)
    with FrameStore(n_local, A, device=device, global_offset=int(starts[rank]),
                    stream=tstream.cuda_stream) as store:
        store.load(xyz)
        store.reset_state()
        shard = DeviceShard(store)
        with torch.cuda.stream(tstream):
            med = []
            if init_centers is not None:
                med = warm_start_sharded(shard, init_centers, group=group)
                # (the shards' tables of accepted centers would lack these: the
                # tile skip is an optimisation only, results do not change)
                use_triangle_inequality = False
            idx, _ = kcenters_sharded(
                shard, len(med), max(0, max_new - len(med)),
                float(dist_cutoff or 0.0), group=group,
                use_triangle_inequality=use_triangle_inequality)
            med = med + [int(g) for g in idx]
            for _ in range(int(n_iters)):
                med = pam_sweep_sharded(shard, med, random_state=rs,
                                        group=group)
            # the centers' own coordinates, from their owners
            K = len(med)
            tab = torch.zeros((max(K, 1), 3 * A), dtype=torch.float32,
                              device="cuda")
            lo, hi = int(starts[rank]), int(starts[rank + 1])
            rows = [r for r, g in enumerate(med) if lo <= g < hi]
            if rows:
                loc = torch.from_numpy(np.ascontiguousarray(
                    xyz[[med[r] - lo for r in rows]].reshape(len(rows), -1)))
                tab[torch.tensor(rows, device="cuda")] = loc.to("cuda")
            dist.all_reduce(tab.view(torch.int32), op=dist.ReduceOp.SUM,
                            group=group)
            centers_xyz = tab.cpu().numpy()
        d, a = store.download_state()
    pairs = []
    for g in med:
        r = int(np.searchsorted(starts, g, side="right") - 1)
        pairs.append((r, int(g - starts[r])))
    centers = [centers_xyz[k].reshape(A, 3).copy() for k in range(len(med))]
    if init_centers is not None and int(n_iters) == 0:
        # k-centers alone returns the caller's initial centers followed by the
        # new ones (kcenters.py:200-240: `centers` starts as init_centers, and
        # the labels and distances of a warm start are measured to THEM), as the
        # single-process path does; `center_indices` stays the occupied labels'
        # closest members + the new centers' frames (kcenters.py:205).  (An
        # initial center that attracts no frame leaves `center_indices` shorter
        # than `centers` and the new labels start at len(center_indices),
        # kcenters.py:306: the reference's own misalignment, reproduced like the
        # single-process path does)
        n_new = len(idx)
        centers = ([np.asarray(as_xyz(c_)[0] if np.ndim(c_) == 3 else c_,
                               dtype=np.float32).reshape(A, 3).copy()
                    for c_ in init_centers] +
                   (centers[len(med) - n_new:] if n_new else []))
    return util.ClusterResult(
        center_indices=pairs, assignments=a.astype(np.int64),
        distances=d.astype(np.float64), centers=centers)
