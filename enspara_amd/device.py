"""Device-resident frames: the Python face of one ``ek_ctx``.

A :class:`FrameStore` holds one shard of a trajectory set in HBM (centred,
frame-minor tiles) together with the k-centers state
(distance and assignment per frame).  All arithmetic happens in the HIP
library; this class only moves pointers and sizes across the C ABI
(include/enspara_hip.h).
"""
import ctypes as C

import numpy as np

from . import _lib
from .exception import DataInvalid


def as_xyz(X):
    """Coordinates of ``X`` as a float32 C-contiguous [n, A, 3] array.

    Accepts what the reference's clustering accepts for metric 'rmsd': an
    object with ``.xyz`` (md.Trajectory; enspara/cluster/kcenters.py:283 indexes
    it, mdtraj reads ``.xyz``), a RaggedArray of per-trajectory coordinate
    blocks (its concatenated ``_data``), or a plain array."""
    if hasattr(X, "xyz"):
        X = X.xyz
    elif hasattr(X, "_data") and hasattr(X, "lengths"):
        X = X._data
    X = np.asarray(X)
    if X.ndim == 2 and X.shape[1] == 3:
        X = X[None]
    if X.ndim != 3 or X.shape[2] != 3:
        raise DataInvalid(
            "RMSD clustering needs coordinates shaped (n_frames, n_atoms, 3); "
            "got %s" % (X.shape,))
    return np.ascontiguousarray(X, dtype=np.float32)


def _option_key(key):
    if isinstance(key, str):
        try:
            return _lib.OPTIONS[key.lower()]
        except KeyError:
            raise ValueError("unknown option %r (one of %s)"
                             % (key, sorted(_lib.OPTIONS)))
    return int(key)


class FrameStore:
    """One shard of frames on one GPU."""

    def __init__(self, n_frames, n_atoms, device=0, global_offset=0,
                 stream=None):
        self.lib = _lib.load()
        self.n = int(n_frames)
        self.A = int(n_atoms)
        self.device = int(device)
        self.global_offset = int(global_offset)
        h = C.c_void_p()
        _lib.check(self.lib.ek_ctx_create(
            self.device, self.n, self.A, self.global_offset,
            C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.ek_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @classmethod
    def from_array(cls, X, device=0, global_offset=0, stream=None):
        xyz = as_xyz(X)
        st = cls(xyz.shape[0], xyz.shape[1], device, global_offset, stream)
        st.load(xyz)
        return st

    # -- data ---------------------------------------------------------------
    def load(self, xyz, first=0):
        xyz = as_xyz(xyz)
        if xyz.shape[1] != self.A:
            raise DataInvalid("atom count %d != %d" % (xyz.shape[1], self.A))
        _lib.check(self.lib.ek_load_frames(
            self._h, xyz.ctypes.data_as(C.c_void_p), int(first), xyz.shape[0], 0))

    def load_device(self, dev_ptr, count, first=0):
        """xyz already in device memory (float32 [count, A, 3])."""
        _lib.check(self.lib.ek_load_frames(
            self._h, C.c_void_p(int(dev_ptr)), int(first), int(count), 1))

    def sync(self):
        _lib.check(self.lib.ek_ctx_sync(self._h))

    @property
    def stream(self):
        return self.lib.ek_ctx_stream(self._h)

    # -- metric -------------------------------------------------------------
    def rmsd_to_frame(self, index):
        out = np.empty(self.n, dtype=np.float32)
        _lib.check(self.lib.ek_rmsd_to_center(self._h, int(index), None,
                                              _lib.f32p(out)))
        return out

    def rmsd_to_xyz(self, center_xyz):
        c = as_xyz(center_xyz)
        if c.shape[0] != 1 or c.shape[1] != self.A:
            raise DataInvalid("center must be one frame of %d atoms" % self.A)
        out = np.empty(self.n, dtype=np.float32)
        _lib.check(self.lib.ek_rmsd_to_center(self._h, -1, _lib.f32p(c),
                                              _lib.f32p(out)))
        return out

    # -- k-centers state ------------------------------------------------------
    def reset_state(self):
        _lib.check(self.lib.ek_state_reset(self._h))

    def download_state(self):
        """-> (distances float32 [n], assignments int32 [n])"""
        d = np.empty(self.n, dtype=np.float32)
        a = np.empty(self.n, dtype=np.int32)
        _lib.check(self.lib.ek_state_download(self._h, _lib.f32p(d),
                                              _lib.i32p(a)))
        return d, a

    def upload_state(self, dist, assign):
        d = np.ascontiguousarray(dist, dtype=np.float32)
        a = np.ascontiguousarray(assign, dtype=np.int32)
        if len(d) != self.n or len(a) != self.n:
            raise DataInvalid("state arrays must have %d entries" % self.n)
        _lib.check(self.lib.ek_state_upload(self._h, _lib.f32p(d),
                                            _lib.i32p(a)))

    def kcenters_run(self, first_label, max_new, dist_cutoff):
        """The device-resident k-centers loop on this shard alone.
        -> (center indices int64 [k], their pre-update distances float32 [k],
            final max distance)"""
        max_new = int(max_new)
        idx = np.empty(max(max_new, 1), dtype=np.int64)
        cd = np.empty(max(max_new, 1), dtype=np.float32)
        n_added = C.c_int32()
        fmax = C.c_float()
        _lib.check(self.lib.ek_kcenters_run(
            self._h, int(first_label), max_new, float(dist_cutoff),
            C.byref(n_added), _lib.i64p(idx), _lib.f32p(cd), C.byref(fmax)))
        k = n_added.value
        return idx[:k].copy(), cd[:k].copy(), fmax.value

    def assign_nearest(self, centers_xyz):
        c = as_xyz(centers_xyz)
        if c.shape[0] and c.shape[1] != self.A:
            raise DataInvalid("centers have %d atoms, frames %d"
                              % (c.shape[1], self.A))
        _lib.check(self.lib.ek_assign_nearest(self._h, _lib.f32p(c),
                                              c.shape[0]))

    # -- PAM ------------------------------------------------------------------
    def pam_begin(self, medoid_frames):
        m = np.ascontiguousarray(medoid_frames, dtype=np.int64)
        _lib.check(self.lib.ek_pam_begin(self._h, _lib.i64p(m), len(m)))

    def pam_count_members(self, cid):
        cnt = C.c_int64()
        _lib.check(self.lib.ek_pam_count_members(self._h, int(cid),
                                                 C.byref(cnt)))
        return cnt.value

    def pam_select_member(self, cid, j):
        f = C.c_int64()
        _lib.check(self.lib.ek_pam_select_member(self._h, int(cid), int(j),
                                                 C.byref(f)))
        return f.value

    def pam_propose(self, cid, frame_index):
        """-> (old cost, new cost, number of ambiguous frames)"""
        oc, nc = C.c_double(), C.c_double()
        na = C.c_int64()
        _lib.check(self.lib.ek_pam_propose(
            self._h, int(cid), int(frame_index), C.byref(oc), C.byref(nc),
            C.byref(na)))
        return oc.value, nc.value, na.value

    def pam_propose_member(self, cid, j):
        """Propose the j-th member of cluster cid (after pam_count_members).
        -> (frame index, old cost, new cost, number of ambiguous frames)"""
        oc, nc = C.c_double(), C.c_double()
        na, fi = C.c_int64(), C.c_int64()
        _lib.check(self.lib.ek_pam_propose_member(
            self._h, int(cid), int(j), C.byref(fi), C.byref(oc), C.byref(nc),
            C.byref(na)))
        return fi.value, oc.value, nc.value, na.value

    def pam_commit(self, accept):
        _lib.check(self.lib.ek_pam_commit(self._h, 1 if accept else 0))

    def pam_count_members_batch(self, cid0, count):
        """Member counts of clusters cid0..cid0+count-1 (count <= 8)."""
        out = np.zeros(count, dtype=np.int64)
        _lib.check(self.lib.ek_pam_count_members_batch(
            self._h, int(cid0), int(count), _lib.i64p(out)))
        return out

    def pam_select_members_batch(self, cid0, js):
        """frames[i] = js[i]-th member of cluster cid0+i (right after
        pam_count_members_batch(cid0, >= len(js)))."""
        js = np.ascontiguousarray(js, dtype=np.int64)
        out = np.zeros(len(js), dtype=np.int64)
        _lib.check(self.lib.ek_pam_select_members_batch(
            self._h, int(cid0), len(js), _lib.i64p(js), _lib.i64p(out)))
        return out

    def pam_prefetch(self, frames, win_lo=0, win_count=0):
        """Compute and keep the distances of every frame to frames[0..8).
        With the window of clusters the caller works through next (their
        proposals are `frames`), only frames a proposal can touch get exact
        distances while the state allows the triangle-inequality test."""
        f = np.ascontiguousarray(frames, dtype=np.int64)
        _lib.check(self.lib.ek_pam_prefetch_window(
            self._h, _lib.i64p(f), len(f), int(win_lo), int(win_count)))

    def pam_prefetch_passes(self):
        """-> (prefetches restricted to the touched frames, over all frames)"""
        a, b = C.c_int64(), C.c_int64()
        _lib.check(self.lib.ek_pam_prefetch_passes(self._h, C.byref(a),
                                                   C.byref(b)))
        return a.value, b.value

    def pam_sparse_stats(self):
        """-> (windows worked through in one workgroup, of them ended early)"""
        a, b = C.c_int64(), C.c_int64()
        _lib.check(self.lib.ek_pam_sparse_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pam_ahead_stats(self):
        """-> slots of those windows whose evaluation ahead of their turn was
        taken over (ek_set_option key 19)"""
        a = C.c_int64()
        _lib.check(self.lib.ek_pam_ahead_stats(self._h, C.byref(a)))
        return a.value

    def pam_propose_ex(self, cid, frame_index, n_members, win_lo=0,
                       win_count=0):
        """-> (old cost, new cost, number of ambiguous frames, moved mask)"""
        oc, nc = C.c_double(), C.c_double()
        na = C.c_int64()
        mask = C.c_uint32()
        _lib.check(self.lib.ek_pam_propose_ex(
            self._h, int(cid), int(frame_index), int(n_members), int(win_lo),
            int(win_count), C.byref(oc), C.byref(nc), C.byref(na),
            C.byref(mask)))
        return oc.value, nc.value, na.value, mask.value

    def pam_window_max(self):
        """The most proposals a window (pam_prefetch + pam_window_run) holds."""
        return int(self.lib.ek_pam_window_max())

    def pam_window_run(self, cid0, frames, n_members, win_count):
        """The proposals frames[i] for clusters cid0 + i (all prefetched),
        decided and committed on the device in order; stops at the first cluster
        whose membership an accepted proposal changed.
        -> (n_done, accept[int], old costs, new costs, ambiguous counts)"""
        f = np.ascontiguousarray(frames, dtype=np.int64)
        m = np.ascontiguousarray(n_members, dtype=np.int64)
        k = len(f)
        nd = C.c_int32()
        acc = np.zeros(k, dtype=np.int32)
        oc = np.zeros(k, dtype=np.float64)
        nc = np.zeros(k, dtype=np.float64)
        na = np.zeros(k, dtype=np.int64)
        _lib.check(self.lib.ek_pam_window_run(
            self._h, int(cid0), k, _lib.i64p(f), _lib.i64p(m), int(cid0),
            int(win_count), C.byref(nd), _lib.i32p(acc), _lib.f64p(oc),
            _lib.f64p(nc), _lib.i64p(na)))
        return nd.value, acc, oc, nc, na

    def pam_sweep_run(self, width, raw, pos, proposals, cid, medoids, accept,
                      old_cost, new_cost, n_amb):
        """The window loop of a sweep inside the library (ek_pam_sweep): from
        cluster `cid` on, draws taken from the raw 32-bit outputs `raw` (uint32,
        `pos` of them consumed so far).  medoids / accept / old_cost / new_cost /
        n_amb (arrays of len(medoids)) are updated in place.
        -> (status, cid, pos): status 0 done, 1 more raw outputs needed, 2 cluster
        `cid` is empty"""
        K = len(medoids)
        p = C.c_int64(int(pos))
        c = C.c_int32(int(cid))
        st = C.c_int32(0)
        props = None
        if proposals is not None:
            props = np.ascontiguousarray(proposals, dtype=np.int64)
        _lib.check(self.lib.ek_pam_sweep(
            self._h, K, int(width),
            raw.ctypes.data_as(C.POINTER(C.c_uint32)), len(raw), C.byref(p),
            _lib.i64p(props) if props is not None else None, C.byref(c),
            _lib.i64p(medoids), _lib.i32p(accept), _lib.f64p(old_cost),
            _lib.f64p(new_cost), _lib.i64p(n_amb), C.byref(st)))
        return st.value, c.value, p.value

    # -- PAM across shards (device pointers are plain ints) ----------------------
    def centered_frames(self, local_frames, rows, aos_ptr, G_ptr):
        """rows[i] of the device arrays at aos_ptr ([.., 3A] float32) / G_ptr
        (float64) := centred coordinates / trace of local frame
        local_frames[i]."""
        f = np.ascontiguousarray(local_frames, dtype=np.int64)
        r = np.ascontiguousarray(rows, dtype=np.int32)
        if len(f) != len(r):
            raise DataInvalid("centered_frames: %d frames, %d rows"
                              % (len(f), len(r)))
        _lib.check(self.lib.ek_centered_frames(
            self._h, _lib.i64p(f), _lib.i32p(r), len(f), int(aos_ptr),
            int(G_ptr)))

    def pam_begin_table(self, aos_ptr, G_ptr, n_medoids):
        _lib.check(self.lib.ek_pam_begin_table(self._h, int(aos_ptr),
                                               int(G_ptr), int(n_medoids)))

    def pam_prefetch_centers(self, aos_ptr, G_ptr, count, win_lo=0,
                             win_count=0):
        _lib.check(self.lib.ek_pam_prefetch_centers_window(
            self._h, int(aos_ptr), int(G_ptr), int(count), int(win_lo),
            int(win_count)))

    def pam_propose_center(self, cid, slot, aos_ptr, G_ptr, n_members_local,
                           win_lo, win_count, out_ptr):
        _lib.check(self.lib.ek_pam_propose_center(
            self._h, int(cid), int(slot), int(aos_ptr), int(G_ptr),
            int(n_members_local), int(win_lo), int(win_count), int(out_ptr)))

    def pam_prefetch_stats(self):
        """-> (proposals served from a prefetched vector, not served)"""
        h, m = C.c_int64(), C.c_int64()
        _lib.check(self.lib.ek_pam_prefetch_stats(self._h, C.byref(h),
                                                  C.byref(m)))
        return h.value, m.value

    # -- multi-shard primitives ----------------------------------------------
    @property
    def record_bytes(self):
        return int(self.lib.ek_record_bytes(self.A))

    def local_candidate(self, rec_dev_ptr=None):
        _lib.check(self.lib.ek_local_candidate(
            self._h, C.c_void_p(int(rec_dev_ptr)) if rec_dev_ptr else None))

    def kcenters_step(self, recs_dev_ptr, n_recs, label, dist_cutoff,
                      own_rec_out=None):
        _lib.check(self.lib.ek_kcenters_step(
            self._h, C.c_void_p(int(recs_dev_ptr)) if recs_dev_ptr else None,
            int(n_recs), int(label), float(dist_cutoff),
            C.c_void_p(int(own_rec_out)) if own_rec_out else None))

    # -- multi-candidate rounds across shards ----------------------------------
    @property
    def candidates(self):
        return int(self.lib.ek_spec_candidates(self._h))

    def ms_diag(self):
        """what the last ms_run spent where -> dict (exchanges, re-offers, microseconds
        waited for the peers' messages / for the own flag, mean microseconds of a
        sampled round's pass, chain kernel with its exchange, plan kernels)"""
        cnt = np.zeros(5, dtype=np.int64)
        ms = np.zeros(3, dtype=np.float64)
        _lib.check(self.lib.ek_ms_diag(self._h, _lib.i64p(cnt), _lib.f64p(ms)))
        return {"exchanges": int(cnt[0]), "reoffers": int(cnt[1]),
                "wait_peers_us": cnt[2] * 0.01, "wait_own_flag_us": cnt[3] * 0.01,
                "rounds_sampled": int(cnt[4]), "pass_us": ms[0] * 1e3,
                "chain_with_exchange_us": ms[1] * 1e3, "plan_us": ms[2] * 1e3}

    def quad_copy_ready(self):
        """True if the third copy of the frames that rounds of 16 / 32 candidates
        stream exists or could be made now (False: no memory for it)"""
        rc = int(self.lib.ek_quad_copy_ready(self._h))
        if rc < 0:
            _lib.check(rc)
        return rc == 1

    @property
    def round_candidates(self):
        """the widest round kcenters_run / ms_run may use: 16 unless the option
        "candidates" pins another form (32 only on request)"""
        return int(self.lib.ek_round_candidates(self._h))

    def spec_begin(self, first_label, limit, recs_out_ptr):
        _lib.check(self.lib.ek_spec_begin(self._h, int(first_label), int(limit),
                                          C.c_void_p(int(recs_out_ptr))))

    def spec_round(self, recs_all_ptr, n_recs, dist_cutoff):
        _lib.check(self.lib.ek_spec_round(
            self._h, C.c_void_p(int(recs_all_ptr)), int(n_recs),
            float(dist_cutoff)))

    def spec_localmax(self, hdr_out_ptr):
        _lib.check(self.lib.ek_spec_localmax(self._h,
                                             C.c_void_p(int(hdr_out_ptr))))

    def spec_apply(self, hdrs_all_ptr, n_hdrs, dist_cutoff):
        _lib.check(self.lib.ek_spec_apply(
            self._h, C.c_void_p(int(hdrs_all_ptr)), int(n_hdrs),
            float(dist_cutoff)))

    def spec_chain_bytes(self):
        """-> (bytes of a shard's candidate rows, of its per-prefix headers)"""
        a, b = C.c_int32(), C.c_int32()
        _lib.check(self.lib.ek_spec_chain_bytes(C.byref(a), C.byref(b)))
        return a.value, b.value

    def spec_chain_rows(self, rows_out_ptr):
        _lib.check(self.lib.ek_spec_chain_rows(
            self._h, C.c_void_p(int(rows_out_ptr))))

    def spec_chain_max(self, rows_all_ptr, n_shards, hdrs_out_ptr):
        _lib.check(self.lib.ek_spec_chain_max(
            self._h, C.c_void_p(int(rows_all_ptr)), int(n_shards),
            C.c_void_p(int(hdrs_out_ptr))))

    def spec_chain_apply(self, hdrs_all_ptr, n_shards, dist_cutoff):
        _lib.check(self.lib.ek_spec_chain_apply(
            self._h, C.c_void_p(int(hdrs_all_ptr)), int(n_shards),
            float(dist_cutoff)))

    def spec_round_end(self, recs_out_ptr):
        _lib.check(self.lib.ek_spec_round_end(self._h,
                                              C.c_void_p(int(recs_out_ptr))))

    def spec_progress(self):
        nd, st = C.c_int32(), C.c_int32()
        _lib.check(self.lib.ek_spec_progress(self._h, C.byref(nd),
                                             C.byref(st)))
        return nd.value, bool(st.value)

    def spec_rounds(self):
        r = C.c_int32()
        _lib.check(self.lib.ek_spec_rounds(self._h, C.byref(r)))
        return r.value

    # -- MSM counts over the resident labels (csrc/ek_msm.hip) ------------------
    def msm_counts(self, lengths, lag_time, n_states, sliding_window=True):
        """Transition counts of the labels this store holds (after a fit or
        assign_nearest), split into trajectories of `lengths` frames: COO
        (rows int32, cols int32, counts int64) sorted by (row, col).  The labels
        do not leave the device (reference flow: cluster -> assigns_to_counts,
        enspara/msm/transition_matrices.py:113-170)."""
        lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        cap = max(1, min(int(self.n), int(n_states) * int(n_states)))
        rows = np.empty(cap, dtype=np.int32)
        cols = np.empty(cap, dtype=np.int32)
        vals = np.empty(cap, dtype=np.int64)
        nnz = C.c_int64()
        _lib.check(self.lib.ek_msm_counts_ctx(
            self._h, _lib.i64p(lengths), len(lengths), int(lag_time),
            1 if sliding_window else 0, int(n_states), cap, _lib.i32p(rows),
            _lib.i32p(cols), _lib.i64p(vals), C.byref(nnz)))
        k = nnz.value
        return rows[:k], cols[:k], vals[:k]

    # -- rounds across shards, one exchange per round (csrc/ek_mshard.hip) ------
    def ms_setup(self, world, rank):
        """-> bytes of a shard's round message"""
        b = C.c_size_t()
        _lib.check(self.lib.ek_ms_setup(self._h, int(world), int(rank),
                                        C.byref(b)))
        return b.value

    def ms_mailbox(self, ipc=False):
        """This shard's mailbox: (address, flags address), or with ipc=True the
        two 64-byte hipIpc handles another process opens (bytes, bytes)."""
        if ipc:
            hm = C.create_string_buffer(64)
            hf = C.create_string_buffer(64)
            _lib.check(self.lib.ek_ms_mailbox(self._h, None, None, hm, hf))
            return hm.raw, hf.raw
        m, f = C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.ek_ms_mailbox(self._h, C.byref(m), C.byref(f), None,
                                          None))
        return m.value, f.value

    def ms_connect(self, peer, mbox=None, flags=None, ipc=None):
        """Where peer's mailbox is: addresses (contexts of one process) or
        ipc=(handle, handle) from that process's ms_mailbox(ipc=True)."""
        if ipc is not None:
            hm = C.create_string_buffer(bytes(ipc[0]), 64)
            hf = C.create_string_buffer(bytes(ipc[1]), 64)
            _lib.check(self.lib.ek_ms_connect(self._h, int(peer), None, None,
                                              hm, hf))
        else:
            _lib.check(self.lib.ek_ms_connect(
                self._h, int(peer), C.c_void_p(mbox), C.c_void_p(flags), None,
                None))

    def ms_begin(self, first_label, limit):
        _lib.check(self.lib.ek_ms_begin(self._h, int(first_label), int(limit)))

    def ms_local(self, dist_cutoff, message_out_ptr):
        _lib.check(self.lib.ek_ms_local(self._h, float(dist_cutoff),
                                        C.c_void_p(int(message_out_ptr))))

    def ms_global(self, dist_cutoff, messages_all_ptr):
        _lib.check(self.lib.ek_ms_global(self._h, float(dist_cutoff),
                                         C.c_void_p(int(messages_all_ptr))))

    def ms_end(self):
        _lib.check(self.lib.ek_ms_end(self._h))

    def ms_state(self):
        """-> (mode, exchanges completed since ms_setup, error code)"""
        m, e, r = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.ek_ms_state(self._h, C.byref(m), C.byref(e),
                                        C.byref(r)))
        return m.value, e.value, r.value

    def reserve_centers(self, n_centers):
        """allocate now what a run of the rounds to ``n_centers`` centers needs
        (nothing is allocated -- no device-wide wait -- inside ms_run then)"""
        _lib.check(self.lib.ek_reserve_centers(self._h, int(n_centers)))

    def debug_guards(self):
        """(EK_POISON=1 in the environment) how many of this context's buffers were
        written past their end; their names go to stderr"""
        rc = int(self.lib.ek_debug_guards(self._h))
        if rc < 0:
            _lib.check(rc)
        return rc

    def ms_run(self, first_label, max_new, dist_cutoff):
        """k-centers over all connected shards, exchange on the device; every
        shard calls it at the same time.  -> as kcenters_run"""
        max_new = int(max_new)
        idx = np.empty(max(max_new, 1), dtype=np.int64)
        cd = np.empty(max(max_new, 1), dtype=np.float32)
        n_added = C.c_int32()
        fmax = C.c_float()
        _lib.check(self.lib.ek_ms_run(
            self._h, int(first_label), max_new, float(dist_cutoff),
            C.byref(n_added), _lib.i64p(idx), _lib.f32p(cd), C.byref(fmax)))
        k = n_added.value
        return idx[:k].copy(), cd[:k].copy(), fmax.value

    def run_stats(self):
        """How the last kcenters_run / ms_run spent its rounds:
        -> {candidates per round: (rounds, centers accepted)} (a round of 32
        streams the frames twice, every other form once)"""
        p = np.zeros(5, dtype=np.int64)
        k = np.zeros(5, dtype=np.int64)
        _lib.check(self.lib.ek_run_stats(self._h, _lib.i64p(p), _lib.i64p(k)))
        return {T: (int(p[i]), int(k[i])) for i, T in enumerate((1, 4, 8, 16, 32))}

    def ti_stats(self):
        """Triangle inequality (set_option(11, 1)): (center, tile) pairs the last
        kcenters_run looked at, and how many it did not have to read."""
        t, k = C.c_int64(), C.c_int64()
        _lib.check(self.lib.ek_ti_stats(self._h, C.byref(t), C.byref(k)))
        return t.value, k.value

    def history(self, first, count):
        idx = np.empty(max(count, 1), dtype=np.int64)
        cd = np.empty(max(count, 1), dtype=np.float32)
        nd = C.c_int32()
        _lib.check(self.lib.ek_history_download(
            self._h, int(first), int(count), _lib.i64p(idx), _lib.f32p(cd),
            C.byref(nd)))
        return idx[:count], cd[:count], nd.value

    def reset_history(self):
        _lib.check(self.lib.ek_history_reset(self._h))

    # -- tuning ---------------------------------------------------------------
    def set_frames_per_lane(self, fpl):
        _lib.check(self.lib.ek_set_frames_per_lane(self._h, int(fpl)))

    def set_option(self, key, value):
        """``key``: a member of include/enspara_hip.h's ``enum ek_option`` -- its
        number, or its name without the prefix (``"candidates"``,
        ``"triangle"`` ...: ``_lib.OPTIONS``)."""
        _lib.check(self.lib.ek_set_option(self._h, _option_key(key), int(value)))

    def get_option(self, key):
        v = C.c_int32()
        _lib.check(self.lib.ek_get_option(self._h, _option_key(key), C.byref(v)))
        return v.value

    def last_run_timing(self):
        ms = C.c_float()
        k = C.c_int32()
        _lib.check(self.lib.ek_last_run_timing(self._h, C.byref(ms),
                                               C.byref(k)))
        return ms.value, k.value

    def timing_begin(self, sample_every=32, max_samples=512):
        _lib.check(self.lib.ek_timing_begin(self._h, int(sample_every),
                                            int(max_samples)))

    def timing_end(self):
        """-> (mean ms per sampled distance-kernel launch, samples)"""
        ms = C.c_float()
        k = C.c_int32()
        _lib.check(self.lib.ek_timing_end(self._h, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def timing_form(self):
        """candidates per pass of the launches timing_end averaged over"""
        t = C.c_int32()
        _lib.check(self.lib.ek_timing_form(self._h, C.byref(t)))
        return t.value
