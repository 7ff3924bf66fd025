"""k-medoids refinement by Partitioning Around Medoids, device resident.

Surface follows the reference's enspara/cluster/kmedoids.py (KMedoids :28-105,
kmedoids() :108-202, _kmedoids_iterations :410-476, _kmedoids_pam_update
:520-699).  For metric 'rmsd' every O(n) pass of a proposal runs on the GPU
(csrc/ek_pam.hip); the host keeps the random stream and the accept test so
that proposals are drawn exactly as the reference draws them.  A callable
metric runs the reference-shaped host loop.
"""
import logging
import time

import numpy as np

from ..device import FrameStore, as_xyz
from ..exception import DataInvalid, ImproperlyConfigured
from . import util
from .kcenters import BaseEstimator, ClusterMixin, check_random_state, _frame_of

logger = logging.getLogger(__name__)


class KMedoids(BaseEstimator, ClusterMixin, util.MolecularClusterMixin):
    """reference kmedoids.py:28-105"""

    def __init__(self, metric, n_clusters=None, n_iters=5, args=None,
                 lengths=None, device=0, mpi_mode=None):
        self.metric = util._get_distance_method(metric)
        self.n_clusters = n_clusters
        self.n_iters = n_iters
        self.args = args
        self.lengths = lengths
        self.device = device
        # None: like the reference, which asks mpi.size() (kmedoids.py:172) --
        # here: an initialised torch.distributed group of more than one rank
        self.mpi_mode = mpi_mode

    def fit(self, X, assignments=None, distances=None,
            cluster_center_inds=None, X_lengths=None, args=None):
        t0 = time.perf_counter()
        self.result_ = kmedoids(
            X, distance_method=self.metric, n_clusters=self.n_clusters,
            n_iters=self.n_iters, assignments=assignments,
            distances=distances, cluster_center_inds=cluster_center_inds,
            X_lengths=X_lengths, device=self.device, mpi_mode=self.mpi_mode)
        self.runtime_ = time.perf_counter() - t0
        return self


def _resolve_inputs(X, distance_method, n_clusters, assignments, distances,
                    cluster_center_inds, X_lengths, random_state):
    """reference _kmedoids_inputs_tree, kmedoids.py:285-363"""
    rng = np.random.default_rng(seed=random_state)
    if (assignments is None) != (distances is None):
        raise ImproperlyConfigured(
            "Assignments and distances need to both be supplied, "
            "or neither supplied.")
    n = len(X._data) if hasattr(X, "_data") else len(X)
    if cluster_center_inds is None:
        if assignments is not None:
            cluster_center_inds = util.find_cluster_centers(assignments,
                                                            distances)
        else:
            cluster_center_inds = np.array([])
            while len(np.unique(cluster_center_inds)) < n_clusters:
                cluster_center_inds = rng.integers(0, n, n_clusters)
    elif hasattr(cluster_center_inds[0], "__len__"):
        cluster_center_inds = [
            sum(X_lengths[:pair[0]]) + pair[1] for pair in cluster_center_inds]
    return cluster_center_inds


def kmedoids(X, distance_method, n_clusters=None, n_iters=5, assignments=None,
             distances=None, cluster_center_inds=None, proposals=None,
             X_lengths=None, args=None, lengths=None, random_state=None,
             device=0, mpi_mode=None):
    """reference kmedoids.py:108-202.  ``mpi_mode``: None = as the reference,
    which takes its MPI branch when ``mpi.size() > 1`` (:172) -- here when an
    initialised torch.distributed group has more than one rank; True / False
    force it.  In that mode every rank passes its own frames and the warm start
    is (assignments, distances, cluster_center_inds, X_lengths), see
    ``enspara_amd.sharded.kmedoids_sharded``."""
    mpi_mode = util.default_mpi_mode(mpi_mode)
    if cluster_center_inds is not None:
        if hasattr(cluster_center_inds[0], "__len__") and X_lengths is None:
            raise ImproperlyConfigured(
                "If cluster_center_inds is given as [[global_traj_id, "
                "frame_id],...] then X_lengths also needs to be supplied")
    if cluster_center_inds is None and n_clusters is None:
        if mpi_mode:
            raise ImproperlyConfigured(
                "Must provide n_clusters or cluster_center_inds, assignments,"
                "and distances for KMedoids in MPI mode.")
        if assignments is None and distances is None:
            raise ImproperlyConfigured(
                "Must provide n_clusters or cluster_center_inds or "
                " (assignments and distances) for KMedoids")
    distance_method = util._get_distance_method(distance_method)
    if mpi_mode:
        if not util.is_device_rmsd(distance_method):
            raise ImproperlyConfigured(
                "KMedoids in MPI mode runs metric 'rmsd' on the device; a "
                "callable metric has no sharded form here")
        if args is not None or lengths is not None:
            raise ImproperlyConfigured(
                "KMedoids in MPI mode takes every rank's own frames; `args` / "
                "`lengths` have no meaning there")
        import torch
        if device not in (0, None) and int(device) != torch.cuda.current_device():
            raise ImproperlyConfigured(
                "KMedoids in MPI mode runs on the process's current device "
                "(%d); device=%r conflicts with it"
                % (torch.cuda.current_device(), device))
        from .. import sharded
        return sharded.kmedoids_fit_sharded(
            X, n_clusters=n_clusters, n_iters=n_iters, assignments=assignments,
            distances=distances, cluster_center_inds=cluster_center_inds,
            X_lengths=X_lengths, proposals=proposals, random_state=random_state)
    inds = _resolve_inputs(X, distance_method, n_clusters, assignments,
                           distances, cluster_center_inds, X_lengths,
                           random_state)
    inds = [int(i) for i in inds]

    if util.is_device_rmsd(distance_method):
        xyz = as_xyz(X)
        with FrameStore.from_array(xyz, device=device) as store:
            if assignments is None:
                store.assign_nearest(xyz[inds])              # :360-361
                # the centers were the medoid frames themselves: every frame's
                # distance is the distance to the medoid its label names
                store.set_option("state_exact", 1)
            else:
                store.upload_state(distances, assignments)
            d0, _ = store.download_state()
            # the medoids must sit at (numerically) zero distance (:197)
            assert np.all(d0[inds] < 0.001)
            return _kmedoids_iterations_device(
                X, store, n_iters, inds, proposals, random_state)

    if assignments is None:
        assignments, distances = util.assign_to_nearest_center(
            X, X[inds], distance_method)
    assert np.all(np.asarray(distances)[inds] < 0.001)
    return _kmedoids_iterations(X, distance_method, n_iters, inds,
                                assignments, distances, proposals=proposals,
                                random_state=random_state)


def _kmedoids_iterations(X, distance_method, n_iters, cluster_center_inds,
                         assignments, distances, proposals=None, args=None,
                         lengths=None, random_state=None, store=None):
    """reference kmedoids.py:410-476.  With metric 'rmsd' and a FrameStore
    holding X and the current state, runs on the device."""
    distance_method = util._get_distance_method(distance_method)
    if util.is_device_rmsd(distance_method):
        own = store is None
        if own:
            store = FrameStore.from_array(as_xyz(X))
            store.upload_state(distances, assignments)
        try:
            return _kmedoids_iterations_device(
                X, store, n_iters, cluster_center_inds, proposals,
                random_state)
        finally:
            if own:
                store.close()
    if hasattr(distance_method, "bind"):     # device metric: X resident for
        distance_method = distance_method.bind(X)   # all sweeps, not per sweep
    result = None
    for i in range(n_iters):
        cluster_center_inds, distances, assignments, centers = \
            _kmedoids_pam_update(X, distance_method, cluster_center_inds,
                                 assignments, distances, proposals=proposals,
                                 random_state=random_state)
        result = util.ClusterResult(center_indices=cluster_center_inds,
                                    assignments=assignments,
                                    distances=distances, centers=centers)
        logger.info("KMedoids update %s", i)
    return result


def _check_proposals(proposals, medoid_inds):
    if proposals is None:
        return
    if len(proposals) != len(medoid_inds):
        raise DataInvalid(
            "Length of 'proposals' didn't match length of 'medoid_inds' "
            "({} != {}).".format(len(proposals), len(medoid_inds)))
    if hasattr(proposals[0], "__len__") != hasattr(medoid_inds[0], "__len__"):
        raise DataInvalid(
            "Depth of 'proposals' didn't match 'medoid_inds' "
            "(proposals[0] == {}, whereas medoid_inds[0] == {})".format(
                proposals[0], medoid_inds[0]))


# Proposals drawn ahead and decided as one window (1..32, the library's
# ek_pam_window_max(); 1 = one distance pass and one read-back per proposal).
# The results do not depend on it: every guess is checked against the state and
# the random stream when its turn comes.
PAM_PREFETCH = 32


class _DrawStream:
    """``random_state.choice(m)`` (kmedoids.py:514 draws a member with
    ``choice(state_inds)``, which is ``state_inds[choice(len(state_inds))]``)
    without a Python-level numpy call per draw, and with free look-ahead.

    numpy's legacy ``RandomState.choice(m)`` / ``randint(0, m)`` take 32-bit
    outputs of the Mersenne Twister, mask them to the bits of ``m - 1`` and
    reject values above it (``m == 1`` consumes nothing).  This class pulls raw
    32-bit outputs from the caller's RandomState in blocks and applies the same
    rule, so every draw is the number ``choice`` would have returned; ``close``
    puts the RandomState where that many ``choice`` calls would have left it.
    Proposals drawn *ahead* for a window are simply the same raw numbers looked
    at early (``peek``): copying a RandomState costs 0.2 ms, more than a whole
    window of proposals takes on the device."""
    BLOCK = 4096

    def __init__(self, random_state):
        self.rs = random_state
        self.state0 = random_state.get_state()
        self.raw = np.empty(0, dtype=np.uint32)
        self.pos = 0            # raw outputs consumed by real draws

    def _need(self, upto):
        while upto > len(self.raw):
            more = self.rs.randint(0, 2 ** 32, size=self.BLOCK, dtype=np.uint32)
            self.raw = np.concatenate([self.raw, more])

    def _draw_at(self, pos, m):
        if m <= 0:
            raise ValueError("a must be greater than 0 unless no samples are "
                             "taken")           # RandomState.choice(0)
        rng = m - 1
        if rng == 0:
            return 0, pos
        if rng > 0xFFFFFFFE:
            raise NotImplementedError("member lists of 2**32 frames and more")
        mask = (1 << rng.bit_length()) - 1
        while True:
            self._need(pos + 1)
            v = int(self.raw[pos]) & mask
            pos += 1
            if v <= rng:
                return v, pos

    def draw(self, m):
        """the next real draw"""
        v, self.pos = self._draw_at(self.pos, int(m))
        return v

    def peek(self, ms):
        """what the next len(ms) real draws will be if the member counts are
        ``ms`` (stops before a count <= 0: that draw raises when it is made)"""
        out, pos = [], self.pos
        for m in ms:
            if m <= 0:
                break
            v, pos = self._draw_at(pos, int(m))
            out.append(v)
        return out

    def close(self):
        self.rs.set_state(self.state0)
        if self.pos:
            self.rs.randint(0, 2 ** 32, size=self.pos, dtype=np.uint32)


class _Window:
    """Clusters [lo, hi) of a sweep with their member counts, the proposals
    guessed for them and the bit mask of clusters whose membership has changed
    since the guesses were made."""
    __slots__ = ("lo", "hi", "m", "j", "frame", "stale")


def _open_window(store, lo, hi, proposals, random_state):
    w = _Window()
    w.lo, w.hi, w.stale = lo, hi, 0
    w.m = [int(x) for x in store.pam_count_members_batch(lo, hi - lo)]
    if proposals is None:
        # the draws the real stream will produce if these counts still hold when
        # each cluster's turn comes (they are made, in order, below)
        w.j = random_state.peek(w.m)
        w.frame = ([int(f) for f in
                    store.pam_select_members_batch(lo, w.j)] if w.j else [])
    else:
        w.j = None
        w.frame = [int(p) for p in proposals[lo:hi]]
    store.pam_prefetch(w.frame, lo, hi - lo)
    return w


# 1: a window's proposals are decided and committed on the device, one
# read-back per window (FrameStore.pam_window_run); 0: one read-back per
# proposal.  Same results.
PAM_DEVICE_DECISIONS = 1

# 1: with PAM_DEVICE_DECISIONS, the loop over the windows of a sweep runs inside
# the library too (ek_pam_sweep: the same calls in the same order, without the
# interpreter between them -- four read-backs per window, and the host came back
# from each 25-60 us later than it had to); 0: the loop below.  Same results.
PAM_NATIVE_SWEEP = 1


def _one_proposal(store, cid, proposals, random_state):
    """kmedoids.py:597-699 for one cluster, counted and drawn right now.
    -> (proposed frame, accepted, old cost, new cost, ambiguous members)"""
    if proposals is None:
        m = store.pam_count_members(cid)                     # :611
        # RandomState.choice(state_inds) == state_inds[choice(len)]
        # (raises ValueError on an empty cluster, like the reference)
        j = random_state.draw(m)                             # :514
        prop, old_cost, new_cost, n_amb = store.pam_propose_member(cid, j)
    else:
        prop = int(proposals[cid])
        old_cost, new_cost, n_amb = store.pam_propose(cid, prop)
    accept = new_cost < old_cost                             # :683
    store.pam_commit(accept)
    return prop, accept, old_cost, new_cost, n_amb


def _pam_sweep_device(store, medoid_inds, proposals, random_state):
    """One sweep of kmedoids.py:575-699 against device-resident state."""
    random_state = check_random_state(random_state)          # :579
    _check_proposals(proposals, medoid_inds)
    stream = _DrawStream(random_state)
    try:
        return _pam_sweep_device_on(store, medoid_inds, proposals, stream)
    finally:
        stream.close()          # the caller's RandomState: as after these draws


def _pam_sweep_device_on(store, medoid_inds, proposals, random_state):
    """(random_state: a _DrawStream over the caller's RandomState)"""
    store.pam_begin(medoid_inds)
    K = len(medoid_inds)
    width = max(1, min(int(PAM_PREFETCH), store.pam_window_max()))
    acceptances = 0
    old_cost = new_cost = float("nan")

    def note(cid, accept, oc, nc, n_amb):
        logger.debug("%s proposed center for k=%s: cost %.5f -> %.5f "
                     "(%d ambiguous).", "Accepted" if accept else "Rejected",
                     cid, oc, nc, n_amb)

    if width > 1 and PAM_DEVICE_DECISIONS and PAM_NATIVE_SWEEP:
        return _pam_sweep_native(store, medoid_inds, proposals, random_state, width)
    cid = 0
    win = None
    while cid < K:
        if width == 1:
            prop, accept, old_cost, new_cost, n_amb = _one_proposal(
                store, cid, proposals, random_state)
            if accept:
                medoid_inds[cid] = prop
                acceptances += 1
            note(cid, accept, old_cost, new_cost, n_amb)
            cid += 1
            continue
        if PAM_DEVICE_DECISIONS:
            # ---- a window decided on the device ---------------------------------
            win = _open_window(store, cid, min(K, cid + width), proposals,
                               random_state)
            n_slots = len(win.frame)     # short if an empty cluster is in the way
            n_done = 0
            if n_slots:
                n_done, acc, ocs, ncs, nas = store.pam_window_run(
                    win.lo, win.frame[:n_slots], win.m[:n_slots],
                    win.hi - win.lo)
                for s in range(n_done):
                    if proposals is None:
                        # the real draws, in order: the member lists are the ones
                        # the guesses were drawn from, so they are the same draws
                        j = random_state.draw(win.m[s])      # :514
                        if j != win.j[s]:
                            raise RuntimeError("PAM window: draw %d for cluster "
                                               "%d, guessed %d"
                                               % (j, win.lo + s, win.j[s]))
                    if acc[s]:
                        medoid_inds[win.lo + s] = win.frame[s]
                        acceptances += 1
                    old_cost, new_cost = float(ocs[s]), float(ncs[s])
                    note(win.lo + s, bool(acc[s]), old_cost, new_cost, int(nas[s]))
            cid = win.lo + n_done
            if cid < win.hi:
                # the window stopped here: this cluster's members changed under an
                # accepted proposal (or it is empty: the draw raises, as the
                # reference's does) -- counted and drawn now, on its own pass
                prop, accept, old_cost, new_cost, n_amb = _one_proposal(
                    store, cid, proposals, random_state)
                if accept:
                    medoid_inds[cid] = prop
                    acceptances += 1
                note(cid, accept, old_cost, new_cost, n_amb)
                cid += 1
            continue
        # ---- one read-back per proposal ---------------------------------------------
        if win is None or cid >= win.hi:
            win = _open_window(store, cid, min(K, cid + width), proposals,
                               random_state)
        slot = cid - win.lo
        exact = not ((win.stale >> slot) & 1)
        counted = False
        if exact:
            m = win.m[slot]
        else:
            m = store.pam_count_members(cid)             # :611
            counted = True
        if proposals is None:
            j = random_state.draw(m)                     # :514
            if exact and slot < len(win.j) and j == win.j[slot]:
                prop = win.frame[slot]
            else:
                if not counted:
                    store.pam_count_members(cid)
                prop = store.pam_select_member(cid, j)
        else:
            prop = win.frame[slot]
        old_cost, new_cost, n_amb, moved = store.pam_propose_ex(
            cid, prop, m, win.lo, win.hi - win.lo)
        accept = new_cost < old_cost                         # :683
        store.pam_commit(accept)
        if accept:
            medoid_inds[cid] = prop
            acceptances += 1
            win.stale |= moved
        note(cid, accept, old_cost, new_cost, n_amb)
        cid += 1
    logger.info("Kmedoid sweep reduced cost to %.7f (%.2f%% acceptance)",
                min(old_cost, new_cost),
                acceptances / len(medoid_inds) * 100)
    return medoid_inds


def _pam_sweep_native(store, medoid_inds, proposals, random_state, width):
    """_pam_sweep_device_on's loop over windows, inside the library
    (FrameStore.pam_sweep_run); random_state: the sweep's _DrawStream."""
    K = len(medoid_inds)
    med = np.ascontiguousarray(medoid_inds, dtype=np.int64)
    accept = np.zeros(K, dtype=np.int32)
    oc = np.zeros(K, dtype=np.float64)
    nc = np.zeros(K, dtype=np.float64)
    na = np.zeros(K, dtype=np.int64)
    cid = 0
    while True:
        if proposals is None:
            # a draw takes two raw outputs on average at worst (rejection above
            # the mask): enough for the rest of the sweep in all likelihood
            random_state._need(random_state.pos + 4 * (K - cid) + 64)
        status, cid, random_state.pos = store.pam_sweep_run(
            width, random_state.raw, random_state.pos, proposals, cid, med, accept,
            oc, nc, na)
        if status == 0:
            break
        if status == 2:
            random_state.draw(0)        # raises what RandomState.choice(0) raises
        random_state._need(len(random_state.raw) + random_state.BLOCK)
    for c in range(K):
        if accept[c]:
            medoid_inds[c] = int(med[c])
    if logger.isEnabledFor(logging.DEBUG):
        for c in range(K):
            logger.debug("%s proposed center for k=%s: cost %.5f -> %.5f "
                         "(%d ambiguous).", "Accepted" if accept[c] else "Rejected",
                         c, oc[c], nc[c], na[c])
    logger.info("Kmedoid sweep reduced cost to %.7f (%.2f%% acceptance)",
                min(oc[K - 1], nc[K - 1]), int(accept.sum()) / K * 100)
    return medoid_inds


def _kmedoids_iterations_device(X, store, n_iters, cluster_center_inds,
                                proposals, random_state):
    medoid_inds = [int(i) for i in cluster_center_inds]
    for i in range(n_iters):
        medoid_inds = _pam_sweep_device(store, medoid_inds, proposals,
                                        random_state)
        logger.info("KMedoids update %s", i)
    d, a = store.download_state()
    return util.ClusterResult(
        center_indices=medoid_inds, assignments=a.astype(np.int64),
        distances=d.astype(np.float64),
        centers=[_frame_of(X, i) for i in medoid_inds])


def _msq(x):
    """reference kmedoids.py:478-479 at one process"""
    return np.square(x).mean()


# 1: a sweep over 'euclidean' / 'manhattan' features runs with the distances, the
# labels and the medoids resident on the device (ek_feat_pam_sweep); 0: the
# reference-shaped loop below around a device metric.  Same results.
PAM_FEATURE_DEVICE = 1
# raw random outputs handed to a resident sweep beyond its position (None: four
# per cluster still to go + 64, which practically never run out; tests set a
# handful to walk the "ran out, call again" path)
FEATURE_RAW_AHEAD = None


def _feature_sweep_applies(X, metric, distances, proposals, assignments=None):
    """A floating-point (or integer) sample matrix without NaN, a finite
    float64 state with integer labels, one of the resident metrics (euclidean,
    manhattan; hamming on integer samples since round 6), the default cost.  (The reference builds ``new_dist = zeros_like(distances)``,
    kmedoids.py:639: with float32 distances every accepted proposal rounds the
    new values to float32 before the next comparison -- the resident sweep
    works in float64 throughout, so it only takes float64 states.)"""
    mid = getattr(metric, "device_metric_id", None)
    if mid not in (0, 1, 2):
        return False
    if mid == 2 and not (isinstance(X, np.ndarray)      # hamming (libdist.pyx:77-95):
                         and np.issubdtype(X.dtype, np.integer)):   # integer samples only
        return False
    if np.asarray(distances).dtype != np.float64:
        return False
    if assignments is not None and not np.issubdtype(
            np.asarray(assignments).dtype, np.integer):
        return False
    if not isinstance(X, np.ndarray) or X.ndim != 2 or X.shape[0] < 1:
        return False
    if not (np.issubdtype(X.dtype, np.floating) or np.issubdtype(X.dtype, np.integer)):
        return False
    if X.dtype.itemsize > 8 or not np.all(np.isfinite(distances)):
        return False
    if np.issubdtype(X.dtype, np.floating) and not np.all(np.isfinite(X)):
        return False
    return True


def _feature_sweep_device(X, metric, medoid_inds, assignments, distances, proposals,
                          random_state):
    """reference kmedoids.py:575-699 (serial branch) on the device; the random
    stream is the caller's: numpy's choice() restated on its raw outputs."""
    from ..geometry import libdist
    bound = metric.bind(X)
    K = len(medoid_inds)
    med = np.ascontiguousarray(medoid_inds, dtype=np.int64).copy()
    d = np.ascontiguousarray(distances, dtype=np.float64).copy()
    a = np.ascontiguousarray(assignments, dtype=np.int32).copy()
    accept = np.zeros(K, dtype=np.int32)
    stream = _DrawStream(random_state)
    try:
        cid = 0
        while True:
            if proposals is None:
                ahead = (4 * (K - cid) + 64 if FEATURE_RAW_AHEAD is None
                         else FEATURE_RAW_AHEAD)
                stream._need(stream.pos + ahead)
            status, cid, stream.pos = bound.res.pam_sweep(
                bound.metric, med, proposals, stream.raw, stream.pos, d, a, accept, cid)
            if status == 0:
                break
            if status == 2:
                stream.draw(0)          # raises what RandomState.choice([]) raises
            stream._need(len(stream.raw) + stream.BLOCK)
    finally:
        stream.close()
    for c in range(K):
        if accept[c]:
            medoid_inds[c] = int(med[c])        # (in place, like the loop below)
    return (medoid_inds, d.astype(np.asarray(distances).dtype),
            a.astype(np.asarray(assignments).dtype), [X[i] for i in medoid_inds])


def _kmedoids_pam_update(X, metric, medoid_inds, assignments, distances,
                         proposals=None, cost=_msq, random_state=None):
    """One PAM sweep for a callable metric (reference kmedoids.py:520-699,
    single-process branch).  For metric 'rmsd' the sweep runs on the device
    and this wrapper moves the state there and back."""
    metric = util._get_distance_method(metric)
    assert np.issubdtype(type(assignments[0]), np.integer)
    assert len(assignments) == len(X) and len(distances) == len(X)
    if util.is_device_rmsd(metric):
        with FrameStore.from_array(as_xyz(X)) as store:
            store.upload_state(distances, assignments)
            inds = _pam_sweep_device(store, [int(i) for i in medoid_inds],
                                     proposals, random_state)
            d, a = store.download_state()
        return (inds, d.astype(np.float64), a.astype(np.int64),
                [_frame_of(X, i) for i in inds])

    random_state = check_random_state(random_state)
    _check_proposals(proposals, medoid_inds)
    if (PAM_FEATURE_DEVICE and cost is _msq and
            _feature_sweep_applies(X, metric, distances, proposals, assignments)):
        return _feature_sweep_device(X, metric, medoid_inds, assignments, distances,
                                     proposals, random_state)
    if hasattr(metric, "bind"):              # device metric: upload X once
        metric = metric.bind(X)
    medoid_coords = [X[i] for i in medoid_inds]
    for cid in range(len(medoid_inds)):
        state_inds = np.where(assignments == cid)[0]
        if proposals is None:
            prop = random_state.choice(state_inds)
        else:
            prop = proposals[cid]
        proposed_center = X[prop]
        nd = metric(X, proposed_center)
        new_dist = np.zeros_like(distances) - 1
        new_assig = np.zeros_like(assignments) - 1
        down = distances > nd
        new_assig[down] = cid
        new_dist[down] = nd[down]
        up_other = (distances <= nd) & (assignments != cid)
        new_assig[up_other] = assignments[up_other]
        new_dist[up_other] = distances[up_other]
        up_this = (distances <= nd) & (assignments == cid)
        trial = list(medoid_coords)
        trial[cid] = proposed_center
        amb_a, amb_d = util.assign_to_nearest_center(X[up_this], trial, metric)
        new_assig[up_this] = amb_a
        new_dist[up_this] = amb_d
        assert np.all(new_assig >= 0) and np.all(new_dist >= 0)
        if cost(new_dist) < cost(distances):
            distances, assignments = new_dist, new_assig
            medoid_coords = trial
            medoid_inds[cid] = prop
    return medoid_inds, distances, assignments, medoid_coords
