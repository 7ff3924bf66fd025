"""k-centers (Gonzalez farthest-point) clustering with the device-resident
RMSD loop.

Surface and semantics follow the reference's enspara/cluster/kcenters.py
(KCenters :18-100, kcenters() :108-240, iteration :243-311).  For metric
'rmsd' the whole loop of :217-231 -- arg-max, distance pass, strict-< update,
max -- runs on the GPU (csrc/ek_kcenters.hip) without a host round trip per
center; frame 0 is always the first center (arg-max of all-inf distances,
:199, :282).  A Python callable metric runs the reference-shaped host loop.
"""
import logging
import time

import numpy as np

from ..device import FrameStore, as_xyz
from ..exception import ImproperlyConfigured
from . import util

logger = logging.getLogger(__name__)

try:  # sklearn is optional plumbing: estimator base classes only
    from sklearn.base import BaseEstimator, ClusterMixin
    from sklearn.utils import check_random_state
except Exception:  # pragma: no cover
    class BaseEstimator(object):
        pass

    class ClusterMixin(object):
        pass

    def check_random_state(seed):
        if seed is None or isinstance(seed, (int, np.integer)):
            return np.random.RandomState(seed)
        return seed


class KCenters(BaseEstimator, ClusterMixin, util.MolecularClusterMixin):
    """sklearn-style k-centers (reference kcenters.py:18-100).

    Parameters are the reference's: ``metric`` ('rmsd' or a callable
    ``f(X, y) -> distances``), ``n_clusters``, ``cluster_radius``,
    ``random_first_center`` (not implemented there either),
    ``random_state``, ``mpi_mode`` (None: True when the torch.distributed group
    has more than one rank, as the reference's ``mpi.size() != 1``,
    kcenters.py:73; every rank then passes its own frames, see
    :mod:`enspara_amd.sharded`).  ``device`` selects the GPU.
    """

    def __init__(self, metric, n_clusters=None, cluster_radius=None,
                 random_first_center=False, random_state=None, mpi_mode=None,
                 device=0):
        if n_clusters is None and cluster_radius is None:
            raise ImproperlyConfigured("Either n_clusters or cluster_radius "
                                       "is required for KHybrid clustering")
        self.metric = util._get_distance_method(metric)
        self.n_clusters = n_clusters
        self.cluster_radius = cluster_radius
        self.random_first_center = random_first_center
        self.random_state = check_random_state(random_state)
        self.mpi_mode = util.default_mpi_mode(mpi_mode)
        self.device = device

    def fit(self, X, init_centers=None):
        t0 = time.perf_counter()
        self.result_ = kcenters(
            X, distance_method=self.metric, n_clusters=self.n_clusters,
            dist_cutoff=self.cluster_radius, init_centers=init_centers,
            random_first_center=self.random_first_center,
            mpi_mode=self.mpi_mode, device=self.device)
        self.runtime_ = time.perf_counter() - t0
        return self


def kcenters_mpi(*args, **kwargs):
    """reference kcenters.py:103-105"""
    kwargs.pop('mpi_mode', None)
    return kcenters(*args, mpi_mode=True, **kwargs)


def kcenters(traj, distance_method, n_clusters=np.inf, dist_cutoff=0,
             init_centers=None, random_first_center=False,
             use_triangle_inequality=False, mpi_mode=False, device=0):
    """Function form (reference kcenters.py:108-240).  Returns a
    ClusterResult(center_indices list, distances float64, assignments int64,
    centers list of frames)."""
    if (n_clusters is np.inf) and (dist_cutoff == 0):
        raise ImproperlyConfigured("Either n_clusters or cluster_radius "
                                   "is required for KHybrid clustering")
    distance_method = util._get_distance_method(distance_method)
    if n_clusters is None and dist_cutoff is None:
        raise ImproperlyConfigured(
            "KCenters must specify 'n_clusters' or 'distance_cutoff'")
    elif n_clusters is None:
        n_clusters = np.inf
    elif dist_cutoff is None:
        dist_cutoff = 0
    if random_first_center:
        raise NotImplementedError(
            "We haven't implemented kcenters 'random_first_center' yet.")
    if mpi_mode:
        # every rank of the torch.distributed group passes its own frames
        # (kcenters.py:314-378); RMSD only
        if not util.is_device_rmsd(distance_method):
            raise ImproperlyConfigured(
                "mpi_mode is available for metric 'rmsd' "
                "(one process per GPU over torch.distributed)")
        from .. import sharded
        return sharded.fit_sharded(
            traj, n_clusters=n_clusters, dist_cutoff=dist_cutoff, n_iters=0,
            use_triangle_inequality=use_triangle_inequality,
            init_centers=init_centers)

    if util.is_device_rmsd(distance_method):
        return _kcenters_device(traj, n_clusters, dist_cutoff, init_centers,
                                device,
                                use_triangle_inequality=use_triangle_inequality)
    return _kcenters_host(traj, distance_method, n_clusters, dist_cutoff,
                          init_centers, use_triangle_inequality)


def _frame_of(traj, i):
    """traj[i] in the caller's own container (what the reference returns as
    a center, kcenters.py:283)."""
    if hasattr(traj, "_data") and hasattr(traj, "lengths"):
        return traj._data[i]
    return traj[i]


def _kcenters_device(traj, n_clusters, dist_cutoff, init_centers, device,
                     store=None, use_triangle_inequality=False):
    """``use_triangle_inequality`` (reference kcenters.py:287-296; like there,
    reachable from this function only, never set by KCenters.fit): one center
    per pass, and the tiles of 256 frames none of whose frames can come closer
    to the new center than to their own are not read.  Same result; pays when
    neighbouring frames share a cluster (trajectories in time order)."""
    xyz = as_xyz(traj) if store is None else None
    own = store is None
    if own:
        store = FrameStore.from_array(xyz, device=device)
    try:
        store.set_option("triangle", 1 if use_triangle_inequality else 0)
        n = store.n
        if n == 0:
            raise ValueError("cannot cluster an empty trajectory")
        if init_centers is None:
            ctr_inds, centers = [], []
            store.reset_state()
        else:
            centers = [c for c in init_centers]
            logger.info("Updating assignments to previous cluster centers")
            store.assign_nearest(util._stack_centers(centers))
            d0, a0 = store.download_state()
            ctr_inds = list(util.find_cluster_centers(a0.astype(np.int64),
                                                      d0.astype(np.float64)))
        budget = n_clusters - len(ctr_inds)
        # A finite n_clusters is honoured exactly (the reference keeps adding
        # centers -- repeats of the frame with the largest residual
        # self-distance -- once every frame is a center).  With only a
        # distance cut-off the reference would loop forever if the cut-off lay
        # below that residual; the request is capped at 2n + 16 here.
        if budget <= 0:
            max_new = 0
        elif np.isinf(budget):
            max_new = 2 * n + 16
        else:
            max_new = int(budget)
        new_idx, new_d, maxdist = store.kcenters_run(
            len(ctr_inds), max_new, float(dist_cutoff))
        d, a = store.download_state()
    finally:
        if own:
            store.close()
    for i in new_idx:
        ctr_inds.append(int(i))
        centers.append(_frame_of(traj, int(i)))
    logger.info("Terminated k-centers with n=%s and d=%0.6f.",
                len(ctr_inds), maxdist)
    return util.ClusterResult(center_indices=ctr_inds,
                              assignments=a.astype(np.int64),
                              distances=d.astype(np.float64),
                              centers=centers)


def _kcenters_host(traj, distance_method, n_clusters, dist_cutoff,
                   init_centers, use_triangle_inequality):
    """The reference's loop for an arbitrary callable metric
    (kcenters.py:195-311)."""
    # 'euclidean' / 'manhattan' (/ hamming) on a plain 2-D array: the loop runs
    # on the device with the state resident there (csrc/ek_features.hip
    # ek_feat_kcenters: same arithmetic per distance, same strict-< update, same
    # first-index arg-max) instead of one device metric call plus numpy passes
    # per center.  The triangle-inequality variant keeps the host loop.
    mid = getattr(distance_method, "device_metric_id", None)
    # (real floating or integer data without NaN only: np.argmax / .max() treat a
    # NaN distance as the maximum and the reference's loop stops on it, which the
    # device arg-max does not reproduce; anything else keeps the host loop)
    resident = (mid is not None and not use_triangle_inequality
                and isinstance(traj, np.ndarray) and traj.ndim == 2
                and len(traj) > 0
                and (np.issubdtype(traj.dtype, np.floating)
                     or np.issubdtype(traj.dtype, np.integer))
                and traj.dtype != np.float16
                and not (np.issubdtype(traj.dtype, np.floating)
                         and np.isnan(traj).any()))
    if not resident and hasattr(distance_method, "bind"):
        distance_method = distance_method.bind(traj)   # upload traj once
    if init_centers is None:
        ctr_inds, centers = [], []
        assignments = np.full(len(traj), -1, dtype=int)
        distances = np.full(len(traj), np.inf, dtype=float)
    else:
        centers = [c for c in init_centers]
        assignments, distances = util.assign_to_nearest_center(
            traj, centers, distance_method)
        ctr_inds = list(util.find_cluster_centers(assignments, distances))

    if resident and np.isnan(distances).any():     # (a warm start's distances)
        resident = False
        if hasattr(distance_method, "bind"):
            distance_method = distance_method.bind(traj)
    if resident:
        from ..geometry import libdist
        budget = n_clusters - len(ctr_inds)
        max_new = (0 if budget <= 0 else
                   (2 * len(traj) + 16 if np.isinf(budget) else int(budget)))
        new_idx, distances, assignments, _ = libdist.kcenters_resident(
            traj, mid, len(ctr_inds), max_new, float(dist_cutoff), distances,
            assignments)
        for i in new_idx:
            ctr_inds.append(int(i))
            centers.append(traj[int(i)])
        return util.ClusterResult(center_indices=ctr_inds,
                                  assignments=assignments,
                                  distances=distances, centers=centers)

    maxdist = distances.max()
    while (len(ctr_inds) < n_clusters) and (maxdist > dist_cutoff):
        new_index = int(np.argmax(distances))
        new_center = traj[new_index]
        if use_triangle_inequality and np.all(assignments >= 0):
            # kcenters.py:287-296: a frame closer to its center than half the
            # center-to-new-center distance cannot move
            cc = distance_method(traj[ctr_inds], new_center)
            redo = distances > (cc[assignments] / 2)
            dist = distances.copy()
            dist[redo] = distance_method(traj[redo], new_center)
        else:
            dist = distance_method(traj, new_center)
        assert len(dist.shape) == len(distances.shape)
        closer = dist < distances
        distances[closer] = dist[closer]
        assignments[closer] = len(ctr_inds)
        ctr_inds.append(new_index)
        centers.append(new_center)
        maxdist = distances.max()
        logger.debug("Center %s gives max dist of %.6f", len(ctr_inds), maxdist)
    return util.ClusterResult(center_indices=ctr_inds, assignments=assignments,
                              distances=distances, centers=centers)
