"""Shared pieces of the clustering surface: result tuple, metric resolution,
nearest-center assignment, center lookup, predict mixin.

Mirrors the algorithmic part of the reference's enspara/cluster/util.py
(:46-242, :289-313).  For metric 'rmsd' every O(n) pass runs on the GPU through
:class:`enspara_amd.device.FrameStore`; a Python callable metric keeps the
reference's host loops (that is the reference's own plug-in contract,
enspara/cluster/kcenters.py:132-137), it is not a fallback for 'rmsd'.
"""
import logging
from collections import namedtuple

import numpy as np

from .. import ra
from ..device import FrameStore, as_xyz
from ..exception import DataInvalid, ImproperlyConfigured

logger = logging.getLogger(__name__)


class _DeviceRmsd:
    """The callable that the string 'rmsd' resolves to
    (reference util.py:290-291 resolves it to mdtraj.rmsd).

    ``rmsd(X, y)`` -> float32 distances of every frame of X to the frame y,
    computed by the HIP one-center-vs-all kernel.  The algorithms below
    recognise this object and keep frames and state resident on the device
    instead of calling it once per center.
    """
    name = "rmsd"

    def __call__(self, X, y):
        with FrameStore.from_array(X) as st:
            return st.rmsd_to_xyz(y)

    def __repr__(self):
        return "<enspara_amd device metric 'rmsd'>"


rmsd = _DeviceRmsd()


def is_device_rmsd(metric):
    return isinstance(metric, _DeviceRmsd)


def _get_distance_method(metric):
    """reference util.py:289-313"""
    if isinstance(metric, str):
        if metric == "rmsd":
            return rmsd
        if metric == "euclidean":
            from ..geometry.libdist import euclidean
            return euclidean
        if metric in ("cityblock", "manhattan"):
            from ..geometry.libdist import manhattan
            return manhattan
        raise ImproperlyConfigured(
            "'{}' is not a recognized metric".format(metric))
    if callable(metric):
        return metric
    raise ImproperlyConfigured(
        "'{}' is not a recognized metric".format(metric))


class ClusterResult(namedtuple("ClusterResult",
                               ["center_indices", "distances", "assignments",
                                "centers"])):
    """reference util.py:105-156"""

    def partition(self, lengths):
        """Split the per-frame arrays per trajectory: ndarrays when every
        trajectory has the same length, RaggedArrays otherwise; center indices
        become (trajectory, frame) pairs."""
        if all(lengths[0] == n for n in lengths):
            return ClusterResult(
                assignments=np.array(ra.partition_list(self.assignments,
                                                       lengths)),
                distances=np.array(ra.partition_list(self.distances, lengths)),
                center_indices=ra.partition_indices(self.center_indices,
                                                    lengths),
                centers=self.centers)
        return ClusterResult(
            assignments=ra.RaggedArray(self.assignments, lengths=lengths),
            distances=ra.RaggedArray(self.distances, lengths=lengths),
            center_indices=ra.partition_indices(self.center_indices, lengths),
            centers=self.centers)


def _stack_centers(cluster_centers):
    """list of frames / Trajectory / array -> float32 [K, A, 3]"""
    if hasattr(cluster_centers, "xyz"):
        return as_xyz(cluster_centers)
    if isinstance(cluster_centers, np.ndarray) and cluster_centers.ndim == 3:
        return as_xyz(cluster_centers)
    rows = [as_xyz(c) for c in cluster_centers]
    if not rows:
        return np.zeros((0, 1, 3), dtype=np.float32)
    return np.concatenate(rows, axis=0)


def assign_to_nearest_center(trajectory, cluster_centers, distance_method):
    """reference util.py:159-205.  Returns (assignments int64, distances
    float64).  Lowest center index wins ties (strict <)."""
    distance_method = _get_distance_method(distance_method)
    if is_device_rmsd(distance_method):
        centers = _stack_centers(cluster_centers)
        store = trajectory if isinstance(trajectory, FrameStore) else None
        own = store is None
        if own:
            store = FrameStore.from_array(trajectory)
        try:
            store.assign_nearest(centers)
            d, a = store.download_state()
        finally:
            if own:
                store.close()
        return a.astype(np.int64), d.astype(np.float64)

    if hasattr(distance_method, "bind"):     # device metric: upload once
        distance_method = distance_method.bind(trajectory)
    assignments = np.zeros(len(trajectory), dtype=int)
    distances = np.full(len(trajectory), np.inf, dtype=float)
    if len(cluster_centers) > len(trajectory) and hasattr(cluster_centers,
                                                           "xyz"):
        for i, frame in enumerate(trajectory):        # util.py:193-197
            d = distance_method(cluster_centers, frame)
            assignments[i] = np.argmin(d)
            distances[i] = np.min(d)
    else:
        for i, center in enumerate(cluster_centers):  # util.py:199-203
            d = distance_method(trajectory, center)
            closer = d < distances
            distances[closer] = d[closer]
            assignments[closer] = i
    return assignments, distances


def find_cluster_centers(assignments, distances):
    """reference util.py:208-242: for each occupied label, the index of its
    member with the smallest distance (first one on ties)."""
    assignments = np.asarray(assignments)
    distances = np.asarray(distances)
    if len(distances) != len(assignments):
        raise DataInvalid(
            "Length of distances (%s) must match length of assignments "
            "(%s)." % (len(distances), len(assignments)))
    if len(assignments) == 0:
        return np.zeros(0, dtype=assignments.dtype)
    # stable sort by (label, distance, index): the first entry of every label
    # run is its first minimum.  O(n log n) instead of the reference's
    # O(n * labels) scan; same result.
    order = np.lexsort((np.arange(len(assignments)), distances, assignments))
    sorted_labels = assignments[order]
    first = np.ones(len(order), dtype=bool)
    first[1:] = sorted_labels[1:] != sorted_labels[:-1]
    return order[first].astype(assignments.dtype)


class MolecularClusterMixin:
    """reference util.py:46-102"""

    def predict(self, X):
        if not hasattr(self, "result_"):
            raise ImproperlyConfigured(
                "To predict the clustering result for new data, the "
                "clusterer first must have fit some data.")
        pred_assigs, pred_dists = assign_to_nearest_center(
            trajectory=X, cluster_centers=self.centers_,
            distance_method=self.metric)
        pred_centers = find_cluster_centers(pred_assigs, pred_dists)
        return ClusterResult(assignments=pred_assigs, distances=pred_dists,
                             center_indices=pred_centers,
                             centers=self.centers_)

    @property
    def labels_(self):
        return self.result_.assignments

    @property
    def distances_(self):
        return self.result_.distances

    @property
    def center_indices_(self):
        return self.result_.center_indices

    @property
    def centers_(self):
        return self.result_.centers


# ---------------------------------------------------------------------------
# result files (reference enspara/cluster/util.py:464-547): what the clustering
# workflow leaves on disk after a fit -- center indices (.npy), the centers
# themselves, and per-trajectory assignments / distances (ra.save: HDF5)
# ---------------------------------------------------------------------------
def _intermediate_path(path, intermediate_n):
    """<dir>/intermediate-<n>/<basename>, directory created (util.py:467-470)"""
    import os
    d = os.path.join(os.path.dirname(path), "intermediate-%s" % intermediate_n)
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, os.path.basename(path))


def write_centers_indices(path, indices, intermediate_n=None):
    """reference util.py:464-478: np.save of the center indices ((trajectory,
    frame) pairs after ClusterResult.partition); a falsy path writes nothing."""
    if not path:
        logger.info("--center-indices not provided, not writing center "
                    "indices to file.")
        return
    if intermediate_n is not None:
        path = _intermediate_path(path, intermediate_n)
    with open(path, "wb") as f:
        np.save(f, indices)


def write_centers(result, args, intermediate_n=None, load_center_frames=None):
    """reference util.py:481-508.  ``args`` carries ``features``,
    ``center_features`` and (coordinates only) ``trajectories`` /
    ``topologies`` / ``subsample``.

    Feature clustering: the centers array is saved (np.save; ra.save for an
    intermediate result, as the reference does).  Coordinate clustering: the
    reference re-reads the center frames from the trajectory files with mdtraj
    (load_asymm_frames) and pickles the list of md.Trajectory objects;
    ``load_center_frames(center_indices, args)`` is that reader when the
    caller has one, otherwise ``result.centers`` -- the centers' coordinates
    as the fit returned them -- is what gets pickled."""
    import os
    import pickle
    if getattr(args, "features", None):
        if intermediate_n is not None:
            ra.save(_intermediate_path(args.center_features, intermediate_n),
                    result.centers)
        else:
            np.save(args.center_features, result.centers)
        return
    if intermediate_n is not None:
        outdir = os.path.join(os.path.dirname(args.center_features),
                              "intermediate-%s" % intermediate_n)
    else:
        outdir = os.path.dirname(args.center_features)
    logger.info("Saving cluster centers at %s", outdir)
    if outdir:
        os.makedirs(outdir, exist_ok=True)
    if load_center_frames is not None:
        centers = load_center_frames(result.center_indices, args)
    else:
        centers = list(result.centers)
    # (the reference writes to args.center_features in both cases, :506)
    with open(args.center_features, "wb") as f:
        pickle.dump(centers, f)


def write_assignments_and_distances_with_reassign(result, args,
                                                  intermediate_n=None,
                                                  load_targets=None):
    """reference util.py:511-547.  With ``args.subsample == 1`` the fit's own
    per-frame results are saved (ra.save, ndarray or RaggedArray).  With
    subsampling, unless ``args.no_reassign``, every frame of every trajectory
    is assigned to its nearest center on the device first (cluster.reassign);
    ``load_targets(args)`` supplies the trajectories to reassign -- arrays of
    coordinates or zero-argument callables returning them -- where the
    reference reads args.trajectories with mdtraj."""
    if args.subsample == 1:
        logger.debug("Subsampling was 1, not reassigning.")
        dist, assig = result.distances, result.assignments
        final = intermediate_n is None
    elif not getattr(args, "no_reassign", False):
        logger.debug("Reassigning data from subsampling of %s", args.subsample)
        if load_targets is None:
            raise ImproperlyConfigured(
                "reassignment after subsampled clustering needs the "
                "trajectories: pass load_targets(args) (the reference reads "
                "args.trajectories with mdtraj, which this build does not "
                "depend on)")
        from .reassign import reassign
        assig, dist = reassign(load_targets(args), centers=result.centers)
        final = True            # :544-545: the final files are written as well
    else:
        logger.debug("Got --no-reassign, not doing reassigment")
        return
    if intermediate_n is not None:
        ra.save(_intermediate_path(args.distances, intermediate_n), dist)
        ra.save(_intermediate_path(args.assignments, intermediate_n), assig)
    if final:
        ra.save(args.distances, dist)
        ra.save(args.assignments, assig)


def default_mpi_mode(mpi_mode=None):
    """``mpi_mode=None`` of KCenters / KHybrid / KMedoids: the reference decides by
    ``mpi.size() != 1`` (kcenters.py:73, hybrid.py:79, kmedoids.py:172); here: by
    whether an initialised torch.distributed group has more than one rank.  The
    three estimators share this rule; the FUNCTIONS ``kcenters`` / ``hybrid``
    default to False as the reference's do (kcenters.py:110, hybrid.py:115)."""
    if mpi_mode is not None:
        return bool(mpi_mode)
    try:
        import torch.distributed as dist
    except ImportError:
        return False
    return bool(dist.is_available() and dist.is_initialized()
                and dist.get_world_size() > 1)
