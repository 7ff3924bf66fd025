"""k-hybrid: k-centers to place the centers, PAM sweeps to refine them.

Surface follows the reference's enspara/cluster/hybrid.py (KHybrid :28-109,
hybrid() :112-162).  For metric 'rmsd' the frames are uploaded once and both
phases run against the same device-resident state.
"""
import logging
import time

import numpy as np

from ..device import FrameStore, as_xyz
from ..exception import ImproperlyConfigured
from . import kcenters as _kc
from . import kmedoids as _km
from . import util
from .kcenters import BaseEstimator, ClusterMixin, check_random_state

logger = logging.getLogger(__name__)


class KHybrid(BaseEstimator, ClusterMixin, util.MolecularClusterMixin):
    """reference hybrid.py:28-109"""

    def __init__(self, metric, n_clusters=None, cluster_radius=None,
                 kmedoids_updates=5, random_first_center=False,
                 random_state=None, mpi_mode=None, args=None, lengths=None,
                 device=0):
        if n_clusters is None and cluster_radius is None:
            raise ImproperlyConfigured("Either n_clusters or cluster_radius "
                                       "is required for KHybrid clustering")
        self.kmedoids_updates = kmedoids_updates
        self.n_clusters = n_clusters
        self.cluster_radius = cluster_radius
        self.random_first_center = random_first_center
        self.metric = util._get_distance_method(metric)
        self.random_state = check_random_state(random_state)
        self.mpi_mode = util.default_mpi_mode(mpi_mode)
        self.args = args
        self.lengths = lengths
        self.device = device

    def fit(self, X, init_centers=None, args=None):
        t0 = time.perf_counter()
        self.result_ = hybrid(
            X, self.metric, n_iters=self.kmedoids_updates,
            n_clusters=self.n_clusters, dist_cutoff=self.cluster_radius,
            random_first_center=self.random_first_center,
            init_centers=init_centers, random_state=self.random_state,
            mpi_mode=self.mpi_mode, device=self.device)
        self.runtime_ = time.perf_counter() - t0
        return self


def hybrid(X, distance_method, n_iters=5, n_clusters=np.inf, dist_cutoff=0,
           random_first_center=False, init_centers=None, random_state=None,
           mpi_mode=False, args=None, lengths=None, device=0):
    """reference hybrid.py:112-162"""
    distance_method = util._get_distance_method(distance_method)
    if not util.is_device_rmsd(distance_method):
        result = _kc.kcenters(
            X, distance_method, n_clusters=n_clusters, dist_cutoff=dist_cutoff,
            init_centers=init_centers,
            random_first_center=random_first_center, mpi_mode=mpi_mode)
        if n_iters > 0:
            return _km._kmedoids_iterations(
                X, distance_method, n_iters, result.center_indices,
                result.assignments, result.distances,
                random_state=random_state)
        return result

    # same argument handling as kcenters() (kcenters.py:177-193)
    if (n_clusters is np.inf) and (dist_cutoff == 0):
        raise ImproperlyConfigured("Either n_clusters or cluster_radius "
                                   "is required for KHybrid clustering")
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
        # hybrid.py:112-162 in MPI mode: every rank passes its own frames
        from .. import sharded
        return sharded.fit_sharded(
            X, n_clusters=n_clusters, dist_cutoff=dist_cutoff, n_iters=n_iters,
            random_state=random_state, init_centers=init_centers)

    with FrameStore.from_array(as_xyz(X), device=device) as store:
        result = _kc._kcenters_device(X, n_clusters, dist_cutoff, init_centers,
                                      device, store=store)
        if n_iters > 0:
            return _km._kmedoids_iterations_device(
                X, store, n_iters, result.center_indices, None, random_state)
        return result
