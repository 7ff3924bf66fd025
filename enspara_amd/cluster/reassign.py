"""Batched reassignment of many trajectories to existing cluster centers.

The embarrassingly parallel consumer of the multi-center kernel: the step
that follows clustering in the reference's workflow
(enspara/cluster/util.py: compute_batches :549-566, determine_batch_size
:569-579, batch_reassign :582-649, reassign :652-734).  The reference sizes
batches from host RAM and loads files with mdtraj; here a batch is what fits a
fraction of the GPU's HBM, and a "target" is either an array of coordinates
``[L_i, n_atoms, 3]`` or a zero-argument callable returning one (so callers can
plug any reader).  Frames are centred on the device as they are loaded
(:624-625 does it with md.Trajectory.center_coordinates()).
"""
import logging
import time

import numpy as np

from .. import ra
from ..device import FrameStore, as_xyz
from ..exception import ImproperlyConfigured
from . import util

logger = logging.getLogger(__name__)


def compute_batches(lengths, batch_size):
    """Consecutive trajectories are packed while the batch stays below
    ``batch_size`` frames (reference util.py:549-566, same rule: a batch is
    closed when adding the next trajectory would reach batch_size)."""
    sizes = [[]]
    indices = [[]]
    for i, n in enumerate(lengths):
        if sum(sizes[-1]) + n < batch_size:
            sizes[-1].append(n)
            indices[-1].append(i)
        else:
            sizes.append([n])
            indices.append([i])
    return indices


def determine_batch_size(n_atoms, dtype_bytes, frac_mem, device=0):
    """Frames per batch so that the device-resident layout (coordinates once,
    plus per-frame state) uses ``frac_mem`` of the GPU's memory
    (reference util.py:569-579 uses host RAM)."""
    import torch
    free, total = torch.cuda.mem_get_info(device)
    bytes_per_frame = n_atoms * 3 * dtype_bytes * 2 + 64   # staging + tiles
    batch_size = int(total * frac_mem / bytes_per_frame)
    return batch_size, batch_size * bytes_per_frame / 1024 ** 3


def _load(target):
    return as_xyz(target() if callable(target) else target)


def batch_reassign(targets, centers, lengths, frac_mem=0.5, n_procs=None,
                   device=0, batch_size=None):
    """-> (list of per-trajectory assignments, list of distances)
    (reference util.py:582-649)."""
    centers = util._stack_centers(centers)
    if batch_size is None:
        batch_size, batch_gb = determine_batch_size(centers.shape[1], 4,
                                                    frac_mem, device)
        logger.info("Batch max size set to %s frames (~%.2f GB of HBM).",
                    batch_size, batch_gb)
    if batch_size < max(lengths):
        raise ImproperlyConfigured(
            'Batch size of %s was smaller than largest file (size %s).' %
            (batch_size, max(lengths)))
    batches = compute_batches(lengths, batch_size)
    assignments, distances = [], []
    for b, idx in enumerate(batches):
        if not idx:
            continue
        tick = time.perf_counter()
        blens = [int(lengths[i]) for i in idx]
        n = sum(blens)
        with FrameStore(n, centers.shape[1], device=device) as store:
            # trajectories go in at tile-aligned offsets only when they are
            # loaded as one block: concatenate on the host per batch
            xyz = np.concatenate([_load(targets[i]) for i in idx])
            if len(xyz) != n:
                raise ImproperlyConfigured(
                    "lengths do not match the loaded trajectories")
            store.load(xyz)
            store.assign_nearest(centers)
            d, a = store.download_state()
        a = a.astype(np.int64)
        d = d.astype(np.float64)
        assignments.extend(ra.partition_list(a, blens))
        distances.extend(ra.partition_list(d, blens))
        logger.info("Finished batch %s of %s in %.1f seconds.", b + 1,
                    len(batches), time.perf_counter() - tick)
    return assignments, distances


def reassign(trajectories, centers, frac_mem=0.5, device=0, batch_size=None):
    """Assign every frame of every trajectory to its nearest center
    (reference util.py:652-734 without the file handling).  Returns
    (assignments, distances) as ndarrays when all trajectories have the same
    length and as RaggedArrays otherwise."""
    trajectories = list(trajectories)
    loaded = [None] * len(trajectories)
    lengths = []
    for i, t in enumerate(trajectories):
        if callable(t):
            loaded[i] = _load(t)
            lengths.append(len(loaded[i]))
        else:
            lengths.append(len(as_xyz(t)))
    targets = [loaded[i] if loaded[i] is not None else t
               for i, t in enumerate(trajectories)]
    assignments, distances = batch_reassign(
        targets, centers, lengths, frac_mem=frac_mem, device=device,
        batch_size=batch_size)
    if all(len(assignments[0]) == len(a) for a in assignments):
        return np.array(assignments), np.array(distances)
    return ra.RaggedArray(assignments), ra.RaggedArray(distances)
