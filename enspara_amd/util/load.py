"""Trajectory files -> one concatenated coordinate array, the input form of the
clustering path (reference enspara/util/load.py:52-161 ``load_as_concatenated``).

The reference hands every file to ``mdtraj.load``; mdtraj is not part of this
build, so the formats read here are the ones this package can decode itself:
mdtraj's HDF5 trajectories (``.h5``: a ``coordinates`` dataset ``[frames, atoms,
3]`` in nm, read through ``h5lite``) and plain ``.npy`` arrays of that shape.
The per-file keyword arguments that select data keep mdtraj's meaning:
``stride``, ``atom_indices``, ``frame``; ``top`` is accepted and unused (HDF5
trajectories carry their own topology and coordinates need none).
"""
import os
from multiprocessing.pool import ThreadPool

import numpy as np

from .. import h5lite
from ..exception import DataInvalid, ImproperlyConfigured

_IGNORED = ("top",)
_KNOWN = ("stride", "atom_indices", "frame") + _IGNORED


def _open_coordinates(filename):
    """-> (n_frames, n_atoms, reader) where reader() returns float32 [L, A, 3]."""
    ext = os.path.splitext(filename)[1].lower()
    if ext in (".h5", ".hdf5", ".lh5"):
        f = h5lite.File(filename)
        try:
            if "coordinates" not in f:
                raise DataInvalid(
                    "%s has no 'coordinates' dataset: not an mdtraj HDF5 "
                    "trajectory" % filename)
            node = f["coordinates"]
            shape = node.shape
        except Exception:
            f.close()
            raise

        def reader():
            try:
                return node.read()
            finally:
                f.close()
        reader.close = f.close      # for callers that only want the shape
    elif ext == ".npy":
        arr = np.load(filename, mmap_mode="r")
        shape = arr.shape

        def reader():
            return np.asarray(arr)
    else:
        raise ImproperlyConfigured(
            "Cannot read '%s': only mdtraj HDF5 (.h5) and .npy coordinate "
            "files are supported without mdtraj." % filename)
    if len(shape) != 3 or shape[2] != 3:
        raise DataInvalid("coordinates in %s have shape %s, expected "
                          "(frames, atoms, 3)" % (filename, (shape,)))
    return shape[0], shape[1], reader


def _coordinates_shape(filename):
    """(n_frames, n_atoms) of a coordinate file; whatever was opened to find
    out is closed again before returning."""
    n_frames, n_atoms, reader = _open_coordinates(filename)
    closer = getattr(reader, "close", None)
    if closer is not None:
        closer()
    return n_frames, n_atoms


def _selected_length(n_frames, kw):
    if "frame" in kw:
        return 1
    stride = kw.get("stride") or 1
    return (n_frames + stride - 1) // stride


def _load_one(job):
    filename, kw, out = job
    n_frames, n_atoms, reader = _open_coordinates(filename)
    xyz = reader()
    if "frame" in kw:
        xyz = xyz[kw["frame"]:kw["frame"] + 1] if kw["frame"] != -1 else xyz[-1:]
    else:
        xyz = xyz[::kw.get("stride") or 1]
    if kw.get("atom_indices") is not None:
        xyz = xyz[:, np.asarray(kw["atom_indices"], dtype=np.int64)]
    if xyz.shape[1:] != out.shape[1:]:
        raise DataInvalid(
            "%s gives %d atoms where the first file gave %d"
            % (filename, xyz.shape[1], out.shape[1]))
    if len(xyz) != len(out):
        return len(xyz)
    out[...] = xyz
    return len(xyz)


def load_as_concatenated(filenames, lengths=None, processes=None, args=None,
                         **kwargs):
    """Load many trajectories into one float32 array ``[sum(lengths), atoms, 3]``.

    ``kwargs`` apply to every file, or ``args`` gives one dict per file (not
    both).  ``lengths`` (frames taken from each file, after striding) saves
    opening every file twice; wrong values raise DataInvalid.  Returns
    ``(lengths, xyz)``.
    """
    filenames = list(filenames)
    if kwargs and args:
        raise ImproperlyConfigured(
            "Give the load options either once as keyword arguments or per "
            "file in `args`, not both")
    elif kwargs:
        args = [kwargs] * len(filenames)
    elif args:
        if len(args) != len(filenames):
            raise ImproperlyConfigured(
                "`args` needs one dict per file: %s dicts for %s files"
                % (len(args), len(filenames)))
    else:
        args = [{}] * len(filenames)
    for kw in args:
        for key in kw:
            if key not in _KNOWN:
                raise ImproperlyConfigured(
                    "load option '%s' needs mdtraj; supported here: %s"
                    % (key, ", ".join(_KNOWN)))
    if not filenames:
        raise ImproperlyConfigured("No trajectory files were given.")

    given = lengths is not None
    if given:
        lengths = [int(v) for v in lengths]
        if len(lengths) != len(filenames):
            raise ImproperlyConfigured(
                "%s lengths were given for %s files"
                % (len(lengths), len(filenames)))
    else:
        lengths = [_selected_length(_coordinates_shape(f)[0], kw)
                   for f, kw in zip(filenames, args)]

    n_atoms = _coordinates_shape(filenames[0])[1]
    if args[0].get("atom_indices") is not None:
        n_atoms = len(args[0]["atom_indices"])
    xyz = np.empty((sum(lengths), n_atoms, 3), dtype=np.float32)
    starts = np.concatenate([[0], np.cumsum(lengths)])
    jobs = [(f, kw, xyz[starts[i]:starts[i + 1]])
            for i, (f, kw) in enumerate(zip(filenames, args))]
    if processes and processes > 1 and len(jobs) > 1:
        with ThreadPool(processes) as pool:    # zlib releases the GIL
            got = pool.map(_load_one, jobs)
    else:
        got = [_load_one(job) for job in jobs]
    if got != lengths:
        raise DataInvalid(
            "The given lengths add up to %s frames but the %s files hold %s."
            % (sum(lengths), len(lengths), sum(got)))
    return lengths, xyz
