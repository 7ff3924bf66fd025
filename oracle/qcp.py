"""ctypes binding of oracle/qcp_oracle.c -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Every function mirrors one piece of the reference's hot path; the C source
cites the reference file:line each one follows.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

TILE = 256

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build()
        L = C.CDLL(path)
        f32p = C.POINTER(C.c_float)
        i32p = C.POINTER(C.c_int32)
        i64p = C.POINTER(C.c_int64)
        L.eko_abi_version.restype = C.c_int
        L.eko_num_threads.restype = C.c_int
        L.eko_set_num_threads.argtypes = [C.c_int]
        L.eko_set_num_threads.restype = None
        L.eko_msd_from_S.restype = C.c_double
        f64p = C.POINTER(C.c_double)
        L.eko_msd_from_S.argtypes = [f32p, C.c_double, C.c_double, C.c_int]
        L.eko_center_and_trace.argtypes = [f32p, C.c_int64, C.c_int, f32p, f64p]
        L.eko_rmsd_one_to_many.argtypes = [f32p, f64p, C.c_int64, C.c_int,
                                           f32p, C.c_double, f32p]
        L.eko_S_one_to_many.argtypes = [f32p, C.c_int64, C.c_int, f32p, f32p]
        L.eko_tiled_floats.restype = C.c_int64
        L.eko_tiled_floats.argtypes = [C.c_int64, C.c_int]
        L.eko_to_tiled.argtypes = [f32p, C.c_int64, C.c_int, f32p]
        L.eko_kcenters_step_tiled.argtypes = [
            f32p, f64p, C.c_int64, C.c_int, f32p, C.c_double, C.c_int32,
            f32p, i32p, f32p, i64p]
        L.eko_rmsd_one_to_many_tiled.argtypes = [
            f32p, f64p, C.c_int64, C.c_int, f32p, C.c_double, f32p]
        L.eko_min_update.argtypes = [f32p, C.c_int64, C.c_int32, f32p, i32p]
        L.eko_argmax_first.restype = C.c_int64
        L.eko_argmax_first.argtypes = [f32p, C.c_int64]
        L.eko_assign_nearest.argtypes = [f32p, f64p, C.c_int64, C.c_int,
                                         f32p, f64p, C.c_int32, i32p, f32p]
        L.eko_pam_trial.argtypes = [
            f32p, f32p, f64p, C.c_int64, C.c_int, f32p, f64p, C.c_int32,
            C.c_int32, f32p, C.c_double, f64p, i64p, f64p, i64p, f32p]
        L.eko_pam_trial.restype = None
        _lib = L
    return _lib


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def num_threads():
    return lib().eko_num_threads()


def set_num_threads(n):
    """OpenMP threads for the calls that follow (bench.py's 1-thread figure)."""
    lib().eko_set_num_threads(int(n))


def as_xyz(X):
    """Coordinates of a trajectory-like as C-contiguous float32 [n, A, 3]."""
    if hasattr(X, "xyz"):
        X = X.xyz
    X = np.ascontiguousarray(X, dtype=np.float32)
    if X.ndim == 2:
        X = X[None]
    assert X.ndim == 3 and X.shape[2] == 3, X.shape
    return X


def center_and_trace(xyz):
    """-> (centred float32 [n, A, 3], traces float64 [n])."""
    xyz = as_xyz(xyz)
    n, A, _ = xyz.shape
    out = np.empty_like(xyz)
    G = np.empty(n, dtype=np.float64)
    lib().eko_center_and_trace(_f32(xyz), n, A, _f32(out), _f64(G))
    return out, G


def msd_from_S(S, Gx, Gy, n_atoms):
    S = np.ascontiguousarray(S, dtype=np.float32).reshape(9)
    return lib().eko_msd_from_S(_f32(S), float(Gx), float(Gy), int(n_atoms))


def S_matrices(cframes, ccenter):
    cframes = as_xyz(cframes)
    ccenter = np.ascontiguousarray(ccenter, dtype=np.float32)
    n, A, _ = cframes.shape
    out = np.empty((n, 9), dtype=np.float32)
    lib().eko_S_one_to_many(_f32(cframes), n, A, _f32(ccenter), _f32(out))
    return out


def rmsd_centered(cframes, G, ccenter, Gc):
    """One centred center vs all centred frames -> float32 [n]."""
    cframes = as_xyz(cframes)
    n, A, _ = cframes.shape
    G = np.ascontiguousarray(G, dtype=np.float64)
    ccenter = np.ascontiguousarray(ccenter, dtype=np.float32).reshape(A, 3)
    out = np.empty(n, dtype=np.float32)
    lib().eko_rmsd_one_to_many(_f32(cframes), _f64(G), n, A, _f32(ccenter),
                               float(Gc), _f32(out))
    return out


def rmsd(X, y):
    """The metric callable the reference plugs in as 'rmsd'
    (enspara/cluster/util.py:289-291): distances of every frame of ``X`` to the
    single frame ``y``, float32 [len(X)].  Both are centred here on every call,
    like mdtraj.rmsd with precentered=False."""
    cx, G = center_and_trace(X)
    cy, Gy = center_and_trace(y)
    return rmsd_centered(cx, G, cy[0], Gy[0])


class Prepared:
    """Frames centred once (enspara/cluster/util.py:624-629 does the same for
    reassignment), in both AoS and the frame-minor tiled layout."""

    def __init__(self, X):
        self.xyz = as_xyz(X)
        self.n, self.A, _ = self.xyz.shape
        self.c, self.G = center_and_trace(self.xyz)
        self._tiled = None

    @property
    def tiled(self):
        if self._tiled is None:
            t = np.empty(lib().eko_tiled_floats(self.n, self.A),
                         dtype=np.float32)
            lib().eko_to_tiled(_f32(self.c), self.n, self.A, _f32(t))
            self._tiled = t
        return self._tiled

    def rmsd_to_frame(self, i):
        return rmsd_centered(self.c, self.G, self.c[i], self.G[i])

    def rmsd_to_frame_tiled(self, i):
        out = np.empty(self.n, dtype=np.float32)
        ctr = np.ascontiguousarray(self.c[i])
        lib().eko_rmsd_one_to_many_tiled(_f32(self.tiled), _f64(self.G),
                                         self.n, self.A, _f32(ctr),
                                         float(self.G[i]), _f32(out))
        return out

    def kcenters_step(self, center_c, Gc, label, dist, assign):
        """In-place fused iteration on float32 dist / int32 assign.
        Returns (max of updated dist, its first index)."""
        mx = C.c_float()
        am = C.c_int64()
        ctr = np.ascontiguousarray(center_c, dtype=np.float32)
        lib().eko_kcenters_step_tiled(
            _f32(self.tiled), _f64(self.G), self.n, self.A, _f32(ctr),
            float(Gc), int(label), _f32(dist), _i32(assign),
            C.byref(mx), C.byref(am))
        return mx.value, am.value


def pam_trial(P, med_c, med_G, cid, prop_c, prop_G, dist, assign, new_dist,
              new_assig, scratch):
    """The trial state of one PAM proposal (kmedoids.py:637-678) into
    ``new_dist`` / ``new_assig`` (float64 / int64 [n], overwritten)."""
    i64p = C.POINTER(C.c_int64)
    prop_c = np.ascontiguousarray(prop_c, dtype=np.float32)
    assert med_c.dtype == np.float32 and med_c.flags.c_contiguous
    assert med_G.dtype == np.float64 and dist.dtype == np.float64
    assert assign.dtype == np.int64 and new_assig.dtype == np.int64
    assert new_dist.dtype == np.float64 and scratch.dtype == np.float32
    lib().eko_pam_trial(
        _f32(P.tiled), _f32(P.c), _f64(P.G), P.n, P.A, _f32(med_c),
        _f64(med_G), int(med_c.shape[0]), int(cid), _f32(prop_c),
        float(prop_G), _f64(dist), assign.ctypes.data_as(i64p),
        _f64(new_dist), new_assig.ctypes.data_as(i64p), _f32(scratch))


def assign_nearest(cframes, G, ccenters, Gc):
    """Center-major nearest-center assignment (util.py:199-203 semantics:
    strict <, lowest center index wins ties) -> (int32 [n], float32 [n])."""
    cframes = as_xyz(cframes)
    ccenters = as_xyz(ccenters)
    n, A, _ = cframes.shape
    K = ccenters.shape[0]
    G = np.ascontiguousarray(G, dtype=np.float64)
    Gc = np.ascontiguousarray(Gc, dtype=np.float64)
    assign = np.empty(n, dtype=np.int32)
    dist = np.empty(n, dtype=np.float32)
    lib().eko_assign_nearest(_f32(cframes), _f64(G), n, A, _f32(ccenters),
                             _f64(Gc), K, _i32(assign), _f32(dist))
    return assign, dist
