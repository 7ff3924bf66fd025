"""euclidean / manhattan / hamming distance of one point to a set of points,
on the device.

Same call surface, validation errors and float64 output as the reference's
Cython module enspara/geometry/libdist.pyx (:148-203; checks :31-74).
``euclidean.bind(X)`` keeps the sample matrix resident on the GPU, so that
clustering loops which call ``metric(X, y)`` once per center upload X once.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..exception import DataInvalid

_KIND = {"float32": 0, "float64": 1, "int64": 2}


def _check(X, y, out):
    """reference libdist.pyx:31-74"""
    if len(X.shape) != 2:
        raise DataInvalid(
            "Data array dimension must be two, got shape %s." % str(X.shape))
    if len(y.shape) != 1:
        raise DataInvalid(
            "Target point dimension must be one, got shape %s." % str(y.shape))
    if X.shape[1] != y.shape[0]:
        raise DataInvalid(
            ("Target data point dimension (%s) must match data "
             "array dimension (%s)") % (y.shape[0], X.shape[1]))
    if out is None:
        return np.zeros((X.shape[0]), dtype=np.float64)
    if out.dtype != np.float64:
        raise DataInvalid(
            "In-place output array must be np.float64, got '%s'." % out.dtype)
    if out.shape[0] != X.shape[0]:
        raise DataInvalid(
            ("In-place output array dimension (%s) must match number of "
             "samples in data array (%s)") % (out.shape[0], X.shape[0]))
    if len(out.shape) != 1:
        raise DataInvalid(
            "In-place output array must be one-dimensional, got shape %s"
            % (out.shape,))
    return out


class _Resident:
    """One sample matrix on the device (ek_feat)."""

    def __init__(self, Xc, kind, device):
        self.L = _lib.load()
        self.kind = kind
        h = C.c_void_p()
        _lib.check(self.L.ek_feat_create(int(device), Xc.shape[0], Xc.shape[1],
                                         kind, C.byref(h)))
        self._h = h
        _lib.check(self.L.ek_feat_load(self._h, Xc.ctypes.data_as(C.c_void_p),
                                       0, Xc.shape[0]))

    def distance(self, metric, yc, out):
        _lib.check(self.L.ek_feat_distance(
            self._h, int(metric), yc.ctypes.data_as(C.c_void_p),
            _lib.f64p(out)))

    def kcenters(self, metric, first_label, max_new, cutoff, dist, assign):
        """ek_feat_kcenters on this matrix; dist (float64) / assign (int32) are
        updated in place.  -> (center samples int64 [k], distances.max())"""
        centers = np.empty(max(int(max_new), 1), dtype=np.int64)
        k = C.c_int32()
        fmax = C.c_double()
        _lib.check(self.L.ek_feat_kcenters(
            self._h, int(metric), int(first_label), int(max_new), float(cutoff),
            _lib.f64p(dist), _lib.i32p(assign), _lib.i64p(centers), C.byref(k),
            C.byref(fmax)))
        return centers[:k.value].copy(), fmax.value

    def pam_sweep(self, metric, medoids, proposals, raw, pos, dist, assign, accept,
                  cid):
        """ek_feat_pam_sweep on this matrix from cluster `cid` on; medoids
        (int64), dist (float64), assign (int32) and accept (int32) are updated
        in place.  -> (status, cid, pos)"""
        p = C.c_int64(int(pos))
        c = C.c_int32(int(cid))
        st = C.c_int32(0)
        props = None
        if proposals is not None:
            props = np.ascontiguousarray(proposals, dtype=np.int64)
        _lib.check(self.L.ek_feat_pam_sweep(
            self._h, int(metric), len(medoids), _lib.i64p(medoids),
            _lib.i64p(props) if props is not None else None,
            raw.ctypes.data_as(C.POINTER(C.c_uint32)), len(raw), C.byref(p),
            _lib.f64p(dist), _lib.i32p(assign), _lib.i32p(accept), C.byref(c),
            C.byref(st)))
        return st.value, c.value, p.value

    def __del__(self):
        try:
            if self._h:
                self.L.ek_feat_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _working_dtype(X, hamming):
    if hamming:
        if not np.issubdtype(X.dtype, np.integer):
            raise TypeError("hamming distance needs integer data, got %s"
                            % X.dtype)
        return np.int64
    if X.dtype == np.float32:
        return np.float32
    # integers and float64 are computed in float64 (exact for integer inputs
    # below 2**26, the range where the reference's integer arithmetic is too)
    return np.float64


class Bound:
    """``metric`` bound to one sample matrix that stays resident on the GPU:
    ``bound(X, y)`` costs one kernel and one read-back when ``X`` is the bound
    array, and falls back to a fresh upload for any other array.  The
    clustering loops bind their data once per fit (the caller must not modify
    the array in place while it is bound)."""

    def __init__(self, metric, X, device=0):
        self.metric = metric
        self.device_metric_id = metric      # (what the unbound callables carry)
        self.X = X
        self.device = device
        Xa = np.asarray(X)
        self.dt = _working_dtype(Xa, metric == 2)
        self.res = None
        if Xa.ndim == 2 and Xa.shape[0] > 0:
            self.res = _Resident(np.ascontiguousarray(Xa, dtype=self.dt),
                                 _KIND[np.dtype(self.dt).name], device)

    def bind(self, X, device=None):
        """The loops bind on entry; a metric already bound to this very array
        is handed on as it is (one upload per fit, not one per sweep)."""
        if X is self.X and (device is None or device == self.device):
            return self
        return Bound(self.metric, X, self.device if device is None else device)

    def __call__(self, X, y, out=None):
        if X is not self.X or self.res is None:
            return _run(self.metric, X, y, out, self.device)
        y = np.asarray(y)
        out = _check(np.asarray(X), y, out)
        self.res.distance(self.metric, np.ascontiguousarray(y, dtype=self.dt),
                          out)
        return out


def _run(metric, X, y, out, device):
    X = np.asarray(X) if not isinstance(X, np.ndarray) else X
    y = np.asarray(y)
    out = _check(X, y, out)
    if X.shape[0] == 0:
        return out
    dt = _working_dtype(X, metric == 2)
    res = _Resident(np.ascontiguousarray(X, dtype=dt),
                    _KIND[np.dtype(dt).name], device)
    res.distance(metric, np.ascontiguousarray(y, dtype=dt), out)
    return out


def euclidean(X, y, out=None, device=0):
    """sqrt(sum_j (X[i, j] - y[j])**2) for every row i -> float64 [n]
    (reference libdist.pyx:148-165)."""
    return _run(0, X, y, out, device)


euclidean.bind = lambda X, device=0: Bound(0, X, device)
euclidean.device_metric_id = 0


def manhattan(X, y, out=None, device=0):
    """sum_j |X[i, j] - y[j]| for every row i -> float64 [n]
    (reference libdist.pyx:167-184)."""
    return _run(1, X, y, out, device)


manhattan.bind = lambda X, device=0: Bound(1, X, device)
manhattan.device_metric_id = 1


def hamming(X, y, out=None, device=0):
    """fraction of features that differ, for every row i -> float64 [n]
    (reference libdist.pyx:187-203)."""
    return _run(2, X, y, out, device)


hamming.bind = lambda X, device=0: Bound(2, X, device)
hamming.device_metric_id = 2


def kcenters_resident(X, metric_id, first_label, max_new, cutoff, distances,
                      assignments, device=0):
    """The k-centers loop (reference kcenters.py:217-231, :243-311) for one of
    the metrics above with X, the float64 distances and the labels resident on
    the device for the whole run: no metric call, numpy pass or arg-max on the
    host per center.  ``distances`` (float64) and ``assignments`` (any integer
    dtype) are the state on entry; returns (center sample indices, distances,
    assignments, distances.max()), the arrays in the dtypes they came in."""
    Xa = np.asarray(X)
    dt = _working_dtype(Xa, metric_id == 2)
    res = _Resident(np.ascontiguousarray(Xa, dtype=dt),
                    _KIND[np.dtype(dt).name], device)
    d = np.ascontiguousarray(distances, dtype=np.float64).copy()
    a = np.ascontiguousarray(assignments, dtype=np.int32).copy()
    centers, fmax = res.kcenters(metric_id, first_label, max_new, cutoff, d, a)
    return centers, d, a.astype(np.asarray(assignments).dtype), fmax
