"""Transition counting and spectra of transition matrices on the device.

Surface follows the reference's enspara/msm/transition_matrices.py
(assigns_to_counts :113-170, eigenspectrum :173-233, eq_probs :304-307).
"""
import ctypes as C
import numbers

import numpy as np
import scipy.sparse

from .. import _lib
from ..exception import DataInvalid
from .trimming import TrimMapping, trim_disconnected  # noqa: F401  (reference module layout)


def _rows_of(assigns):
    """-> (concatenated int32 states, int64 lengths)"""
    if hasattr(assigns, "_data") and hasattr(assigns, "lengths"):
        flat = np.asarray(assigns._data)
        lengths = np.asarray(assigns.lengths, dtype=np.int64)
    else:
        if isinstance(assigns, np.ndarray) and assigns.dtype != object:
            if assigns.ndim == 1:
                raise DataInvalid(
                    'The given assignments array has 1-dimensional shape %s. '
                    'Two dimensional shapes = (n_trj, n_frames) are expected. '
                    'If this is really what you want, try using '
                    'assignments.reshape(1, -1) to create a single-row 2d '
                    'array.' % (assigns.shape,))
            flat = assigns.reshape(-1)
            lengths = np.full(assigns.shape[0], assigns.shape[1],
                              dtype=np.int64)
        else:
            rows = [np.asarray(r) for r in assigns]
            lengths = np.array([len(r) for r in rows], dtype=np.int64)
            flat = np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)
    if flat.size and not np.issubdtype(flat.dtype, np.integer):
        raise DataInvalid("assignments must be integers, got %s" % flat.dtype)
    return flat, lengths


def assigns_to_counts(assigns, lag_time, max_n_states=None,
                      sliding_window=True, device=0, lengths=None):
    """Count transitions (reference transition_matrices.py:113-170).

    Returns a scipy.sparse.coo_matrix of int counts, shape
    (max_n_states, max_n_states).  Unlike the reference's COO (one entry of 1
    per observed transition, summed only on conversion) duplicates are already
    summed and entries are sorted by (row, col); every derived quantity
    (todense, tocsr, arithmetic) is identical.

    ``assigns`` may also be a :class:`enspara_amd.device.FrameStore` whose
    labels are resident on the GPU after a fit -- the reference pipes
    ``result.assignments`` straight into this function; here they need not
    leave the device in between.  Then ``lengths`` (frames per trajectory, in
    the store's frame order) and ``max_n_states`` are required.
    """
    if not isinstance(lag_time, numbers.Integral):
        raise DataInvalid("The lag time must be an integer. Got %s type %s."
                          % (lag_time, type(lag_time)))
    if lag_time < 1:
        raise DataInvalid("Lag times must be be strictly greater than 0. "
                          "Got '%s'." % lag_time)
    if hasattr(assigns, "msm_counts"):
        if lengths is None or max_n_states is None:
            raise DataInvalid("counting over a FrameStore's resident labels "
                              "needs lengths= and max_n_states=")
        if sum(int(v) for v in lengths) != assigns.n:
            raise DataInvalid("lengths add up to %d frames, the store holds %d"
                              % (sum(int(v) for v in lengths), assigns.n))
        max_n_states = int(max_n_states)
        if max_n_states == 0 or assigns.n == 0:
            return scipy.sparse.coo_matrix((max_n_states, max_n_states),
                                           dtype=int)
        r, c, v = assigns.msm_counts(lengths, lag_time, max_n_states,
                                     sliding_window)
        return scipy.sparse.coo_matrix((v.astype(int), (r, c)),
                                       shape=(max_n_states, max_n_states))
    flat, lengths = _rows_of(assigns)
    if max_n_states is None:
        kept = flat[flat != -1]
        max_n_states = int(kept.max()) + 1 if kept.size else 0
    max_n_states = int(max_n_states)
    if flat.size:
        lo, hi = int(flat.min()), int(flat.max())
        if lo < -1:
            raise ValueError("negative row index found")   # scipy's wording
        if hi >= max_n_states:
            raise ValueError("row index exceeds matrix dimensions")
    if max_n_states == 0 or flat.size == 0:
        return scipy.sparse.coo_matrix((max_n_states, max_n_states), dtype=int)
    a32 = np.ascontiguousarray(flat, dtype=np.int32)
    cap = max(1, int(a32.size))
    rows = np.empty(cap, dtype=np.int32)
    cols = np.empty(cap, dtype=np.int32)
    vals = np.empty(cap, dtype=np.int64)
    nnz = C.c_int64()
    L = _lib.load()
    _lib.check(L.ek_msm_counts(
        int(device), _lib.i32p(a32), _lib.i64p(lengths), len(lengths),
        int(lag_time), 1 if sliding_window else 0, max_n_states, cap,
        _lib.i32p(rows), _lib.i32p(cols), _lib.i64p(vals), C.byref(nnz)))
    k = nnz.value
    return scipy.sparse.coo_matrix(
        (vals[:k].astype(int), (rows[:k], cols[:k])),
        shape=(max_n_states, max_n_states))


# ---------------------------------------------------------------------------
# leading eigenpairs: Arnoldi / Krylov-Schur with the basis on the device
# ---------------------------------------------------------------------------
class DeviceKrylov:
    """Krylov basis + sparse operator resident on one GPU (csrc/ek_krylov.hip).
    The solver below talks to this small protocol only."""

    def __init__(self, A_csr, m_max, device=0):
        A = scipy.sparse.csr_matrix(A_csr).astype(np.float64)
        A.sort_indices()
        self.n = A.shape[0]
        self.m_max = int(m_max)
        self.L = _lib.load()
        indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(A.indices, dtype=np.int32)
        data = np.ascontiguousarray(A.data, dtype=np.float64)
        h = C.c_void_p()
        _lib.check(self.L.ek_krylov_create(
            int(device), self.n, _lib.i64p(indptr), _lib.i32p(indices),
            _lib.f64p(data), self.m_max, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.L.ek_krylov_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_vector(self, j, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        _lib.check(self.L.ek_krylov_set_vector(self._h, int(j), _lib.f64p(v)))

    def get_vector(self, j):
        v = np.empty(self.n, dtype=np.float64)
        _lib.check(self.L.ek_krylov_get_vector(self._h, int(j), _lib.f64p(v)))
        return v

    def set_filter(self, degree, a=0.0, b=1.0):
        """the operator of step() / expand(): A (degree 0) or the Chebyshev
        polynomial of that degree on [a, b] (ek_krylov_set_filter)"""
        _lib.check(self.L.ek_krylov_set_filter(self._h, int(degree), float(a),
                                               float(b)))

    def step(self, j, apply=True):
        h = np.empty(j + 2, dtype=np.float64)
        _lib.check(self.L.ek_krylov_step(self._h, int(j), 1 if apply else 0,
                                         _lib.f64p(h)))
        return h

    def expand(self, j0, m):
        """Steps j0..m-1 in one call -> Hessenberg columns [m+1, m] (only
        columns j0.. are filled)."""
        H = np.zeros((m + 1, m), dtype=np.float64, order="F")
        _lib.check(self.L.ek_krylov_expand(
            self._h, int(j0), int(m),
            H.ctypes.data_as(C.POINTER(C.c_double)), m + 1))
        return H

    def rotate(self, m, Q, move_last):
        Q = np.asfortranarray(Q, dtype=np.float64)
        kk = Q.shape[1]
        _lib.check(self.L.ek_krylov_rotate(
            self._h, int(m), int(kk),
            Q.ctypes.data_as(C.POINTER(C.c_double)), 1 if move_last else 0))


    def combine(self, m, Q):
        """-> [kk, n] rows = V[:m].T @ Q[:, c]; the basis is not modified"""
        Q = np.asfortranarray(Q, dtype=np.float64)
        kk = Q.shape[1]
        out = np.empty((kk, self.n), dtype=np.float64)
        step = self.m_max + 1
        for c0 in range(0, kk, step):
            c1 = min(kk, c0 + step)
            Qc = np.asfortranarray(Q[:, c0:c1])
            _lib.check(self.L.ek_krylov_combine(
                self._h, int(m), c1 - c0,
                Qc.ctypes.data_as(C.POINTER(C.c_double)),
                out[c0:c1].ctypes.data_as(C.POINTER(C.c_double))))
        return out


def _start_vector(n, seed=0):
    v = np.random.RandomState(seed).uniform(0.5, 1.5, size=n)
    return v / np.linalg.norm(v)


def _expand(space, H, j0, m, rng_seed=1):
    """Arnoldi steps j0..m-1.  Returns the effective m (smaller only if the
    basis became complete)."""
    n = space.n
    if hasattr(space, "expand") and m > j0:
        # all steps enqueued at once; fall back to single steps from the
        # first breakdown (tiny sub-diagonal), if any
        Hn = space.expand(j0, m)
        j_bad = None
        for j in range(j0, m):
            h = Hn[:j + 2, j]
            scale = np.linalg.norm(h[:j + 1]) + h[j + 1]
            if not np.all(np.isfinite(h)) or \
                    h[j + 1] <= 1e-13 * max(scale, 1e-300):
                j_bad = j
                break
            H[:j + 2, j] = h
        if j_bad is None:
            return m
        j0 = j_bad
    for j in range(j0, m):
        h = space.step(j, True)
        H[:j + 2, j] = h
        scale = np.linalg.norm(h[:j + 1]) + h[j + 1]
        if h[j + 1] <= 1e-13 * max(scale, 1e-300):
            # invariant subspace: continue with a fresh direction (the
            # projected matrix becomes block triangular, H[j+1, j] = 0)
            H[j + 1, j] = 0.0
            if j + 1 >= n:
                return j + 1
            for attempt in range(5):
                space.set_vector(j + 1, _start_vector(n, rng_seed + 17 * j
                                                      + attempt))
                hh = space.step(j, False)
                if hh[j + 1] > 1e-8:
                    break
            else:
                return j + 1
    return m


# The restarted iteration on a polynomial of the matrix (round 5).  1: once the first
# cycle has shown where the wanted eigenvalues end, the steps apply the Chebyshev
# polynomial that keeps everything below that within [-1, 1] and pulls what is above
# apart; 0: the plain iteration (round 4's).  Same eigenpairs: they are the
# MATRIX's, by Rayleigh-Ritz on the converged subspace, residuals checked.
FILTER = 1
LAST_RUN = {}           # (diagnostics of the last _leading_eigs call)


def _cheb(z, d):
    return np.cosh(d * np.arccosh(np.asarray(z, dtype=complex)))


def _degree_for(top, a, b):
    """Chebyshev degree on [a, b] that amplifies `top` about thirtyfold (cosh 4)"""
    w1 = (top - 0.5 * (a + b)) / (0.5 * (b - a))
    if not (w1 > 1.0 + 1e-10):
        return None
    return int(np.clip(np.ceil(4.0 / np.arccosh(w1)), 4, 64))


def _plan_filter(theta, k, scout=None):
    """theta: estimates of the leading eigenvalues (the Ritz values of a cycle on the
    matrix, or _unfilter's of a cycle on a polynomial).  -> (degree, a, b, kf):
    Chebyshev degree, damped interval, how many pairs to converge; or None.
    scout = (n, m): the first plan, from the Ritz values of ONE cycle on the matrix.
    Where the leading eigenvalues cluster those place the end of the wanted ones far
    too low (the 26th of 60 at 0.54 for eigenvalues within 0.003 of 1), but they
    sample the spectrum the way Chebyshev nodes do: the j-th largest sits near the
    n sin^2(pi j / 2m)-th eigenvalue.  The interval ends at the one that stands for
    somewhat more eigenvalues than a restart keeps; if that was too high -- fewer
    pairs above it than wanted -- the cycle on the polynomial shows it and the
    interval is lowered (_leading_eigs)."""
    th = theta[np.argsort(-theta.real, kind="stable")]
    kf = min(len(th) - 2, k + max(4, k // 4))
    if kf < k or kf < 1:
        return None
    top = float(th[0].real)
    lo = float(th.real.min())
    a = min(-1.0, lo - 0.05 * (top - lo))
    keep = min(len(th) - 2, kf + (len(th) - kf) // 2)   # what a restart keeps
    if scout is not None:
        n, m = scout
        j = int(np.ceil(2.0 * m / np.pi * np.arcsin(np.sqrt(min(1.0, 1.3 * keep / n))))) + 1
        j = int(np.clip(j, 2, kf))
        b = float(th[j].real)
        d = _degree_for(top, a, b) if top > b > a else None
        if not d:
            return None
        # Eigenvalues off the real axis grow under the polynomial with their distance
        # from [a, b] in ANY direction: a pair at 0.1 +- 0.7 i would pass the real ones
        # at 0.67.  The cycle's Ritz values show the periphery of the spectrum; none
        # below the interval's end may come out of the polynomial beyond the band the
        # real ones there are confined to (else: the plain iteration)
        pv = _cheb((th - 0.5 * (a + b)) / (0.5 * (b - a)), d)
        below = th.real <= b
        if below.any() and np.abs(pv[below]).max() > 1.5:
            return None
        return d, a, b, kf
    # b: the highest of the estimates below what a restart keeps (each is, at worst, a
    # lower bound of its eigenvalue: the wanted ones stay outside the interval, and so
    # does what they converge against) for which a polynomial of reasonable degree
    # lifts ALL the pairs to converge well clear of [-1, 1], where everything else ends
    # up, and keeps their order by real part (eigenvalues off the real axis turn with
    # the degree)
    for idx in range(keep, len(th) - 1):
        b = float(th[idx].real)
        d = _degree_for(top, a, b) if top > b > a else None
        if not d:
            continue
        c, e = 0.5 * (a + b), 0.5 * (b - a)
        pv = _cheb((th[:kf] - c) / e, d).real
        if np.all(np.diff(pv) <= 1e-6 * abs(pv[0])) and pv[kf - 1] >= 4.0:
            return d, a, b, kf
    return None


def _unfilter(theta_b, flt):
    """Estimates of the matrix's eigenvalues from the Ritz values of its polynomial:
    the Chebyshev map inverted where it is one to one (beyond 1); what lies in the
    damped band is only known to be below b."""
    d, a, b = flt
    c, e = 0.5 * (a + b), 0.5 * (b - a)
    z = np.asarray(theta_b, dtype=complex)
    out = np.full(z.shape, b, dtype=complex)
    up = z.real > 1.0
    out[up] = c + e * np.cosh(np.arccosh(z[up]) / d)
    return out


def _rayleigh_ritz(A, X):
    """Eigenpairs of A restricted to span(X) (complex [n, q], an approximately
    invariant subspace).  -> (values, vectors [n, r], relative residuals)"""
    import scipy.linalg
    Z = np.concatenate([X.real, X.imag], axis=1)
    U, sv, _ = np.linalg.svd(Z, full_matrices=False)
    r = int((sv > 1e-9 * sv[0]).sum())
    Q = U[:, :r]
    AQ = A @ Q
    lam, W = scipy.linalg.eig(Q.T @ AQ)
    Xn = Q @ W
    res = np.linalg.norm(AQ @ W - Xn * lam, axis=0) / np.linalg.norm(Xn, axis=0)
    return lam, Xn, res


def _leading_eigs(space, k, tol=1e-12, max_restarts=500, A=None):
    """k eigenpairs of largest real part.  -> (vals complex [k], vecs [n, k]).
    A: the matrix itself (scipy sparse), for the filtered iteration's Rayleigh-Ritz;
    None: the plain iteration only."""
    import scipy.linalg
    n = space.n
    m = space.m_max
    full = m >= n
    if full:
        m = n
    H = np.zeros((m + 1, m))
    space.set_vector(0, _start_vector(n))
    j0 = 0
    can_filter = bool(FILTER) and A is not None and not full and hasattr(space, "set_filter")
    phase = 0               # 0: on the matrix; 1: on the polynomial; 2: on the matrix again
    k_want = k
    flt = None              # (degree, a, b) of the polynomial in use
    stats = {"restarts": [0, 0, 0], "filter": None, "fallback": False, "plans": 0}
    LAST_RUN.clear()
    LAST_RUN.update(stats)
    restart = 0
    while restart < max_restarts:
        restart += 1
        LAST_RUN["restarts"][phase] += 1
        m_eff = _expand(space, H, j0, m)
        Hm = H[:m_eff, :m_eff]
        b = H[m_eff, :m_eff].copy() if m_eff < H.shape[0] else np.zeros(m_eff)
        if full or m_eff < m:
            break                      # the basis spans an invariant subspace
        # One real Schur form of the projected matrix (LAPACK dgees, no Python
        # callback), its eigenvalues from the same call, the wanted ones moved
        # to the top by dtrsen: the three decompositions per restart this loop
        # used to make (eigvals, schur with a sort callable, eig) were 2 ms of
        # host time per restart next to ~0.5 ms of device work.
        S, _, wr, wi, Z, _, info = scipy.linalg.lapack.dgees(
            lambda re, im: False, np.asfortranarray(Hm), sort_t=0)
        if info != 0:
            raise np.linalg.LinAlgError("dgees failed (info %d)" % info)
        order = np.argsort(-wr, kind="stable")
        # keep k wanted + some extra, never splitting a conjugate pair (the two
        # rows of a 2 x 2 block are neighbours: wi > 0 then wi < 0)
        p = min(m - 1, k_want + max(1, (m - k_want) // 2))
        select = np.zeros(m_eff, dtype=np.int32)
        select[order[:p]] = 1
        for i in np.flatnonzero(select):
            if wi[i] > 0 and i + 1 < m_eff:
                select[i + 1] = 1
            elif wi[i] < 0 and i > 0:
                select[i - 1] = 1
        if select.sum() >= m:           # (a pair pushed it to the full basis)
            drop = order[p - 1]
            select[drop] = 0
            if wi[drop] > 0:
                select[drop + 1] = 0
            elif wi[drop] < 0:
                select[drop - 1] = 0
        out = scipy.linalg.lapack.dtrsen(select, S, Z, job="N", wantq=1)
        S, Z, p, info = out[0], out[1], int(out[4]), out[-1]
        if info not in (0, 1):
            raise np.linalg.LinAlgError("dtrsen failed (info %d, %d selected)"
                                        % (info, p))
        if info == 1 or p < 1 or p >= m:
            # info 1: eigenvalues too close to swap (the leading eigenvalues of
            # a metastable chain cluster at 1) -- S, Z are still a Schur form of
            # the projected matrix, only partly reordered.  Go on with what it
            # kept (or the plain truncation after k), not splitting a 2 x 2
            # block: a restart from a Schur form that is not perfectly sorted
            # converges a little later, it does not fail.
            p = p if 1 <= p < m else max(1, min(m - 1, k_want))
            if p < m_eff and S[p, p - 1] != 0.0:
                p = p + 1 if p + 1 < m else p - 1
        bz = b @ Z
        # residuals of the wanted Ritz pairs
        sv, sy = scipy.linalg.eig(S[:p, :p])
        o2 = np.argsort(-sv.real, kind="stable")[:k_want]
        res = np.abs(bz[:p] @ sy[:, o2])
        converged = np.all(res <= tol * np.maximum(np.abs(sv[o2]), 1e-3))
        if converged and phase != 1:
            break
        if (phase == 0 and can_filter) or (phase == 1 and not converged and
                                           LAST_RUN["plans"] < 4):
            # ---- a cycle is through: go on with a (better) polynomial of the matrix? ----
            plan = None
            if phase == 0:
                plan = _plan_filter(wr + 1j * wi, k, scout=(n, m))
                phase = 2
            else:
                top = float(_unfilter(np.array([wr.max()]), flt)[0].real)
                if int((wr > 1.5).sum()) < k_want + 2:
                    # fewer pairs above the interval than are to converge: it ends too
                    # high -- four times as far below the largest eigenvalue
                    fb = top - 4.0 * (top - flt[2])
                    d = _degree_for(top, flt[1], fb) if fb > flt[1] else None
                    plan = (d, flt[1], fb, k_want) if d else None
                    if plan is None:
                        # (no interval left to try: the plain iteration from here)
                        LAST_RUN["fallback"] = True
                        y = (Z[:, :p] @ sy[:, o2]).real.sum(axis=1)
                        x0 = space.combine(m_eff, y.reshape(-1, 1))[0]
                        space.set_filter(0)
                        space.set_vector(0, x0 / np.linalg.norm(x0))
                        H[:] = 0.0
                        j0 = 0
                        k_want = k
                        phase = 2
                        continue
                else:
                    # every change of polynomial starts the basis over from one vector:
                    # only for an interval that leaves less than half as much above it
                    # (and only away from a weak one: past a degree of 16 what a higher
                    # interval gains in restarts it pays in products, measured)
                    plan = _plan_filter(_unfilter(wr + 1j * wi, flt), k) if flt[0] < 16 \
                        else None
                    if plan is not None and not (top - plan[2] < 0.5 * (top - flt[2])):
                        plan = None
            if plan is not None:
                d, fa, fb, kf = plan
                LAST_RUN["plans"] += 1
                flt = (d, fa, fb)
                LAST_RUN["filter"] = {"degree": d, "a": fa, "b": fb, "pairs": kf}
                # a start vector rich in the wanted directions: the sum of the
                # leading Ritz vectors (x = V Z y)
                y = (Z[:, :p] @ sy[:, np.argsort(-sv.real, kind="stable")[:kf]]).real.sum(axis=1)
                x0 = space.combine(m_eff, y.reshape(-1, 1))[0]
                nx = np.linalg.norm(x0)
                if nx > 0 and np.all(np.isfinite(x0)):
                    space.set_filter(d, fa, fb)
                    space.set_vector(0, x0 / nx)
                    H[:] = 0.0
                    j0 = 0
                    k_want = kf
                    phase = 1
                    continue
        if phase == 1 and converged:
            # ---- the polynomial's leading invariant subspace: the matrix's pairs in it ---
            Y = Z[:, :p] @ sy[:, o2]
            X = space.combine(m_eff, np.concatenate([Y.real, Y.imag], axis=1))
            kk = Y.shape[1]
            lam, Xn, rr = _rayleigh_ritz(A, (X[:kk] + 1j * X[kk:]).T)
            o3 = np.argsort(-lam.real, kind="stable")[:k]
            # (every pair converged on lies clear of the band everything else is damped
            # into: they ARE the polynomial's leading ones, hence the matrix's)
            # ... and every one of the matrix's pairs found lies beyond the interval: none
            # came in from off the real axis)
            clear = sv[o2].real.min() > 1.5 and \
                not np.any((rr <= 1e-6) & (lam.real <= flt[2]))
            if clear and len(o3) == k and \
                    np.all(rr[o3] <= 1e-9 * np.maximum(np.abs(lam[o3]), 1e-3)):
                space.set_filter(0)
                return lam[o3], Xn[:, o3]
            # not the matrix's pairs to the accuracy asked: the plain iteration from
            # here (it converges from whatever it is given)
            LAST_RUN["fallback"] = True
            space.set_filter(0)
            x0 = Xn[:, o3].real.sum(axis=1)
            space.set_vector(0, x0 / np.linalg.norm(x0))
            H[:] = 0.0
            j0 = 0
            k_want = k
            phase = 2
            continue
        # restart: V <- V Z[:, :p], residual direction moves to slot p
        space.rotate(m, Z[:, :p], True)
        H[:] = 0.0
        H[:p, :p] = S[:p, :p]
        H[p, :p] = bz[:p]
        j0 = p
    if phase == 1:          # (the restarts ran out on the polynomial)
        space.set_filter(0)
        raise np.linalg.LinAlgError("eigenspectrum: no convergence in %d restarts"
                                    % max_restarts)
    # Ritz pairs of the final projected matrix
    Hm = H[:m_eff, :m_eff]
    vals, Y = scipy.linalg.eig(Hm)
    order = np.argsort(-vals.real, kind="stable")[:k]
    vals = vals[order]
    Y = Y[:, order]
    # x = V y: real and imaginary parts through the device rotation
    kk = len(order)
    X = space.combine(m_eff, np.concatenate([Y.real, Y.imag], axis=1))
    vecs = (X[:kk] + 1j * X[kk:]).T
    return vals, vecs


def eigenspectrum(T, n_eigs=None, left=True, maxiter=100000, tol=1E-30,
                  device=0, _space_factory=None):
    """Eigenvalues / eigenvectors of a transition probability matrix, sorted by
    descending real part, first vector scaled to sum 1, real parts returned
    (reference transition_matrices.py:173-233).  Matrices with up to 1000
    states get a complete Arnoldi basis (all eigenpairs, like the reference's
    dense branch); larger ones a thick-restarted Krylov-Schur iteration for
    the ``n_eigs`` of largest real part (the reference's ARPACK ``which='LR'``
    branch).  SpMV, orthogonalisation and basis rotations run on the GPU."""
    n = T.shape[0]
    if n_eigs is None:
        n_eigs = n
    elif n_eigs < 2:
        raise ValueError('n_eig must be greater than or equal to 2')
    n_eigs = min(int(n_eigs), n)
    A = scipy.sparse.csr_matrix(T.T if left else T)
    if n <= 1000:
        m_max = n
        k = n
    else:
        k = n_eigs
        # (a basis of 3 k: a third of the restarts 2 k + 1 needs on matrices whose
        # leading eigenvalues cluster at 1, for 40 % fewer Arnoldi steps)
        m_max = min(n - 1, max(3 * k, 60))
    make = _space_factory or (lambda A_, m_: DeviceKrylov(A_, m_, device))
    space = make(A, m_max)
    try:
        vals, vecs = _leading_eigs(space, k, tol=max(tol, 1e-13),
                                   max_restarts=max(10, min(maxiter, 2000)), A=A)
    finally:
        if hasattr(space, "close"):
            space.close()
    order = np.argsort(-np.real(vals), kind="stable")
    vals = vals[order]
    vecs = vecs[:, order]
    vecs[:, 0] /= vecs[:, 0].sum()
    return np.real(vals[:n_eigs]), np.real(vecs[:, :n_eigs])


def eq_probs(T, maxiter=100000, tol=1E-30, device=0):
    """reference transition_matrices.py:304-307"""
    val, vec = eigenspectrum(T, n_eigs=3, left=True, maxiter=maxiter, tol=tol,
                             device=device)
    return vec[:, 0]
