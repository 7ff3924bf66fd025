"""Counts -> transition probabilities (reference enspara/msm/builders.py:
normalize :123-155, transpose :83-120, _row_normalize :171-204)."""
import numpy as np
import scipy.sparse

from .. import _lib


def _apply_prior_counts(C, prior_counts):
    """reference builders.py:158-168"""
    if prior_counts is not None:
        try:
            C = C + prior_counts
        except NotImplementedError:
            C = np.array(C.todense()) + prior_counts
    return C


def _row_normalize(C, device=0):
    """Row-normalise on the device; zero rows stay zero
    (reference builders.py:171-204).  Sparse in -> same sparse type out."""
    sparse_in = scipy.sparse.issparse(C)
    csr = scipy.sparse.csr_matrix(C).astype(np.float64)
    n = csr.shape[0]
    indptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)
    data = np.ascontiguousarray(csr.data, dtype=np.float64)
    out = np.empty_like(data)
    L = _lib.load()
    _lib.check(L.ek_msm_row_normalize(int(device), _lib.i64p(indptr),
                                      _lib.f64p(data), n, _lib.f64p(out),
                                      None))
    T = scipy.sparse.csr_matrix((out, csr.indices, csr.indptr),
                                shape=csr.shape)
    if sparse_in:
        return type(C)(T)
    return np.asarray(T.todense())


def normalize(C, prior_counts=None, calculate_eq_probs=True, device=0):
    """reference builders.py:123-155"""
    from .transition_matrices import eq_probs
    C = _apply_prior_counts(C, prior_counts)
    probs = _row_normalize(C, device=device)
    equilibrium = None
    if calculate_eq_probs:
        equilibrium = eq_probs(probs, device=device)
    return C, probs, equilibrium


def transpose(C, prior_counts=None, calculate_eq_probs=True, device=0):
    """reference builders.py:83-120"""
    C = _apply_prior_counts(C, prior_counts)
    C_sym = C + C.T
    probs = _row_normalize(C_sym, device=device)
    if type(C) is not type(probs):
        probs = type(C)(probs)
        C_sym = type(C)(C_sym)
    equilibrium = None
    if calculate_eq_probs:
        equilibrium = np.array(C_sym.sum(axis=1) / C_sym.sum()).flatten()
    return C_sym / 2, probs, equilibrium
