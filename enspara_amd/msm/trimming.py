"""Ergodic trimming of a count matrix (host-side graph work).

Behaviour follows the reference's enspara/msm/transition_matrices.py:
TrimMapping :26-110 and trim_disconnected :236-301.  The reference turns the
counts into a dense array first (:262-263); here the matrix stays sparse --
20 000 states are 3.2 GB dense -- and only the result is converted back to
the caller's type.  Component labels come from the same scipy routine on the
same edge set, so the kept states, their order and the mapping are identical.
"""
import csv

import numpy as np
import scipy.sparse
from scipy.sparse.csgraph import connected_components


class TrimMapping:
    """State ids before and after trimming.  ``to_original``: trimmed id ->
    original id; ``to_mapped`` is its inverse (assignable)."""

    __slots__ = ["to_original"]

    def __init__(self, transformations=None):
        # pairs are (original state id, trimmed state id)
        if transformations:
            self.to_original = {new: old for old, new in transformations}

    @property
    def to_mapped(self):
        return {old: new for new, old in self.to_original.items()}

    @to_mapped.setter
    def to_mapped(self, value):
        self.to_original = {new: old for old, new in value.items()}

    # CSV with the header "original,mapped", rows ordered by original id
    def write(self, file):
        w = csv.writer(file)
        w.writerow(["original", "mapped"])
        w.writerows(sorted(self.to_mapped.items(), key=lambda kv: kv[0]))

    def save(self, filename):
        with open(filename, "w") as fh:
            self.write(fh)

    @classmethod
    def read(cls, file):
        rows = csv.reader(file)
        header = next(rows)
        assert header == ["original", "mapped"]
        pairs = [(int(o), int(m)) for o, m in rows]
        return TrimMapping(pairs)

    @classmethod
    def load(cls, filename):
        with open(filename, "r") as fh:
            return cls.read(fh)

    def __eq__(self, other):
        if self is other:
            return True
        if hasattr(other, "to_original") and hasattr(other, "to_mapped"):
            return (self.to_original == other.to_original and
                    self.to_mapped == other.to_mapped)
        try:
            return TrimMapping(other) == self
        except Exception:
            return False

    def __str__(self):
        return "to_original:" + str(self.to_original)

    __repr__ = __str__


def trim_disconnected(counts, threshold=1, renumber_states=True):
    """Keep the strongly connected component with the most counts.

    An edge i->j exists where ``counts[i, j] >= threshold``; components are
    ranked by the sum of their states' row sums of the *unthresholded* counts
    (first one on ties).  With ``renumber_states`` the kept states are packed
    in ascending order into a smaller matrix; otherwise the rows and columns
    of the dropped states are zeroed.  Returns ``(TrimMapping, counts)`` with
    the counts in the caller's container type."""
    out_type = type(counts)
    dense_in = not scipy.sparse.issparse(counts)
    C = scipy.sparse.csr_matrix(np.asarray(counts) if dense_in else counts)
    C.sum_duplicates()

    edges = C.copy()
    edges.data = np.where(edges.data < threshold, 0, edges.data)
    edges.eliminate_zeros()
    n_sub, labels = connected_components(edges, connection="strong",
                                         directed=True)
    pops = np.asarray(C.sum(axis=1)).ravel()
    sub_pops = np.zeros(n_sub, dtype=pops.dtype)
    np.add.at(sub_pops, labels, pops)
    biggest = int(np.argmax(sub_pops))
    keep = np.flatnonzero(labels == biggest)

    if renumber_states:
        trimmed = C[keep][:, keep]
        mapping = TrimMapping(zip(keep.tolist(), range(len(keep))))
    else:
        mask = np.zeros(C.shape[0], dtype=C.dtype)
        mask[keep] = 1
        D = scipy.sparse.diags(mask, dtype=C.dtype)
        trimmed = (D @ C @ D).tocsr()
        trimmed.eliminate_zeros()
        mapping = TrimMapping(zip(keep.tolist(), keep.tolist()))
    trimmed = trimmed.astype(C.dtype)
    trimmed.sort_indices()

    if dense_in:
        dense = trimmed.toarray()
        return mapping, (dense if out_type is np.ndarray else out_type(dense))
    if type(trimmed) is not out_type:
        trimmed = out_type(trimmed)
    return mapping, trimmed
