"""RaggedArray: rows of unequal length stored as one concatenated array.

The input/output container of the clustering surface: trajectories of unequal
length go in as a RaggedArray of per-trajectory coordinate blocks and
per-frame results come back partitioned the same way
(reference enspara/ra/ra.py:487-855 for the class, :223-242 partition_indices,
:361-376 partition_list, :27-43 where; behaviour pinned by the reference's
enspara/test/test_ra.py).  Pure numpy; ``save``/``load`` (reference
ra.py:45-89, :117-220) read and write the same one-dataset-per-row HDF5 files
through the package's own format implementation (``h5lite``), PyTables not
being part of this build.

Design differs from the reference: only ``_data`` (concatenated) and
``lengths`` are stored; rows are views into ``_data`` computed on demand from
the row offsets, so there is no second object array to keep in sync.
"""
import numbers
import operator

import numpy as np

from . import h5lite
from .exception import DataInvalid, ImproperlyConfigured


def _is_seq(x):
    return isinstance(x, (list, tuple, np.ndarray, RaggedArray)) or (
        hasattr(x, "__iter__") and not isinstance(x, (str, bytes)))


def partition_list(flat, lengths):
    """Split ``flat`` into consecutive pieces of the given lengths
    (reference ra.py:361-376).  DataInvalid if the lengths do not add up."""
    lengths = np.asarray(lengths, dtype=np.int64)
    total = int(lengths.sum()) if len(lengths) else 0
    if total != len(flat):
        raise DataInvalid(
            "Number of elements in list (%d) does not equal the sum of the "
            "lengths to partition (%d)" % (len(flat), total))
    out, start = [], 0
    for n in lengths:
        out.append(flat[start:start + int(n)])
        start += int(n)
    return out


def partition_indices(indices, traj_lengths):
    """Concatenated-frame indices -> (trajectory, frame) pairs
    (reference ra.py:223-242).  Indices past the end are dropped, as there."""
    lengths = np.asarray(traj_lengths, dtype=np.int64)
    ends = np.cumsum(lengths)
    out = []
    for index in indices:
        index = int(index)
        # first trajectory whose end lies beyond the index
        t = int(np.searchsorted(ends, index, side="right"))
        if t >= len(lengths):
            continue
        start = int(ends[t] - lengths[t])
        out.append((t, index - start))
    return out


def where(mask):
    """np.where for either a RaggedArray (-> (rows, columns)) or an ndarray
    (reference ra.py:27-43)."""
    if isinstance(mask, RaggedArray):
        flat = np.flatnonzero(mask._data)
        rows = np.searchsorted(mask._ends, flat, side="right")
        cols = flat - mask.starts[rows] if len(flat) else flat
        return (np.asarray(rows), np.asarray(cols))
    return np.where(mask)


def save(filename, array, compression_level=1, tag='arr'):
    """Write a RaggedArray (one dataset per row, ``<tag>_<zero-padded row>``) or
    an ndarray (``<tag>_0``) to an HDF5 file, chunked with shuffle + zlib at
    ``compression_level`` (reference ra.py:45-89)."""
    if hasattr(array, "lengths"):
        n_zeros = len(str(len(array.lengths))) + 1
        dtype = array._data.dtype
        rows = [np.asarray(array[i], dtype=dtype) for i in range(len(array))]
    else:
        n_zeros = 1
        rows = [np.asarray(array)]
    named = {}
    for i, row in enumerate(rows):
        named[tag + '_' + str(i).zfill(n_zeros)] = row
    h5lite.write(filename, named, compression_level=compression_level)
    return filename


def load(input_name, keys=..., stride=1):
    """Read a RaggedArray back (reference ra.py:117-220).  ``keys=...`` takes
    every dataset of the file as a row, in name order; a single key comes back
    as a plain ndarray; ``keys=None`` reads the old ``array``+``lengths`` (or
    lone ``arr_0``) layout.  ``stride`` keeps every stride-th entry of each
    row."""
    with h5lite.File(input_name) as handle:
        if keys is None:
            if 'lengths' in handle:
                a = RaggedArray(handle['array'].read(),
                                lengths=handle['lengths'].read())
                return a[::stride]
            return handle['arr_0'].read()[::stride]
        if keys is Ellipsis:
            keys = handle.keys()
        keys = [k.lstrip('/') for k in keys]
        try:
            nodes = [handle[k] for k in keys]
        except KeyError as e:
            raise KeyError("no dataset %s in %s" % (e, input_name))
        if len(nodes) == 1:
            return nodes[0].read()
        shapes = [node.shape for node in nodes]
        if not all(len(shapes[0]) == len(shape) for shape in shapes):
            raise DataInvalid(
                "The datasets to stack as rows differ in their number of "
                "dimensions: shapes %s" % shapes)
        for dim in range(1, len(shapes[0])):
            if not all(shapes[0][dim] == shape[dim] for shape in shapes):
                raise DataInvalid(
                    "The datasets to stack as rows must agree in every "
                    "dimension but the first; dimension %s differs: shapes %s"
                    % (dim, shapes))
        dtype = nodes[0].dtype
        if not all(dtype == node.dtype for node in nodes):
            raise DataInvalid(
                "The datasets %s do not all have dtype %s" % (keys, dtype))
        lengths = [(shape[0] + stride - 1) // stride for shape in shapes]
        concat = np.zeros((sum(lengths),) + tuple(shapes[0][1:]), dtype=dtype)
        start = 0
        for node, n in zip(nodes, lengths):
            concat[start:start + n] = node.read()[::stride]
            start += n
        return RaggedArray(array=concat, lengths=lengths, copy=False)


def zeros_like(array, *args, **kwargs):
    """reference ra.py:17-24"""
    if isinstance(array, RaggedArray):
        return RaggedArray(np.zeros_like(array._data), lengths=array.lengths)
    return np.zeros_like(array, *args, **kwargs)


class RaggedArray(object):
    """2-D-indexable view of rows with different lengths.

    ``RaggedArray([[1, 2], [3]])`` or ``RaggedArray(flat, lengths=[2, 1])``.
    """

    __slots__ = ("_data", "lengths", "_ends")

    def __init__(self, array, lengths=None, error_checking=True, copy=True):
        if lengths is None:
            if isinstance(array, RaggedArray):
                data = np.array(array._data, copy=copy)
                lengths = array.lengths.copy()
            elif len(array) == 0:
                data = np.array([])
                lengths = np.array([], dtype=np.int64)
            else:
                first_is_seq = _is_seq(array[0])
                if error_checking and len(array) <= 20000:
                    for row in array:
                        if _is_seq(row) != first_is_seq:
                            raise DataInvalid(
                                "The array elements in the input are not "
                                "consistent.")
                if first_is_seq:
                    rows = [np.asarray(r) for r in array]
                    lengths = np.array([len(r) for r in rows], dtype=np.int64)
                    nonempty = [r for r in rows if len(r)]
                    if nonempty:
                        data = np.concatenate(nonempty)
                    else:
                        data = np.array([])
                else:
                    data = np.array(array, copy=copy)
                    lengths = np.array([len(data)], dtype=np.int64)
        else:
            data = np.array(array, copy=copy) if copy else np.asarray(array)
            lengths = np.asarray(lengths, dtype=np.int64).copy()
            if int(lengths.sum()) != len(data):
                raise DataInvalid(
                    "Sum of lengths (%s) didn't match data shape (%s)." %
                    (int(lengths.sum()), data.shape))
        self._data = data
        self.lengths = lengths
        self._ends = np.cumsum(lengths)

    # ---- basic properties ----------------------------------------------------
    @property
    def starts(self):
        if len(self.lengths) == 0:
            return np.array([0])
        return np.append([0], self._ends[:-1])

    @property
    def dtype(self):
        return self._data.dtype

    @property
    def size(self):
        return self._data.size

    @property
    def shape(self):
        if len(self.lengths) and np.all(self.lengths == self.lengths[0]):
            second = int(self.lengths[0])
        else:
            second = None
        if self._data.ndim > 1:
            return (len(self.lengths), second) + tuple(self._data.shape[1:])
        return (len(self.lengths), second)

    def __len__(self):
        return len(self.lengths)

    def _row(self, i):
        n = len(self.lengths)
        if i < -n or i >= n:
            raise IndexError("row %d out of range for %d rows" % (i, n))
        i %= n
        end = int(self._ends[i])
        return self._data[end - int(self.lengths[i]):end]

    def __iter__(self):
        for i in range(len(self.lengths)):
            yield self._row(i)

    def _rows(self):
        return [self._row(i) for i in range(len(self.lengths))]

    def __repr__(self):
        return self._fmt("RaggedArray([\n", "      ", "])", repr)

    def __str__(self):
        return self._fmt("[", " ", "]", str)

    def _fmt(self, head, pad, tail, f):
        def line(r):
            return pad + f(r).split(")")[0].split("(")[-1]
        n = len(self)
        if n > 6:
            body = [line(self._row(i)) for i in (0, 1, 2)] + [pad + "..."] + \
                   [line(self._row(i)) for i in (-3, -2, -1)]
        else:
            body = [line(r) for r in self]
        return head + ",\n".join(body) + tail

    # ---- flat index helpers ----------------------------------------------------
    def _flat(self, rows, cols):
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        n = len(self.lengths)
        if np.any(rows < -n) or np.any(rows >= n):
            raise IndexError("row index out of range")
        rows = np.where(rows < 0, rows + n, rows)
        lens = self.lengths[rows]
        if np.any(cols < -lens) or np.any(cols >= lens):
            raise IndexError("column index out of range")
        cols = np.where(cols < 0, cols + lens, cols)
        return self.starts[rows] + cols

    def _row_list(self, sel):
        n = len(self.lengths)
        if isinstance(sel, slice):
            return list(range(*sel.indices(n)))
        if isinstance(sel, numbers.Integral):
            return [int(sel)]
        sel = np.asarray(sel)
        if sel.dtype == bool:
            return list(np.flatnonzero(sel))
        return [int(i) for i in sel]

    def _slice_cols(self, rows, sl):
        """flat indices and new lengths of rows[i][sl]"""
        flat, lens = [], []
        for r in rows:
            r = r % len(self.lengths) if r < 0 else r
            L = int(self.lengths[r])
            idx = np.arange(*sl.indices(L))
            flat.append(self.starts[r] + idx)
            lens.append(len(idx))
        flat = np.concatenate(flat) if flat else np.array([], dtype=np.int64)
        return flat.astype(np.int64), np.array(lens, dtype=np.int64)

    # ---- get / set ----------------------------------------------------------------
    def __getitem__(self, key):
        if isinstance(key, numbers.Integral):
            return self._row(int(key))
        if isinstance(key, RaggedArray):
            return self._data[key._data.astype(bool)]
        if isinstance(key, (slice, list, np.ndarray)):
            rows = self._row_list(key)
            return RaggedArray([self._row(i) for i in rows]) if rows else \
                RaggedArray([])
        if isinstance(key, tuple):
            if len(key) != 2:
                raise IndexError("RaggedArray takes at most 2 indices")
            r, c = key
            if isinstance(r, numbers.Integral):
                return self._row(int(r))[c]
            if isinstance(c, slice):
                flat, lens = self._slice_cols(self._row_list(r), c)
                return RaggedArray(self._data[flat], lengths=lens)
            if isinstance(r, slice):
                rows = self._row_list(r)
                cols = [int(c)] if isinstance(c, numbers.Integral) else list(c)
                rr = np.repeat(rows, len(cols))
                cc = np.tile(cols, len(rows))
                return RaggedArray(
                    self._data[self._flat(rr, cc)],
                    lengths=np.full(len(rows), len(cols), dtype=np.int64))
            return self._data[self._flat(r, c)]
        raise IndexError("unsupported index %r" % (key,))

    def __setitem__(self, key, value):
        if isinstance(value, RaggedArray):
            value = value._rows()
        if isinstance(key, RaggedArray):
            self._data[key._data.astype(bool)] = value
            return
        if isinstance(key, numbers.Integral):
            value = np.asarray(value)
            row = self._row(int(key))
            if value.shape[:1] == row.shape[:1] or value.ndim == 0:
                row[...] = value
            else:                                   # row changes length
                rows = self._rows()
                rows[int(key)] = value
                self.__init__(rows)
            return
        if isinstance(key, (slice, list, np.ndarray)):
            rows = self._rows()
            sel = self._row_list(key)
            if np.ndim(value) == 0:
                for i in sel:
                    rows[i][...] = value
            else:
                for i, v in zip(sel, value):
                    rows[i] = np.asarray(v)
                self.__init__(rows)
            return
        if isinstance(key, tuple):
            r, c = key
            if isinstance(r, numbers.Integral):
                self._row(int(r))[c] = value
                return
            if isinstance(c, slice):
                flat, _ = self._slice_cols(self._row_list(r), c)
            elif isinstance(r, slice):
                rows = self._row_list(r)
                cols = [int(c)] if isinstance(c, numbers.Integral) else list(c)
                flat = self._flat(np.repeat(rows, len(cols)),
                                  np.tile(cols, len(rows)))
            else:
                flat = self._flat(r, c)
            if _is_seq(value) and len(value) and _is_seq(value[0]):
                value = np.concatenate([np.asarray(v) for v in value])
            self._data[flat] = value
            return
        raise IndexError("unsupported index %r" % (key,))

    # ---- elementwise operators -------------------------------------------------------
    def map_operator(self, name, other):
        if isinstance(other, RaggedArray):
            other = other._data
        res = getattr(self._data, name)(other)
        if res is NotImplemented:
            return NotImplemented
        return RaggedArray(res, lengths=self.lengths, error_checking=False)

    def __invert__(self):
        return RaggedArray(~self._data, lengths=self.lengths)

    __hash__ = None

    # ---- reductions and growth -----------------------------------------------------------
    def all(self):
        return np.all(self._data)

    def any(self):
        return np.any(self._data)

    def max(self):
        return self._data.max()

    def min(self):
        return self._data.min()

    def flatten(self):
        return self._data.flatten()

    def append(self, values):
        if isinstance(values, RaggedArray):
            values = values._rows()
        if not _is_seq(values):
            raise DataInvalid("Expected an array of values or a ragged array")
        if len(self._data) == 0:
            self.__init__(values)
            return
        if len(values) and _is_seq(values[0]):
            new_rows = [np.asarray(v) for v in values]
        else:
            new_rows = [np.asarray(values)]
        self.__init__(self._rows() + new_rows)


def _install_operators():
    names = ["eq", "ne", "lt", "le", "gt", "ge", "add", "radd", "sub", "rsub",
             "mul", "rmul", "truediv", "rtruediv", "floordiv", "rfloordiv",
             "pow", "rpow", "mod", "rmod", "and", "or", "xor"]
    for n in names:
        dunder = "__%s__" % n

        def method(self, other, _d=dunder):
            return self.map_operator(_d, other)
        method.__name__ = dunder
        setattr(RaggedArray, dunder, method)


_install_operators()
del operator
