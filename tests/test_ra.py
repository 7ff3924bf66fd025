"""RaggedArray / partition helpers against the reference's behaviour
(tests/golden/ra_golden.npz from the real enspara.ra; cases modelled on
enspara/test/test_ra.py)."""
import os

import numpy as np
import pytest

from enspara_amd import ra
from enspara_amd.exception import DataInvalid


@pytest.fixture(scope="module")
def R(golden_dir):
    return np.load(os.path.join(golden_dir, "ra_golden.npz"))


ROWS = [np.arange(5), np.arange(3) + 10, np.arange(7) + 20, np.arange(2)]


def test_against_reference(R):
    a = ra.RaggedArray(ROWS)
    np.testing.assert_array_equal(a.lengths, R["rows_lengths"])
    np.testing.assert_array_equal(a._data, R["data"])
    np.testing.assert_array_equal(a.starts, R["starts"])
    np.testing.assert_array_equal(a[1:3].lengths, R["slice_1_3_lengths"])
    np.testing.assert_array_equal(a[1:3]._data, R["slice_1_3_data"])
    np.testing.assert_array_equal(a[:, 0:2]._data, R["cols_0_2_data"])
    np.testing.assert_array_equal(a[:, 0:2].lengths, R["cols_0_2_lengths"])
    np.testing.assert_array_equal(a[:, :-1]._data, R["cols_neg_data"])
    np.testing.assert_array_equal(a[:, :-1].lengths, R["cols_neg_lengths"])
    np.testing.assert_array_equal(a[:, 1:6:2]._data, R["cols_step_data"])
    np.testing.assert_array_equal(a[:, 1:6:2].lengths, R["cols_step_lengths"])
    np.testing.assert_array_equal(a[([0, 2, 2], [1, 0, 6])], R["fancy"])
    np.testing.assert_array_equal((a > 11)._data, R["gt_data"])
    w = ra.where(a > 11)
    np.testing.assert_array_equal(w[0], R["where_rows"])
    np.testing.assert_array_equal(w[1], R["where_cols"])
    np.testing.assert_array_equal(a[a > 11], R["mask_get"])
    np.testing.assert_array_equal((a + 1)._data, R["add_data"])
    got = ra.partition_indices([0, 4, 5, 7, 8, 15, 16], [5, 3, 7, 2])
    np.testing.assert_array_equal(np.array(got), R["partition_indices"])


def test_construct_and_index():
    a = ra.RaggedArray(ROWS)
    assert len(a) == 4 and a.size == 17 and a.shape == (4, None)
    np.testing.assert_array_equal(a[2], ROWS[2])
    assert a[2, 3] == 23 and a[-1, -1] == 1
    with pytest.raises(IndexError):
        a[4]
    with pytest.raises(IndexError):
        a[1, 3]
    b = ra.RaggedArray(np.arange(6), lengths=[3, 3])
    assert b.shape == (2, 3)
    with pytest.raises(DataInvalid):
        ra.RaggedArray(np.arange(6), lengths=[3, 4])
    with pytest.raises(DataInvalid):
        ra.RaggedArray([[1, 2], 3])
    e = ra.RaggedArray([])
    assert len(e) == 0
    # rows are views into the concatenated storage
    a[0][1] = 99
    assert a._data[1] == 99
    c = ra.RaggedArray(np.zeros((6, 4, 3)), lengths=[2, 4])
    assert c.shape == (2, None, 4, 3) and c[1].shape == (4, 4, 3)


def test_set_and_ops():
    a = ra.RaggedArray([[1, 2, 3], [4, 5]])
    a[0, 1] = 7
    assert a[0, 1] == 7
    a[([0, 1], [0, 1])] = [10, 11]
    np.testing.assert_array_equal(a._data, [10, 7, 3, 4, 11])
    a[a > 9] = 0
    np.testing.assert_array_equal(a._data, [0, 7, 3, 4, 0])
    a[:, 0] = 5
    np.testing.assert_array_equal(a._data, [5, 7, 3, 5, 0])
    b = a * 2 - a
    np.testing.assert_array_equal(b._data, a._data)
    assert (a == b).all() and not (a != b).any()
    assert a.max() == 7 and a.min() == 0
    with pytest.raises(TypeError):
        a + "x"
    a.append([[9, 9, 9, 9]])
    np.testing.assert_array_equal(a.lengths, [3, 2, 4])
    z = ra.zeros_like(a)
    assert z._data.sum() == 0 and list(z.lengths) == [3, 2, 4]
    np.testing.assert_array_equal((~(a > 4))._data, ~(a._data > 4))
    assert "RaggedArray" in repr(a) and str(a).startswith("[")


def test_partition_list():
    flat = np.arange(10)
    parts = ra.partition_list(flat, [3, 0, 7])
    assert [len(p) for p in parts] == [3, 0, 7]
    np.testing.assert_array_equal(parts[2], np.arange(3, 10))
    with pytest.raises(DataInvalid):
        ra.partition_list(flat, [3, 3])
