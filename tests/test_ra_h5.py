"""HDF5 save/load of RaggedArrays (reference
enspara/ra/ra.py:45-89 save, :117-220 load; behaviour pinned by the reference's
enspara/test/test_ra.py:63-124).

Three independent anchors for the format implementation in
enspara_amd/h5lite.py:
 * tests/golden/frame0.h5 is the reference's own PyTables-written data file
   (enspara/test/data/frame0.h5): the reader must reproduce the trajectory that
   the XTC fixture of the same frames holds;
 * libhdf5's h5dump must read every file written here and return the same
   bytes;
 * files rewritten by libhdf5's h5repack (other layouts and filter chains) must
   read back identically.
The libhdf5 tools are looked up at run time and those tests skip without them.
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from enspara_amd import h5lite, ra
from enspara_amd.exception import DataInvalid

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, os.path.dirname(HERE))


def _tool(name):
    found = shutil.which(name)
    if found is None and os.path.exists(os.path.join("/opt/conda/bin", name)):
        found = os.path.join("/opt/conda/bin", name)
    return found


H5DUMP, H5REPACK = _tool("h5dump"), _tool("h5repack")
needs_tools = pytest.mark.skipif(H5DUMP is None or H5REPACK is None,
                                 reason="libhdf5 command-line tools not installed")


def _dump(path, name, dtype, shape, tmp_path):
    out = str(tmp_path / "dump.bin")
    r = subprocess.run([H5DUMP, "-d", "/" + name, "-b", "FILE", "-o", out, path],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    return raw.view(dtype).reshape(shape)


# ---------------------------------------------------------------------------
# reader against the reference's PyTables-written fixture
# ---------------------------------------------------------------------------
def test_reads_pytables_fixture():
    from oracle import xtc
    with h5lite.File(os.path.join(GOLDEN, "frame0.h5")) as f:
        assert f.keys() == ["cell_angles", "cell_lengths", "coordinates",
                            "time", "topology"]
        node = f["coordinates"]
        assert node.shape == (501, 22, 3) and node.dtype == np.float32
        assert [fid for fid, _ in node._filters] == [2, 1]   # shuffle, deflate
        xyz = node.read()
        time = f["time"].read()
        topology = json.loads(f["topology"].read()[0].decode("ascii"))
        angles = f["cell_angles"].read()
    same = xtc.read_xtc(os.path.join(GOLDEN, "frame0.xtc"))
    # the two copies of these frames were rounded separately: they agree to
    # 1e-3 nm, no closer
    np.testing.assert_allclose(xyz, same["xyz"], atol=1.01e-3, rtol=0)
    np.testing.assert_allclose(time, 500 + np.arange(501), rtol=1e-6)
    atoms = sum(len(r["atoms"]) for c in topology["chains"] for r in c["residues"])
    assert atoms == 22
    assert angles.shape == (501, 3)


@needs_tools
def test_fixture_matches_libhdf5_exactly(tmp_path):
    path = os.path.join(GOLDEN, "frame0.h5")
    with h5lite.File(path) as f:
        for key in f.keys():
            node = f[key]
            got = _dump(path, key, node.dtype, node.shape, tmp_path)
            np.testing.assert_array_equal(node.read(), got, err_msg=key)


# ---------------------------------------------------------------------------
# reference test_ra.py:63-124
# ---------------------------------------------------------------------------
def _ragged(dtype=np.float64, inner=()):
    rng = np.random.default_rng(3)
    return ra.RaggedArray([(rng.random((n,) + inner) * 50).astype(dtype)
                           for n in (10, 5, 0, 17, 1)])


def test_save_load_ragged(tmp_path):
    a = _ragged()
    name = str(tmp_path / "a.h5")
    assert ra.save(name, a) == name
    b = ra.load(name)
    assert isinstance(b, ra.RaggedArray)
    np.testing.assert_array_equal(b.lengths, a.lengths)
    np.testing.assert_array_equal(b._data, a._data)
    assert b.dtype == a.dtype
    with h5lite.File(name) as f:      # row keys: tag + zero-padded row number
        assert f.keys() == ["arr_00", "arr_01", "arr_02", "arr_03", "arr_04"]


def test_save_load_strided(tmp_path):
    a = _ragged(np.float32, (4, 3))
    name = str(tmp_path / "a.h5")
    ra.save(name, a)
    b = ra.load(name, stride=3)
    np.testing.assert_array_equal(b.lengths, [(n + 2) // 3 for n in a.lengths])
    for i in range(len(a)):
        np.testing.assert_array_equal(b[i], a[i][::3])


def test_save_load_ndarray(tmp_path):
    a = np.arange(60, dtype=np.int64).reshape(20, 3)
    name = str(tmp_path / "a.h5")
    ra.save(name, a)
    with h5lite.File(name) as f:
        assert f.keys() == ["arr_0"]
    b = ra.load(name)
    assert isinstance(b, np.ndarray)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ra.load(name, keys=None, stride=4), a[::4])


def test_load_specified_keys(tmp_path):
    rows = [np.arange(12.0).reshape(4, 3), np.arange(6.0).reshape(2, 3) + 50,
            np.arange(9.0).reshape(3, 3) - 7]
    name = str(tmp_path / "k.h5")
    h5lite.write(name, {"key0": rows[0], "key1": rows[1], "key2": rows[2]},
                 pytables=False)
    b = ra.load(name, keys=["key1", "key2"])
    np.testing.assert_array_equal(b.lengths, [2, 3])
    np.testing.assert_array_equal(b[0], rows[1])
    np.testing.assert_array_equal(b[1], rows[2])
    one = ra.load(name, keys=["key0"])
    assert isinstance(one, np.ndarray)
    np.testing.assert_array_equal(one, rows[0])
    with pytest.raises(KeyError):
        ra.load(name, keys=["key0", "nope"])


def test_load_rejects_mismatched_rows(tmp_path):
    name = str(tmp_path / "bad.h5")
    h5lite.write(name, {"key0": np.zeros((4, 3)), "key1": np.zeros((4, 2))})
    with pytest.raises(DataInvalid):
        ra.load(name, keys=["key0", "key1"])
    h5lite.write(name, {"key0": np.zeros((4, 3)), "key1": np.zeros(4)})
    with pytest.raises(DataInvalid):
        ra.load(name, keys=["key0", "key1"])
    h5lite.write(name, {"key0": np.zeros(4, dtype="f4"), "key1": np.zeros(4)})
    with pytest.raises(DataInvalid):
        ra.load(name)


def test_load_old_style(tmp_path):
    a = _ragged(np.int32)
    name = str(tmp_path / "old.h5")
    h5lite.write(name, {"array": a._data, "lengths": a.lengths}, pytables=False)
    b = ra.load(name, keys=None)
    np.testing.assert_array_equal(b.lengths, a.lengths)
    np.testing.assert_array_equal(b._data, a._data)


def test_not_hdf5(tmp_path):
    name = str(tmp_path / "x.h5")
    with open(name, "wb") as f:
        f.write(b"not a container" * 100)
    with pytest.raises(DataInvalid):
        ra.load(name)


# ---------------------------------------------------------------------------
# against libhdf5
# ---------------------------------------------------------------------------
DTYPES = ["f4", "f8", "f2", "i8", "i4", "i2", "i1", "u1", "u2", "u4", "u8"]


@needs_tools
@pytest.mark.parametrize("level", [0, 1, 6])
def test_libhdf5_reads_what_we_write(tmp_path, level):
    rng = np.random.default_rng(level)
    arrays = {}
    for i, dt in enumerate(DTYPES):
        shape = (50 + i, 3) if i % 2 else (40 + i,)
        arrays["t_" + dt] = (rng.random(shape) * 100).astype(dt)
    arrays["flag"] = rng.random((33, 2)) < 0.5
    arrays["words"] = np.array([b"ab", b"", b"hello"], dtype="S5")
    arrays["empty"] = np.zeros((0, 4), dtype="f4")
    arrays["scalar"] = np.array(2.5)
    name = str(tmp_path / "w.h5")
    h5lite.write(name, arrays, compression_level=level)
    r = subprocess.run([H5DUMP, "-H", name], capture_output=True, text=True)
    assert r.returncode == 0 and "error" not in r.stderr.lower(), r.stderr
    for key, a in arrays.items():
        if a.size == 0:
            assert 'DATASET "%s"' % key in r.stdout
            continue
        disk = np.int8 if a.dtype == bool else a.dtype
        got = _dump(name, key, disk, a.shape, tmp_path)
        np.testing.assert_array_equal(got, a.astype(disk), err_msg=key)
    assert 'H5T_ENUM' in r.stdout and '"TRUE"' in r.stdout
    with h5lite.File(name) as f:                 # and our own reader
        for key, a in arrays.items():
            b = f[key].read()
            assert b.dtype == a.dtype and b.shape == a.shape
            np.testing.assert_array_equal(b, a)


@needs_tools
def test_deep_trees_read_by_libhdf5(tmp_path, monkeypatch):
    """More rows than one group B-tree node indexes (8 x 32) and more chunks
    than one chunk B-tree node holds (64): both trees get a second level."""
    monkeypatch.setattr(h5lite, "_CHUNK_BYTES", 256)
    rng = np.random.default_rng(0)
    rows = [rng.integers(0, 1000, size=(rng.integers(0, 6), 2)).astype("i4")
            for _ in range(700)]
    rows[5] = rng.integers(0, 1000, size=(9001, 2)).astype("i4")   # 282 chunks
    a = ra.RaggedArray(rows)
    name = str(tmp_path / "deep.h5")
    ra.save(name, a, tag="row")
    r = subprocess.run([_tool("h5ls"), name], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    listed = [line.split()[0] for line in r.stdout.splitlines()]
    assert listed == ["row_%04d" % i for i in range(700)]
    for i in (0, 5, 255, 256, 257, 699):
        if len(rows[i]):
            got = _dump(name, "row_%04d" % i, "i4", rows[i].shape, tmp_path)
            np.testing.assert_array_equal(got, rows[i])
    b = ra.load(name)
    np.testing.assert_array_equal(b.lengths, a.lengths)
    np.testing.assert_array_equal(b._data, a._data)
    # the same content with both kinds of tree rebuilt by libhdf5
    again = str(tmp_path / "deep_repacked.h5")
    r = subprocess.run([H5REPACK, "-l", "CHUNK=32x2", "-f", "GZIP=1", name, again],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    c = ra.load(again)
    np.testing.assert_array_equal(c.lengths, a.lengths)
    np.testing.assert_array_equal(c._data, a._data)


@needs_tools
@pytest.mark.parametrize("args", [
    ["-l", "CONTI"],
    ["-f", "NONE", "-l", "COMPA"],
    ["-l", "CHUNK=7x2x3", "-f", "GZIP=9"],
    ["-l", "CHUNK=64x5x1", "-f", "SHUF", "-f", "FLET"],
    ["-l", "CHUNK=3x5x3", "-f", "SHUF", "-f", "GZIP=4", "-f", "FLET"],
    ["-l", "CHUNK=1x1x1"],
])
def test_reads_what_libhdf5_writes(tmp_path, args):
    rng = np.random.default_rng(11)
    rows = [rng.normal(size=(n, 5, 3)).astype("f4") for n in (150, 40, 77)]
    a = ra.RaggedArray(rows)
    ours, theirs = str(tmp_path / "ours.h5"), str(tmp_path / "theirs.h5")
    ra.save(ours, a)
    r = subprocess.run([H5REPACK] + args + [ours, theirs],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    b = ra.load(theirs)
    np.testing.assert_array_equal(b.lengths, a.lengths)
    np.testing.assert_array_equal(b._data, a._data)
    with h5lite.File(theirs) as f:
        kind = f["arr_00"]._layout[0]
    assert kind == {"CONTI": "contiguous", "COMPA": "compact"}.get(
        args[-1], "chunked")


# ---------------------------------------------------------------------------
# trajectories -> concatenated coordinates (reference util/load.py:52-161,
# test_ra.py:411-520 with the HDF5 copy of the same frames)
# ---------------------------------------------------------------------------
def test_load_as_concatenated(tmp_path):
    from enspara_amd.exception import ImproperlyConfigured
    from enspara_amd.util.load import load_as_concatenated
    h5 = os.path.join(GOLDEN, "frame0.h5")
    with h5lite.File(h5) as f:
        xyz0 = f["coordinates"].read()

    lengths, xyz = load_as_concatenated([h5] * 3, processes=2)
    assert lengths == [501] * 3 and xyz.dtype == np.float32
    np.testing.assert_array_equal(xyz, np.concatenate([xyz0] * 3))

    lengths, xyz = load_as_concatenated(iter([h5, h5]), stride=10, top=None)
    assert lengths == [51, 51]
    np.testing.assert_array_equal(xyz, np.concatenate([xyz0[::10]] * 2))

    sel = np.array([1, 3, 6])
    lengths, xyz = load_as_concatenated([h5] * 2, atom_indices=sel, processes=3)
    np.testing.assert_array_equal(xyz, np.concatenate([xyz0[:, sel]] * 2))

    npy = str(tmp_path / "more.npy")
    np.save(npy, xyz0[:40, [2, 4, 7]])
    lengths, xyz = load_as_concatenated(
        [h5, npy, h5], args=[{"atom_indices": sel}, {"stride": 3}, {"frame": 7,
                             "atom_indices": [2, 4, 7]}])
    assert lengths == [501, 14, 1]
    np.testing.assert_array_equal(
        xyz, np.concatenate([xyz0[:, sel], xyz0[:40:3, [2, 4, 7]],
                             xyz0[7:8, [2, 4, 7]]]))

    lengths, same = load_as_concatenated([h5, npy], lengths=[501, 40],
                                         args=[{"atom_indices": sel}, {}])
    assert len(same) == 541
    with pytest.raises(DataInvalid):
        load_as_concatenated([h5, npy], lengths=[501, 39],
                             args=[{"atom_indices": sel}, {}])
    with pytest.raises(DataInvalid):
        load_as_concatenated([h5, npy])             # 22 atoms, then 3
    with pytest.raises(ImproperlyConfigured):
        load_as_concatenated([h5], args=[{}], stride=2)
    with pytest.raises(ImproperlyConfigured):
        load_as_concatenated([h5, h5], args=[{}])
    with pytest.raises(ImproperlyConfigured):
        load_as_concatenated(["frames.xtc"])
    with pytest.raises(ImproperlyConfigured):
        load_as_concatenated([h5], selection="name CA")
