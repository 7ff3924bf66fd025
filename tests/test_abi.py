"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/enspara_hip.h declares; device calls fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "enspara_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ek_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from enspara_amd import _lib
    L = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    assert sorted(_lib.SYMBOLS) == names
    assert L.ek_abi_version() == 1
    assert L.ek_record_bytes(300) == 32 + 3600
    assert L.ek_record_bytes(22) % 16 == 0


def test_no_torch_types_in_the_header():
    src = open(os.path.join(ROOT, "include", "enspara_hip.h")).read()
    assert "extern \"C\"" in src
    for bad in ("at::", "torch::", "Tensor", "std::"):
        assert bad not in src


def _has_gpu():
    from enspara_amd import _lib
    return _lib.load().ek_device_count() > 0


def test_rmsd_path_fails_loudly_without_a_device():
    """On a box with no HIP device the 'rmsd' path must raise, never fall back
    to a CPU computation."""
    if _has_gpu():
        pytest.skip("a HIP device is present")
    from enspara_amd import _lib
    from enspara_amd.cluster import KCenters
    x = np.zeros((10, 4, 3), dtype=np.float32)
    with pytest.raises(_lib.HipError):
        KCenters("rmsd", n_clusters=2).fit(x)
    L = _lib.load()
    h = ctypes.c_void_p()
    rc = L.ek_ctx_create(0, 10, 4, 0, None, ctypes.byref(h))
    assert rc == _lib.EK_EHIP
    assert b"hip" in L.ek_last_error().lower()


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from enspara_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.load()


def test_argument_errors():
    from enspara_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.ek_ctx_create(0, -1, 4, 0, None, ctypes.byref(h)) == _lib.EK_EARG
    assert L.ek_ctx_create(0, 10, 0, 0, None, ctypes.byref(h)) == _lib.EK_EARG
    assert L.ek_ctx_create(0, 10, 100000, 0, None,
                           ctypes.byref(h)) == _lib.EK_EARG
    assert L.ek_ctx_create(0, 10, 4, 0, None, None) == _lib.EK_EARG
    assert L.ek_state_reset(None) == _lib.EK_EARG
    assert L.ek_kcenters_run(None, 0, 1, 0.0, None, None, None,
                             None) == _lib.EK_EARG


def test_product_does_not_import_the_oracle():
    """enspara_amd/ must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "enspara_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text,
                                     flags=re.M), f
                assert "liboracle" not in text, f


def test_option_names_match_the_header():
    """enspara_amd._lib's EK_OPT_* constants are include/enspara_hip.h's
    `enum ek_option`, member for member"""
    from enspara_amd import _lib
    src = open(os.path.join(ROOT, "include", "enspara_hip.h")).read()
    body = src[src.index("enum ek_option {"):]
    body = re.sub(r"/\*.*?\*/", "", body[:body.index("};")], flags=re.S)
    pairs = {n: int(v) for n, v in re.findall(r"(EK_OPT_[A-Z_0-9]+)\s*=\s*(-?\d+)", body)}
    assert len(pairs) >= 20
    mine = {k: v for k, v in vars(_lib).items() if k.startswith("EK_OPT_")}
    assert mine == pairs
    assert _lib.OPTIONS["candidates"] == pairs["EK_OPT_CANDIDATES"] == 4
    assert len(set(pairs.values())) == len(pairs)
