"""tools/c4/: the integer synthetic-trajectory generator behind the one-GPU run
of BASELINE.json configs[3] (tools/c4_one_gpu.py).  Its numpy restatement is
what the oracle checks of that run are made with, so it has to equal the kernel
bit for bit: against the kernel's own source compiled for the host (no GPU),
and on the device."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "c4"))


def _cases():
    return [(0, 300, 7, 50, 37), (123456789012, 200, 2 ** 40 + 5, 7, 5),
            (65536 * 19 - 3, 64, 2, 1, 300)]


def test_numpy_restatement_equals_the_kernel_source_on_the_host():
    import c4gen
    L = C.CDLL(c4gen.build())
    for first, count, seed, T, A in _cases():
        t = c4gen.templates_int(T, A, 3)
        out = np.empty((count, A, 3), np.float32)
        L.c4gen_frames_host(out.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                            C.c_int64(T), C.c_int32(A), C.c_int64(first), C.c_int64(count),
                            C.c_uint64(seed))
        idx = np.arange(first, first + count)
        np.testing.assert_array_equal(c4gen.frames(t, seed, idx), out)
        # any subset, any order
        sub = idx[::-7]
        np.testing.assert_array_equal(c4gen.frames(t, seed, sub), out[::-7])
    # the frames are rigid motions of noisy templates: bonds 0.38 nm +- noise
    x = c4gen.frames(c4gen.templates_int(20, 60, 1), 9, np.arange(500))
    b = np.linalg.norm(np.diff(x, axis=1), axis=2)
    assert 0.37 < b.mean() < 0.42 and b.std() < 0.1


@pytest.mark.gpu
def test_device_generator_equals_numpy():
    import c4gen
    hip = c4gen.Hip()
    for first, count, seed, T, A in _cases():
        t = c4gen.templates_int(T, A, 3)
        gen = c4gen.DeviceGenerator(t, hip)
        buf = hip.malloc(count * A * 12)
        gen.fill(buf, first, count, seed)
        hip.sync()
        got = np.empty((count, A, 3), dtype=np.float32)
        hip.to_host(got, buf)
        hip.free(buf)
        np.testing.assert_array_equal(
            got, c4gen.frames(t, seed, np.arange(first, first + count)))
