"""The integer synthetic-trajectory generator of tools/c4/c4gen.hip restated in
numpy, bit for bit, for ANY subset of the frames -- plus the ctypes binding of
the device version.  Measurement infrastructure (tools/), not product code.

    templates_int(T, A, seed)          int32 [T, A, 3], units of 1e-6 nm
    frames(tmpl, seed, indices)        float32 [len(indices), A, 3]  (host, numpy)
    DeviceGenerator(tmpl).fill(ptr, first, count, seed, stream)      (device)
    Hip()                              hipMalloc / hipMemcpy through ctypes
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libc4gen.so")
U = np.uint64
FRAME = U(0xFFFFFFFF)


def _mix(z):
    z = z + U(0x9E3779B97F4A7C15)
    z = (z ^ (z >> U(30))) * U(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> U(27))) * U(0x94D049BB133111EB)
    return z ^ (z >> U(31))


def _key(seed, f, a, k):
    with np.errstate(over="ignore"):
        return _mix(_mix(U(seed) * U(0x100000001B3) + f) + (a * U(8) + U(k)))


def templates_int(T, A, seed):
    """random-walk chains, bond 0.38 nm, as int32 in units of 1e-6 nm (the
    float64 chain of enspara_amd.synth.templates, rounded)"""
    from enspara_amd import synth
    return np.rint(synth.templates(T, A, seed) * 1e6).astype(np.int32)


def frames(tmpl, seed, indices):
    """the frames with the given global indices, float32 [m, A, 3]"""
    with np.errstate(over="ignore"):
        f = np.asarray(indices, dtype=np.uint64).reshape(-1)
        T, A = tmpl.shape[0], tmpl.shape[1]
        which = (_key(seed, f, FRAME, 0) % U(T)).astype(np.int64)
        hq = _key(seed, f, FRAME, 1)
        q = [((hq >> U(s)) & U(0xFFFF)).astype(np.int64) - 32768 for s in (0, 16, 32, 48)]
        w, x, y, z = q
        N = w * w + x * x + y * y + z * z
        small = N < (1 << 24)
        w = np.where(small, 1, w)
        x, y, z = (np.where(small, 0, c) for c in (x, y, z))
        N = np.where(small, 1, N)
        R = [w * w + x * x - y * y - z * z, 2 * (x * y - z * w), 2 * (x * z + y * w),
             2 * (x * y + z * w), w * w - x * x + y * y - z * z, 2 * (y * z - x * w),
             2 * (x * z - y * w), 2 * (y * z + x * w), w * w - x * x - y * y + z * z]
        a = np.arange(A, dtype=np.uint64)[None, :]
        v = []
        for k in range(3):
            h = _key(seed, f[:, None], a, k)
            s = sum(((h >> U(sh)) & U(0xFFFF)).astype(np.int64) for sh in (0, 16, 32, 48)) - 131070
            noise = (s * 86603) // 65536            # floor division, as the kernel's
            v.append(tmpl[which, :, k].astype(np.int64) + noise)
        out = np.empty((len(f), A, 3), dtype=np.float32)
        for i in range(3):
            ht = _key(seed, f, FRAME, 2 + i)
            t = (((ht & U(0xFFFFFF)) * U(10000000)) >> U(24)).astype(np.int64) - 5000000
            num = (R[3 * i][:, None] * v[0] + R[3 * i + 1][:, None] * v[1] +
                   R[3 * i + 2][:, None] * v[2])
            r = num // N[:, None]
            out[:, :, i] = (r + t[:, None]).astype(np.float32) * np.float32(1e-6)
        return out


def build(force=False):
    src = os.path.join(HERE, "c4gen.hip")
    if (not force and os.path.exists(SO)
            and os.path.getmtime(SO) >= os.path.getmtime(src)):
        return SO
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"),
                           "--offload-arch=gfx950", "-O3", "-fPIC", "-shared",
                           "-ffp-contract=off", "-o", SO, src])
    return SO


class Hip:
    """the few runtime calls this tool needs, through ctypes (so that it can run
    with or without torch in the process: EK_C4_NO_TORCH=1)"""

    def __init__(self):
        # (by soname: the copy the process has loaded already -- torch's, or the
        # system's -- not a second runtime beside it)
        self.L = C.CDLL("libamdhip64.so.7")
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipFree.argtypes = [C.c_void_p]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (hipError %d)" % (what, rc))

    def malloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.L.hipMalloc(C.byref(p), nbytes), "hipMalloc(%d)" % nbytes)
        return p.value

    def free(self, ptr):
        self.L.hipFree(C.c_void_p(ptr))

    def to_device(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.L.hipMemcpy(C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p),
                                    arr.nbytes, 1), "hipMemcpy H2D")

    def to_host(self, arr, ptr):
        self.check(self.L.hipMemcpy(arr.ctypes.data_as(C.c_void_p), C.c_void_p(ptr),
                                    arr.nbytes, 2), "hipMemcpy D2H")

    def sync(self):
        self.check(self.L.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def mem_info(self):
        f, t = C.c_size_t(), C.c_size_t()
        self.check(self.L.hipMemGetInfo(C.byref(f), C.byref(t)), "hipMemGetInfo")
        return f.value, t.value


class DeviceGenerator:
    """templates on the device, frames written into device memory the caller
    names by address"""

    def __init__(self, tmpl, hip=None):
        self.hip = hip or Hip()
        self.L = C.CDLL(build())
        self.L.c4gen_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                        C.c_int64, C.c_int64, C.c_uint64, C.c_void_p]
        self.T, self.A = int(tmpl.shape[0]), int(tmpl.shape[1])
        t = np.ascontiguousarray(tmpl, dtype=np.int32)
        self.tmpl = self.hip.malloc(t.nbytes)
        self.hip.to_device(self.tmpl, t)

    def fill(self, out_ptr, first, count, seed, stream=None):
        rc = self.L.c4gen_frames(C.c_void_p(int(out_ptr)), C.c_void_p(self.tmpl),
                                 self.T, self.A, int(first), int(count), int(seed),
                                 C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise RuntimeError("c4gen_frames failed (%d)" % rc)
