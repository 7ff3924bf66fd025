"""Inner-product matrices for the soundness tests of the early-stopped QCP solve
and its float32 certificate (enspara_amd/csrc/ek_qcp.h): S = U diag(s1, s2, s3)
V^T with the spectrum drawn from a named family -- generic, and every way two
roots of the quartic can come close.  Shared by tests/test_qcp_host.py (the
header compiled with g++: correctly rounded 1/x, sqrt) and
tests/test_gpu_qcp_device.py (the same functions as a kernel: v_rcp_f32,
v_sqrt_f32, v_rsq_f32, the instructions that ship)."""
import zlib

import numpy as np

FAMILIES = ["generic", "s1~s2", "s2~-s3", "s2~s3", "rank1", "rank2", "isotropic",
            "small", "large", "tiny", "huge"]


def rotations(rng, m):
    q = rng.normal(size=(m, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
                    axis=1).reshape(m, 3, 3)


def family_case(family, m, A, seed=0):
    """-> S float32 [m, 9], Gsum [m], and of the float32 matrix the kernels
    would hold: singular values sv [m, 3], the signed third t3, q = sum sv^2,
    lam = the quartic's largest root"""
    rng = np.random.default_rng(zlib.crc32(family.encode()) % 10**6 + seed)
    s1 = A * 10.0 ** rng.uniform(-1, 1, m)
    u, v = rng.random(m), rng.random(m)
    tiny = 10.0 ** rng.uniform(-9, -1, m)
    sign = np.where(rng.random(m) < 0.5, -1.0, 1.0)
    if family == "generic":
        s2, s3 = s1 * u, s1 * u * v * sign
    elif family == "s1~s2":
        s2, s3 = s1 * (1 - tiny), s1 * u * sign
    elif family == "s2~-s3":
        s2 = s1 * u
        s3 = -s2 * (1 - tiny)
    elif family == "s2~s3":
        s2 = s1 * u
        s3 = s2 * (1 - tiny)
    elif family == "rank1":
        s2, s3 = s1 * tiny, s1 * tiny * v * sign
    elif family == "rank2":
        s2, s3 = s1 * u, s1 * tiny * u * sign
    elif family == "isotropic":
        s2, s3 = s1 * (1 - tiny), s1 * (1 - tiny * (1 + v)) * sign
    else:
        s1 = s1 * {"small": 1e-5, "large": 1e4, "tiny": 1e-15, "huge": 1e9}[family]
        s2, s3 = s1 * u, s1 * u * v * sign
    sig = np.stack([s1, s2, s3], axis=1)
    S = np.einsum("mik,mk,mjk->mij", rotations(rng, m), sig, rotations(rng, m))
    S = np.ascontiguousarray(S.reshape(m, 9), dtype=np.float32)
    # the spectrum of the float32 matrix the kernels would hold
    M = S.astype(np.float64).reshape(m, 3, 3)
    sv = np.linalg.svd(M, compute_uv=False)
    t3 = np.where(np.linalg.det(M) < 0, -sv[:, 2], sv[:, 2])
    lam = sv[:, 0] + sv[:, 1] + t3
    q = (sv ** 2).sum(1)
    Gsum = 2 * lam + A * lam.clip(1e-300) / A * 10.0 ** rng.uniform(-7, 1, m)
    return S, np.ascontiguousarray(Gsum), sv, t3, q, lam


def coincident_case(eps, m, A):
    """S = U diag(s1, s2, -s2 (1 + eps)) V^T: the two largest roots of the quartic,
    s1+s2+s3 and s1-s2-s3, differ by 2 s2 eps -> S, Gx (= Gy)"""
    rng = np.random.default_rng(int(eps * 1e13) + 5)
    s1 = A * rng.uniform(0.5, 3.5, m)
    s2 = s1 * rng.random(m)
    s3 = -s2 * (1 + eps)
    sig = np.stack([s1, s2, s3], axis=1)
    S = np.einsum("mik,mk,mjk->mij", rotations(rng, m), sig, rotations(rng, m))
    S = np.ascontiguousarray(S.reshape(m, 9), dtype=np.float32)
    top = np.maximum(s1 + s2 + s3, s1 - s2 - s3)
    Gsum = 2 * top + A * 10.0 ** rng.uniform(-6, 0.5, m)
    return S, np.ascontiguousarray(Gsum / 2)


def structure_pairs(rng, A, m, squash=1.0):
    """m (frame, center) pairs of A atoms; squash < 1 flattens y and z."""
    scale = rng.uniform(0.5, 3.5, size=(m, 1, 1))
    shape = np.array([1.0, squash, squash])
    x = rng.normal(size=(m, A, 3)) * scale * shape
    similar = rng.random(m) < 0.5
    y = np.where(similar[:, None, None],
                 x + 0.05 * rng.normal(size=(m, A, 3)) * shape,
                 rng.normal(size=(m, A, 3)) * scale * shape)
    return x.astype(np.float32), y.astype(np.float32)
