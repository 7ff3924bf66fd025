"""Synthetic trajectory generator for tests and bench.py.

Frames are noisy, randomly rotated and translated copies of a small set of
random-walk "C-alpha" templates, so pairwise RMSDs spread over a realistic
range instead of the near-constant values iid coordinates would give.
Seeding is counter based per chunk of CHUNK frames: any sharding of the frame
axis that is aligned to CHUNK reproduces the same data.
"""
import numpy as np

CHUNK = 65536
BOND_NM = 0.38
NOISE_NM = 0.05


def templates(n_templates, n_atoms, seed):
    """float64 [n_templates, n_atoms, 3] random-walk chains, bond 0.38 nm."""
    rng = np.random.Generator(np.random.PCG64([seed, 0x7e3a]))
    steps = rng.normal(size=(n_templates, n_atoms, 3))
    steps /= np.linalg.norm(steps, axis=2, keepdims=True)
    steps *= BOND_NM
    steps[:, 0] = 0.0
    return np.cumsum(steps, axis=1)


def _quat_to_rot(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - z * w)
    R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w)
    R[:, 2, 1] = 2 * (y * z + x * w)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def synth_chunk(chunk_index, count, tmpl, seed):
    """Frames [chunk_index*CHUNK, +count) as float32 [count, A, 3]."""
    rng = np.random.Generator(np.random.PCG64([seed, 1 + chunk_index]))
    T = len(tmpl)
    which = rng.integers(0, T, size=count)
    xyz = tmpl[which] + rng.normal(scale=NOISE_NM, size=(count,) + tmpl.shape[1:])
    q = rng.normal(size=(count, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    R = _quat_to_rot(q)
    xyz = np.einsum("nij,naj->nai", R, xyz)
    xyz += rng.uniform(-5.0, 5.0, size=(count, 1, 3))
    return xyz.astype(np.float32)


def synth(n_frames, n_atoms, n_templates, seed, first_frame=0):
    """float32 [n_frames, n_atoms, 3] starting at global frame ``first_frame``
    (which must be a multiple of CHUNK)."""
    if first_frame % CHUNK:
        raise ValueError("first_frame must be a multiple of %d" % CHUNK)
    tmpl = templates(n_templates, n_atoms, seed)
    out = np.empty((n_frames, n_atoms, 3), dtype=np.float32)
    done = 0
    c = first_frame // CHUNK
    while done < n_frames:
        cnt = min(CHUNK, n_frames - done)
        out[done:done + cnt] = synth_chunk(c, cnt, tmpl, seed)
        done += cnt
        c += 1
    return out


def walk(n_frames, n_atoms, seed, kappa=1e-3, sigma=0.03):
    """float32 [n_frames, n_atoms, 3]: ONE time-ordered trajectory on a
    continuous landscape -- every coordinate an Ornstein-Uhlenbeck process around
    a random-walk chain (x[t+1] = x[t] + kappa (x0 - x[t]) + N(0, sigma)),
    stationary spread sigma / sqrt(2 kappa) = 0.67 nm, neighbouring frames
    ~0.05 nm apart, frames 10^4 steps apart unrelated -- then a random rigid
    motion per frame.  No templates, no discrete clusters: the unfriendly case
    for guessing several farthest points ahead (bench.py --data walk)."""
    from scipy.signal import lfilter
    x0 = templates(1, n_atoms, seed)[0]
    rng = np.random.Generator(np.random.PCG64([seed, 0x0a1c]))
    out = np.empty((n_frames, n_atoms, 3), dtype=np.float32)
    zi = np.zeros((1, n_atoms * 3))
    a = np.array([1.0, -(1.0 - kappa)])
    done = 0
    while done < n_frames:
        cnt = min(CHUNK, n_frames - done)
        noise = rng.normal(scale=sigma, size=(cnt, n_atoms * 3))
        dev, zi = lfilter([1.0], a, noise, axis=0, zi=zi)
        xyz = x0[None] + dev.reshape(cnt, n_atoms, 3)
        q = rng.normal(size=(cnt, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        xyz = np.einsum("nij,naj->nai", _quat_to_rot(q), xyz)
        xyz += rng.uniform(-5.0, 5.0, size=(cnt, 1, 3))
        out[done:done + cnt] = xyz.astype(np.float32)
        done += cnt
    return out
