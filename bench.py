#!/usr/bin/env python3
"""bench.py -- frame x center RMSD pairs/s of the k-centers hot path on MI355X.

    python bench.py [--gpus N] [--steps S] [--warmup W]

Workload: BASELINE.json configs[1] -- 1,000,000 synthetic frames x 300 atoms,
5000 centers -- whatever S is.  The timed region is the WHOLE fit, from the
untouched state (every distance +inf) to 5000 centers; a "step" is
5000 / S consecutive centers of it (250 with the driver's --steps 20; one
center with --steps 5000), each center being its RMSD to every frame, the
strict-< update and the farthest-point reduction: n_frames pairs.
`value` = n_frames * 5000 / elapsed; `ms_per_step` = elapsed / S.
The W warm-up steps are W * (5000 / S) centers of a throw-away fit from the
same untouched state; the state is reset before the clock starts.

N > 1: `python bench.py --gpus N` starts N worker processes itself (one per
GPU, RCCL) before touching any GPU; under `torch.distributed.run` the ranks
it started are used as they are.  Default there is STRONG scaling -- the same
1,000,000 frames split into N contiguous blocks --
`--scaling weak` holds 1,000,000 frames per GPU instead.

The frames are streamed once per ROUND against up to 16 candidate centers and
further centers are accepted from the stored distances while the farthest
point is one of them (csrc/ek_spec.hip, ek_pass16.hip): the same
sequential algorithm and bit-identical results with fewer passes over HBM.
The fit moves between 1, 8 and 16 candidates per round by the centers per
millisecond each achieves (`--candidates 1|4|8|16|32` pins one form; 1 is the HBM
roofline case of BASELINE.md; rounds of 32 only when pinned: measured to lose).  Every reported pair is a distance the result
depends on; guesses that were never used are not counted ("pairs_computed" has
the total).  `roofline` describes the kernel most passes ran: the 16-candidate
pass is a dense contraction on the matrix cores (bound "mfma", its HBM figures
beside it), the 8-candidate and one-center passes are HBM-bound.

N > 1: the rounds exchange one message per shard and round through peer
mailboxes written on the device (csrc/ek_mshard.hip; `--transport gather` uses
one RCCL all-gather per round instead).

Inputs are resident in HBM (already centred and laid out frame-minor) when the
timed region starts; generation, upload and layout are reported separately in
"setup".  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E nominal (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X dense f32 matrix peak (same guide)


def bytes_per_pair(n_atoms):
    """Algorithmic HBM bytes per frame x center pair of the one-center pass
    (DESIGN.md section 5): coordinates 12*A, trace 8 (f64), distance
    read+write 8, label write 4."""
    return 12 * n_atoms + 20


def bytes_per_frame_pass(n_atoms, cands):
    """Algorithmic HBM bytes per frame of one multi-candidate pass: the same
    stream plus one stored float32 distance per extra candidate."""
    return 12 * n_atoms + 20 + 4 * (cands - 1)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--centers", type=int, default=5000,
                   help="centers of the fit (BASELINE.json configs[1]: 5000)")
    p.add_argument("--frames", type=int, default=1_000_000,
                   help="frames in total (strong scaling) or per GPU (weak)")
    p.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                   help="N > 1: split --frames over the GPUs (strong, the "
                        "target of BASELINE.md) or hold --frames on each")
    p.add_argument("--atoms", type=int, default=300)
    p.add_argument("--templates", type=int, default=5000)
    p.add_argument("--data", choices=["templates", "walk"], default="templates",
                   help="templates: noisy copies of --templates chains in random "
                        "order (BASELINE's synthetic set); walk: one time-ordered "
                        "trajectory on a continuous landscape (one GPU only)")
    p.add_argument("--triangle", type=int, default=0, choices=[0, 1],
                   help="the reference's use_triangle_inequality (kcenters.py:287-296): "
                        "rounds leave out the tiles none of their candidates can change; "
                        "`pairs_evaluated` then counts what was really streamed")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--fpl", type=int, default=0,
                   help="frames per lane of the one-center kernel (0 = auto)")
    p.add_argument("--candidates", type=int, default=-1,
                   help="candidate centers per pass: -1 by measured rate "
                        "(1, 8, 16), or pin 1, 4, 8, 16 or 32 (32: two "
                        "passes of 16 behind one plan and chain; never "
                        "chosen automatically)")
    p.add_argument("--transport", choices=["mailbox", "gather"], default="mailbox",
                   help="N > 1 / --sharded: a round's exchange through peer "
                        "mailboxes on the device, or one all-gather per round")
    p.add_argument("--no-msm", action="store_true",
                   help="skip the MSM block (BASELINE.json configs[4])")
    p.add_argument("--msm-frames", type=int, default=10_000_000)
    p.add_argument("--cpu-seconds", type=float, default=15.0,
                   help="time budget of the CPU baseline leg (rank 0, N=1); 0 = "
                        "no budget: the oracle replays ALL centers of the fit "
                        "(labels and distances compared too) and the complete "
                        "PAM sweep -- minutes of CPU work")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--pam-runs", type=int, default=2,
                   help="repetitions of the PAM measurement, each from the same "
                        "k-centers result (s_per_sweep is the best, all are listed)")
    p.add_argument("--pam-sweeps", type=int, default=None,
                   help="after the timed k-centers region, time this many PAM "
                        "(k-medoids) sweeps over the centers found -- the "
                        "k-hybrid refinement of BASELINE.json configs[2]; "
                        "reported beside the metric, never part of it "
                        "(default: 1 on one GPU, 0 on several)")
    p.add_argument("--sharded", action="store_true",
                   help="use the torch.distributed driver even with one rank "
                        "(exercises the RCCL path on a 1-GPU box)")
    return p.parse_args()


# ---------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks here, before any GPU is touched
# ---------------------------------------------------------------------------
def spawn_ranks(args):
    """Start --gpus worker processes of this same script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment), forward rank 0's JSON line
    and exit non-zero if any of them failed.  The parent never initialises
    HIP: it does not import torch."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r),
                    "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY":
                        os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
            env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's line is read by a thread; all children are watched: the first one
    # to fail takes the others down instead of leaving them in a collective
    import threading
    out = []
    t = threading.Thread(target=lambda: out.append(procs[0].stdout.read()))
    t.start()
    bad = []
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c]
        if bad or all(c is not None for c in codes):
            break
        time.sleep(0.2)
    if bad:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    t.join(timeout=10)
    sys.stdout.write(b"".join(out).decode())
    sys.stdout.flush()
    if bad:
        raise SystemExit("ranks failed (rank, exit code): %s" % bad)


def rank_frames(args, rank, world):
    """(first frame of the generator's stream, global offset, count) of this
    rank's contiguous block."""
    from enspara_amd import sharded, synth
    if world == 1:
        return 0, 0, args.frames
    if args.scaling == "strong":
        lo, cnt = sharded.shard_bounds(args.frames, world, rank)
        return lo, lo, cnt
    # weak: every rank its own --frames, generated from a chunk boundary on
    per = (args.frames + synth.CHUNK - 1) // synth.CHUNK * synth.CHUNK
    return rank * per, rank * args.frames, args.frames


def stream_length(args, rank, world):
    """Length of the generator stream this rank's block is cut from."""
    from enspara_amd import synth
    if world == 1 or args.scaling == "strong":
        return args.frames
    per = (args.frames + synth.CHUNK - 1) // synth.CHUNK * synth.CHUNK
    return rank * per + args.frames


def make_shard(args, lo, count, n_stream):
    """Frames [lo, lo + count) of the synthetic stream of n_stream frames,
    float32 [count, atoms, 3].  The generator is seeded per 65,536-frame chunk
    and a chunk's draws depend on its length, so every chunk is generated at
    the length it has in the whole stream and then cut: a block is the same
    data whatever the number of ranks."""
    from enspara_amd import synth
    tmpl = synth.templates(args.templates, args.atoms, args.seed)
    out = np.empty((count, args.atoms, 3), dtype=np.float32)
    c = lo // synth.CHUNK
    done = 0
    while done < count:
        c_lo = c * synth.CHUNK
        c_len = min(synth.CHUNK, n_stream - c_lo)
        chunk = synth.synth_chunk(c, c_len, tmpl, args.seed)
        a = max(lo, c_lo) - c_lo
        b = min(lo + count, c_lo + c_len) - c_lo
        out[done:done + b - a] = chunk[a:b]
        done += b - a
        c += 1
    return out


def host_threads():
    """Threads the CPU leg may really use: the affinity mask and the cgroup
    CPU quota, not the machine's core count."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, len(os.sched_getaffinity(0)), quota


def cpu_baseline(x, gpu_centers, seconds, gpu_state=None):
    """Time the CPU oracle (oracle/qcp_oracle.c: OpenMP, frames centred once,
    frame-minor tiles, AVX2 FMA across frames) on the same frames for as many
    leading k-centers iterations as fit the time budget, and check the GPU's
    centers for those iterations against it.  ``seconds <= 0`` (--cpu-seconds
    0): no budget -- every center of the fit, and with ``gpu_state`` (the
    distances and labels the same fit left on the device) every frame's label
    and float32 distance too.  The oracle is the checker and the reported
    baseline, never the product path."""
    from oracle import qcp
    usable, affinity, quota = host_threads()
    qcp.set_num_threads(usable)
    t0 = time.perf_counter()
    P = qcp.Prepared(x)
    _ = P.tiled          # written by the threads that read it later (first touch)
    prep = time.perf_counter() - t0
    n = P.n
    dist = np.full(n, np.inf, dtype=np.float32)
    assign = np.full(n, -1, dtype=np.int32)
    centers = []
    nxt = 0
    t0 = time.perf_counter()
    while len(centers) < len(gpu_centers):
        centers.append(nxt)
        mx, nxt = P.kcenters_step(P.c[nxt], P.G[nxt], len(centers) - 1, dist,
                                  assign)
        if seconds > 0 and time.perf_counter() - t0 > seconds:
            break
    wall = time.perf_counter() - t0
    k = len(centers)
    ok = [int(c) for c in gpu_centers[:k]] == centers
    state_ok = None
    if k == len(gpu_centers) and gpu_state is not None:
        state_ok = {"labels_equal": bool(np.array_equal(gpu_state["assign"], assign)),
                    "distances_equal_f32": bool(np.array_equal(
                        np.asarray(gpu_state["dist"], dtype=np.float32), dist))}
    if seconds <= 0:
        seconds = 15.0          # the one-thread figure keeps its small budget
    cores = qcp.num_threads()
    # the same loop on one thread, for a fifth of the budget
    one = None
    if cores > 1:
        qcp.set_num_threads(1)
        try:
            d1 = np.full(n, np.inf, dtype=np.float32)
            a1 = np.full(n, -1, dtype=np.int32)
            k1, nxt = 0, 0
            t0 = time.perf_counter()
            while k1 < max(1, len(gpu_centers)):
                mx, nxt = P.kcenters_step(P.c[nxt], P.G[nxt], k1, d1, a1)
                k1 += 1
                if time.perf_counter() - t0 > seconds / 5.0:
                    break
            w1 = time.perf_counter() - t0
            one = {"value": n * k1 / w1,
                   "sample": "%d iterations, %.1f s" % (k1, w1)}
        finally:
            qcp.set_num_threads(cores)
    out = {
        "value": n * k / wall,
        "unit": "pairs/s",
        "cores": cores,
        "kind": "port",
        "sample": "all %d frames x first %d k-centers iterations "
                  "(%.1f s; centring+layout %.1f s not included)"
                  % (n, k, wall, prep),
        "centers_checked": k,
        "centers_match_gpu": bool(ok),
        "whole_fit_state_vs_gpu": state_ok,
        "host": {"logical_cpus": os.cpu_count(), "affinity_cpus": affinity,
                 "cgroup_cpu_quota": quota,
                 "threads_policy": "OpenMP static schedule over tiles; the "
                                   "tiled copy is first-touched by the threads "
                                   "that stream it; no explicit NUMA binding"},
    }
    if one:
        out["one_thread"] = one
    out["mdtraj"] = mdtraj_leg(x, centers, min(seconds, 10.0))
    return out


def mdtraj_leg(x, centers, seconds):
    """If mdtraj happens to be importable on this box, time the reference's own
    metric -- md.rmsd(traj, traj, frame, precentered=True), what enspara
    binds as 'rmsd' (enspara/cluster/util.py:289-291, :629) -- through the same
    leading iterations.  Probed at run time, never assumed."""
    try:
        import mdtraj as md
    except Exception as e:
        return {"available": False, "why": "%s: %s" % (type(e).__name__, e)}
    try:
        top = md.Topology()
        ch = top.add_chain()
        for _ in range(x.shape[1]):
            r = top.add_residue("ALA", ch)
            top.add_atom("CA", md.element.carbon, r)
        t = md.Trajectory(x, top)
        t.center_coordinates()
        dist = np.full(len(x), np.inf)
        k = 0
        t0 = time.perf_counter()
        for c in centers:
            d = md.rmsd(t, t, int(c), precentered=True)
            m = d < dist
            dist[m] = d[m]
            k += 1
            if time.perf_counter() - t0 > seconds:
                break
        wall = time.perf_counter() - t0
        return {"available": True, "label": "enspara/mdtraj",
                "version": md.__version__, "value": len(x) * k / wall,
                "unit": "pairs/s", "sample": "%d iterations, %.1f s" % (k, wall)}
    except Exception as e:      # an installed but unusable mdtraj is not fatal
        return {"available": False, "why": "%s: %s" % (type(e).__name__, e)}


def copy_ceiling(nbytes, device):
    """-> (copy GB/s, read-only GB/s) of this GPU, in this process: a
    16-byte-per-lane non-temporal copy of `nbytes` (read + write = 2 x nbytes of
    HBM traffic) and the same stream only read (what the distance kernels do),
    best of four each (csrc/ek_api.hip ek_hbm_copy_rate).  The ceilings a
    streaming kernel can be compared with besides the nominal peak."""
    import ctypes as C
    from enspara_amd import _lib
    g = (C.c_double * 2)()
    _lib.check(_lib.load().ek_hbm_copy_rate(int(device), int(nbytes), g))
    return g[0], g[1]


def msm_block(args):
    """BASELINE.json configs[4]: 10^7 frames of assignments -> sparse transition
    counts -> row-normalise -> top-20 eigenpairs, on the device, with a scipy
    construction of the same matrix (and ARPACK on it) as the CPU figure.  The
    assignments are a seeded banded walk inside metastable blocks with rare hops
    (SURVEY.md 8d), ~0.1 % of the frames -1."""
    import scipy.sparse
    import scipy.sparse.linalg
    from enspara_amd.device import FrameStore
    from enspara_amd.msm import assigns_to_counts, builders, eigenspectrum
    K, L, lag = 5000, 10000, 1
    n_trj = max(1, args.msm_frames // L)
    rng = np.random.RandomState(11)
    steps = rng.choice(np.array([-3, -2, -1, 0, 0, 1, 2, 3], dtype=np.int8),
                       size=(n_trj, L))
    inblock = (rng.randint(100, size=(n_trj, 1)) +
               np.cumsum(steps, axis=1, dtype=np.int32)) % 100
    hops = np.cumsum(rng.rand(n_trj, L) < 0.002, axis=1, dtype=np.int32)
    block = (rng.randint(K // 100, size=(n_trj, 1)) + hops * 7) % (K // 100)
    A = (block * 100 + inblock).astype(np.int32)
    A[rng.rand(n_trj, L) < 0.001] = -1
    del steps, inblock, hops, block
    n = A.size

    def best_of(f, reps=3):
        out, best = None, None
        for _ in range(reps):
            t = time.perf_counter()
            out = f()
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        return out, best
    C, t_host = best_of(lambda: assigns_to_counts(A, lag, max_n_states=K))
    # the same labels resident on the device, as a fit leaves them
    st = FrameStore(n, 1, device=0)
    st.load(np.zeros((n, 1, 3), dtype=np.float32))
    st.upload_state(np.zeros(n, dtype=np.float32), A.reshape(-1))
    st.msm_counts([L] * n_trj, lag, K)          # (buffers allocated)
    st.sync()
    res, t_res = best_of(lambda: st.msm_counts([L] * n_trj, lag, K))
    st.close()
    Cr = scipy.sparse.coo_matrix((res[2], (res[0], res[1])), shape=(K, K)).tocsr()
    (Cn, T, _), t_norm = best_of(
        lambda: builders.normalize(C, calculate_eq_probs=False), reps=2)
    (vals, vecs), t_eig = best_of(lambda: eigenspectrum(T, n_eigs=20), reps=2)
    # CPU: the reference's own construction (a COO of ones, summed on conversion)
    t0 = time.perf_counter()
    rows, cols = [], []
    for a in A:
        a = a[a != -1]
        rows.append(a[:-lag])
        cols.append(a[lag:])
    ref = scipy.sparse.coo_matrix(
        (np.ones(sum(len(r) for r in rows), dtype=np.int64),
         (np.concatenate(rows), np.concatenate(cols))), shape=(K, K)).tocsr()
    t_cpu_counts = time.perf_counter() - t0
    t0 = time.perf_counter()
    Tref = scipy.sparse.diags(1.0 / np.asarray(ref.sum(axis=1)).ravel()) @ ref
    t_cpu_norm = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = scipy.sparse.linalg.eigs(Tref.T.tocsr(), k=20, which="LR", tol=1e-12,
                                    return_eigenvectors=False)
    t_cpu_eig = time.perf_counter() - t0
    want = np.sort(want.real)[::-1]
    return {
        "workload": "MSM build, %d frames of assignments in %d trajectories, %d "
                    "states, lag %d, top-20 eigenpairs (BASELINE.json configs[4])"
                    % (n, n_trj, K, lag),
        "counts_s_labels_resident_on_device": t_res,
        "counts_s_labels_from_host_arrays": t_host,
        "transitions_per_s_resident": n / t_res,
        "GBps_vs_8_bytes_per_transition_resident": 8.0 * n / t_res / 1e9,
        # what bounds the counts (SURVEY.md 8(d) asks for GB/s against 8 N bytes): not
        # HBM -- one scattered 4-byte atomic add per transition on a dense
        # n_states^2 table; MI355X_MICROARCH.md "Global float atomics": 64 lanes in 64
        # different rows = 0.08 TB/s of 4-byte adds = 2e10 atomics/s chip-wide.  The
        # whole call (two compactions, the histogram, two read-backs) against it; the
        # histogram kernel alone runs at 3.4e10 / s (0.29 ms of the call; a walk's
        # neighbouring positions fall on neighbouring cells: profiles/r06/):
        "counts_roofline": {
            "bound": "scattered 4-byte atomics (one per transition)",
            "peak_atomics_per_s": 2.0e10,
            "achieved_transitions_per_s_whole_call": n / t_res,
            "frac": n / t_res / 2.0e10,
            "hbm_frac_vs_8_bytes_per_transition": 8.0 * n / t_res / 1e9 / HBM_PEAK_GBS},
        "nonzero_cells": int(C.nnz),
        "normalize_s": t_norm,
        "top20_eigenpairs_s": t_eig,
        "counts_equal_scipy": bool((C.tocsr() != ref).nnz == 0 and
                                   (Cr != ref).nnz == 0),
        "probabilities_max_abs_diff_vs_scipy": float(abs(T.tocsr() - Tref).max()),
        "eigenvalues_max_abs_diff_vs_arpack": float(np.abs(vals - want).max()),
        "cpu_scipy": {"counts_s": t_cpu_counts, "normalize_s": t_cpu_norm,
                      "top20_arpack_s": t_cpu_eig,
                      "what": "scipy.sparse COO of ones -> CSR (the reference's "
                              "assigns_to_counts), diags @ CSR, scipy.sparse."
                              "linalg.eigs(k=20, which='LR', tol=1e-12)"},
    }


def khybrid_check(x, store, start_medoids, gpu_medoids, seed, seconds,
                  second_sweep=None):
    """Full-size parity of the k-hybrid leg (BASELINE.json configs[2]): the
    first proposals of the sweep replayed by the oracle's PAM
    (oracle/cluster.py pam_update = enspara/cluster/kmedoids.py:575-699) on the
    same frames from the same k-centers state and the same random stream, as
    many as fit the time budget -- the medoids must be the GPU's --, and after
    the GPU's whole sweep a sample of frames: every one's distance must be, bit
    for bit, its RMSD to the medoid its label names.  ``second_sweep`` (with no
    budget): the GPU's medoids and state after the sweep that FOLLOWS from the
    carried RandomState (kmedoids.py:410-476), replayed whole as well.
    ``store`` holds the state the GPU's FIRST sweep left."""
    from oracle import cluster as oc
    from oracle import qcp
    usable, _, _ = host_threads()
    qcp.set_num_threads(usable)
    P = qcp.Prepared(x)
    d0, a0 = start_medoids["dist"], start_medoids["assign"]
    done = []
    t0 = time.perf_counter()
    rs_oracle = np.random.RandomState(seed)
    med, od, oa = oc.pam_update(P, list(start_medoids["medoids"]),
                                a0.astype(np.int64), d0.astype(np.float64),
                                random_state=rs_oracle,
                                budget_s=seconds if seconds > 0 else None,
                                done=done)
    wall = time.perf_counter() - t0
    k = done[0] if done else 0
    same = [int(m) for m in med[:k]] == [int(m) for m in gpu_medoids[:k]]
    moved = sum(1 for i in range(k)
                if int(med[i]) != int(start_medoids["medoids"][i]))
    # sampled state check after the whole sweep
    d, a = store.download_state()
    rng = np.random.RandomState(7)
    sample = rng.choice(len(x), size=min(4000, len(x)), replace=False)
    ok = True
    for lab in np.unique(a[sample]):
        fr = sample[a[sample] == lab]
        want = qcp.rmsd_centered(np.ascontiguousarray(P.c[fr]),
                                 np.ascontiguousarray(P.G[fr]),
                                 P.c[int(gpu_medoids[lab])],
                                 float(P.G[int(gpu_medoids[lab])]))
        ok = ok and bool(np.array_equal(want, d[fr]))
    whole = None
    if k == len(gpu_medoids):   # a complete sweep: the states must be equal
        whole = {"labels_equal": bool(np.array_equal(a, oa)),
                 "distances_equal": bool(np.array_equal(
                     np.asarray(d, dtype=np.float64), od))}
    second = None
    if k == len(gpu_medoids) and second_sweep is not None:
        # the sweep AFTER it (kmedoids.py:410-476: the RandomState carried over,
        # the state the first sweep left), replayed whole
        m1, d1, a1, rs2 = med, od, oa, rs_oracle
        t1 = time.perf_counter()
        m2, d2, a2 = oc.pam_update(P, [int(m) for m in m1], a1, d1,
                                   random_state=rs2)
        second = {"medoids_match_gpu": [int(m) for m in m2] ==
                  [int(m) for m in second_sweep["medoids"]],
                  "labels_equal": bool(np.array_equal(second_sweep["assign"], a2)),
                  "distances_equal": bool(np.array_equal(
                      np.asarray(second_sweep["dist"], dtype=np.float64), d2)),
                  "of_them_accepted": sum(1 for i in range(len(m2))
                                          if int(m2[i]) != int(m1[i])),
                  "oracle_s": time.perf_counter() - t1}
    return {"proposals_replayed_by_oracle": k, "medoids_match_gpu": bool(same),
            "whole_sweep_state_vs_oracle": whole,
            "second_sweep_vs_oracle": second,
            "of_them_accepted": moved, "oracle_s": wall,
            "sampled_frames": int(len(sample)),
            "sampled_state_is_rmsd_to_own_medoid_bit_exact": bool(ok)}


def km_width(sharded_driver):
    """Proposals drawn ahead and decided per window."""
    from enspara_amd.cluster import kmedoids as km
    return 8 if sharded_driver else int(km.PAM_PREFETCH)


def kernel_source_hash():
    """Identifies the code the distance kernels were built from: the traffic
    figure below is only quoted for the sources it was measured on."""
    h = hashlib.sha256()
    for f in ("ek_spec.hip", "ek_pass16.hip", "ek_round.hip", "ek_kcenters.hip",
              "ek_qcp.h", "ek_common.h", "ek_reduce.h", "ek_chain_dev.h",
              "ek_top_dev.h"):
        with open(os.path.join(ROOT, "enspara_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_traffic(args, n_local, cands):
    """HBM bytes per distance-kernel launch from the committed rocprofv3 --pmc
    runs of this same command (profiles/traffic.json) -- but only if that file
    was produced from the kernel sources in this tree and for this shape;
    otherwise null.  -> (bytes or None, provenance dict)"""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    src = kernel_source_hash()
    why = None
    try:
        with open(path) as fh:
            entries = json.load(fh)
        # "cands16" is the default shape; other shapes sit beside it ("cands16_a500")
        t = entries.get("cands%d" % cands)
        for key, e in entries.items():
            if (key.startswith("cands%d_" % cands) and e.get("frames") == n_local
                    and e.get("atoms") == args.atoms):
                t = e
        if t is None:
            why = "no entry for %d candidates per pass" % cands
        elif t.get("frames") != n_local or t.get("atoms") != args.atoms:
            why = "measured at another shape"
        elif t.get("kernel_source_sha256_16") != src:
            why = ("measured on other kernel sources (%s, tree has %s)"
                   % (t.get("kernel_source_sha256_16"), src))
        else:
            return t.get("hbm_bytes_per_launch"), {
                "file": "profiles/traffic.json", "profile": t.get("profile"),
                "kernel_source_sha256_16": src}
    except Exception as e:
        why = "%s: %s" % (type(e).__name__, e)
    return None, {"file": "profiles/traffic.json", "unused_because": why,
                  "kernel_source_sha256_16": src}


def one_center_leg(store, args, n_local, launches=48):
    """BASELINE.md / SURVEY.md 8(d)'s roofline, measured in this run: the
    ONE-center pass (ek_step_kernel -- one frame read serves one center, the
    HBM-bound form north_star describes) for a few dozen launches from the
    untouched state, each timed with HIP events on the context's stream.
    Algorithmic bytes per pair 12 A + 16 (coordinates, trace as counted by
    BASELINE.md, distance read + write, label write); the kernel's own traffic
    is 12 A + 20 (its traces are float64)."""
    before = store.get_option("candidates")
    first = min(2000, max(args.centers - launches, 0))
    try:
        # (in mid-fit, where a fit spends its time: the first `first` centers in
        # the default form, then `launches` one-center passes go on from there)
        store.reset_state()
        store.sync()
        if first:
            store.kcenters_run(0, first, 0.0)
        store.set_option("candidates", 1)
        store.timing_begin(sample_every=1, max_samples=launches)
        t0 = time.perf_counter()
        store.kcenters_run(first, launches, 0.0)
        store.sync()
        wall = time.perf_counter() - t0
        ms, n_samp = store.timing_end()
    finally:
        store.set_option("candidates", before)
    bpp = 12 * args.atoms + 16
    ach = n_local * bpp / (ms * 1e-3) / 1e9 if ms > 0 else None
    traffic, src = load_traffic(args, n_local, 1)
    return {"kernel": "ek_step_kernel<FPL,0,NT>", "bound": "hbm",
            "bytes_per_pair": bpp, "pairs_per_launch": n_local,
            "avg_launch_ms": ms, "launches_sampled": n_samp,
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS if ach else None,
            "pairs_per_s_kernel": n_local / (ms * 1e-3) if ms > 0 else None,
            "pairs_per_s_with_the_pick_between_launches": n_local * launches / wall,
            "ceiling_pairs_per_s": HBM_PEAK_GBS * 1e9 / bpp,
            "traffic": traffic, "traffic_source": src,
            "centers_before_the_sampled_passes": first,
            "what": "BASELINE.md's HBM roofline: one center per pass over the "
                    "frames, 12 A + 16 algorithmic bytes per pair"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    # Libraries underneath (RCCL: "Librccl path : ...") print to stdout; the
    # contract is ONE JSON line there.  Keep the real stdout aside and point
    # fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("--steps >= 1 and --warmup >= 0")
    if args.steps > args.centers:
        raise SystemExit("--steps %d > --centers %d: a step is at least one "
                         "center" % (args.steps, args.centers))

    import torch
    import torch.distributed as dist
    from enspara_amd.device import FrameStore
    from enspara_amd import sharded

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU path")
    # EK_BENCH_ONE_DEVICE=1 (a check of the N > 1 code path on a one-GPU box, not a
    # measurement): every rank on device 0, rendezvous and control tensors over
    # gloo -- RCCL refuses two ranks on one device; the mailboxes do not care
    one_device = os.environ.get("EK_BENCH_ONE_DEVICE") == "1" and world > 1
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.sharded
    ctl = "cpu" if one_device else "cuda"
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29541")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(
                "nccl", rank=rank, world_size=world,
                device_id=torch.device("cuda", local_rank))

    # a step = cps consecutive centers of the one fit
    cps = max(1, args.centers // args.steps)
    centers_total = cps * args.steps
    warm_centers = min(cps * args.warmup, centers_total)
    data_lo, lo, n_local = rank_frames(args, rank, world)
    n_stream = stream_length(args, rank, world)
    n_total = args.frames if (world == 1 or args.scaling == "strong") \
        else args.frames * world
    if centers_total > n_total:
        raise SystemExit("more centers than frames")

    # ---- setup: synthetic frames -> HBM (centred, frame-minor) ------------
    t0 = time.perf_counter()
    if args.data == "walk":
        if world > 1:
            raise SystemExit("--data walk is a one-GPU probe")
        from enspara_amd import synth
        x = synth.walk(n_local, args.atoms, args.seed)
    else:
        x = make_shard(args, data_lo, n_local, n_stream)
    t_gen = time.perf_counter() - t0
    tstream = torch.cuda.Stream(device=local_rank) if use_dist else None
    stream = tstream.cuda_stream if use_dist else None
    t0 = time.perf_counter()
    store = FrameStore(n_local, args.atoms, device=local_rank,
                       global_offset=lo, stream=stream)
    store.sync()
    t_alloc = time.perf_counter() - t0
    # the frames go up twice: the first load also pins its two host buffers and
    # allocates the device staging (what a process pays once), the second is the
    # rate a fit's upload runs at after that
    t0 = time.perf_counter()
    store.load(x)
    store.sync()
    t_load_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    store.load(x)
    store.sync()
    t_load = time.perf_counter() - t0
    store.set_frames_per_lane(args.fpl)
    store.set_option("candidates", args.candidates)
    store.set_option("triangle", args.triangle)
    cands = store.round_candidates

    shard = sharded.DeviceShard(store) if use_dist else None
    transport = None
    if use_dist:
        transport = "gather"
        if args.transport == "mailbox" and cands > 1:
            # every rank's mailbox mapped into every other (hipIpc); collective,
            # the same verdict on every rank (no peer access anywhere: one
            # all-gather per round instead)
            if sharded.connect_mailboxes(shard):
                transport = "mailbox"
            else:
                print("mailboxes unavailable (%s): gather transport"
                      % shard.ms_connect_error, file=sys.stderr)
        if transport == "mailbox" and world > 1:
            # a short fit through the mailboxes before anything is timed: if the
            # peers' stores do not arrive on this node (no peer access after
            # all), every rank falls back to the all-gather transport together
            ok = 1
            try:
                store.reset_state()
                with torch.cuda.stream(tstream):
                    sharded.kcenters_sharded(shard, 0, min(64, args.centers), 0.0,
                                             fresh=True)
            except Exception as e:
                print("mailbox transport failed (%s): gather transport" % e,
                      file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                transport, shard.ms_connected = "gather", 0

    def run(count):
        """`count` centers from the untouched state"""
        store.reset_state()
        if use_dist:
            with torch.cuda.stream(tstream):
                idx, _ = sharded.kcenters_sharded(shard, 0, count, 0.0,
                                                  fresh=True)
        else:
            store.sync()
            idx, _, _ = store.kcenters_run(0, count, 0.0)
        return idx

    # ---- setup, untimed: the GPU has sat idle while the host generated the frames
    # (~20 s); a throw-away fit brings clocks and caches to their working state
    # before the W warm-up steps the contract asks for
    t0 = time.perf_counter()
    run(min(centers_total, 2000))
    t_wake = time.perf_counter() - t0

    # ---- warm-up: W steps of a throw-away fit ----------------------------------
    if warm_centers:
        run(warm_centers)

    # ---- timed region: the whole fit, --steps steps of cps centers --------------
    store.reset_state()
    store.sync()
    store.timing_begin(sample_every=2, max_samples=1024)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if use_dist:
        with torch.cuda.stream(tstream):
            idx, _ = sharded.kcenters_sharded(shard, 0, centers_total, 0.0,
                                              fresh=True)
    else:
        idx, _, _ = store.kcenters_run(0, centers_total, 0.0)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if len(idx) != centers_total:
        raise SystemExit("only %d of %d centers" % (len(idx), centers_total))
    kern_ms, n_samp = store.timing_end()
    timed_form = store.timing_form()
    ti_report = None
    if args.triangle:
        # (tile, candidate) pairs of 256 frames each the rounds of 16 / 32 looked at
        # and left out (a tile is left out only if NO candidate can change it)
        looked, left_out = store.ti_stats()
        ti_report = {"tile_candidate_pairs": looked, "left_out": left_out,
                     "pairs_evaluated_in_those_rounds":
                         float(looked - left_out) * 256.0}
    if use_dist:
        # (the mailbox loop moves between rounds of 8 and of 16 and counts both;
        # the gather loop runs one form)
        mix = {T: pc for T, pc in store.run_stats().items() if pc[0]} \
            if (transport == "mailbox" and cands > 1) else {}
        if not mix:
            rounds = store.spec_rounds() if cands > 1 else centers_total
            mix = {cands: (rounds, centers_total)}
        rounds = sum(p * (2 if T == 32 else 1) for T, (p, _) in mix.items())
    else:
        mix = {T: pc for T, pc in store.run_stats().items() if pc[0]} \
            if cands > 1 else {1: (centers_total, centers_total)}
        # (passes over the frames: a round of 32 is two)
        rounds = sum(p * (2 if T == 32 else 1) for T, (p, _) in mix.items())

    # ---- N > 1: what every rank spent where, so that a scaling curve comes with
    # its cause attached (round-4 review): the sampled pass / chain (its exchange's
    # wait inside) / plan kernel times of this rank's rounds, how long it waited for
    # its peers' messages, exchanges without a pass, the transport really used and
    # which devices can reach which (mailboxes need peer access)
    per_rank = None
    if use_dist:
        mine = {"rank": rank, "device": local_rank, "frames": n_local,
                "elapsed_s": elapsed, "transport": transport,
                "rounds_by_candidates": {str(T): {"rounds": p_, "centers": k_}
                                         for T, (p_, k_) in sorted(mix.items())},
                "pass_kernel_ms_by_events": kern_ms}
        if transport == "mailbox" and cands > 1:
            dg = store.ms_diag()
            ex = max(dg["exchanges"], 1)
            mine.update({
                "exchanges": dg["exchanges"], "exchanges_without_a_pass": dg["reoffers"],
                "wait_for_peers_us_per_exchange": dg["wait_peers_us"] / ex,
                "wait_for_own_flag_us_per_exchange": dg["wait_own_flag_us"] / ex,
                "sampled_round_us": {"pass": dg["pass_us"],
                                     "chain_with_exchange": dg["chain_with_exchange_us"],
                                     "plan": dg["plan_us"],
                                     "rounds_sampled": dg["rounds_sampled"]}})
        try:
            ndev = torch.cuda.device_count()
            mine["can_access_peer"] = [bool(torch.cuda.can_device_access_peer(local_rank, d))
                                       if d != local_rank else True for d in range(ndev)]
        except Exception as e:      # (not every build exposes it)
            mine["can_access_peer"] = "unknown: %s" % e
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs = float(n_total) * centers_total
    value = pairs / elapsed
    bpp = bytes_per_pair(args.atoms)
    # the dominant kernel: the form most of the sampled rounds ran.  A round of
    # 32 candidates is TWO launches of the 16-candidate kernel (candidates 0..15,
    # then 16..31 against the state the first left): the sampled duration spans
    # both, the roofline is quoted per launch (half of it, half the work).
    dom = timed_form if timed_form >= 1 else (cands if cands > 1 else 1)
    launches_per_round = 2 if dom == 32 else 1
    round_ms = kern_ms
    kern_ms = kern_ms / launches_per_round
    if dom == 32:
        # first launch: the 16-form's bytes; second: coordinates, trace,
        # distance read, 16 kept distances + the mask word
        launch_bytes = n_local * (bytes_per_frame_pass(args.atoms, 16) +
                                  12 * args.atoms + 12 + 4 * 16) / 2.0
    else:
        launch_bytes = n_local * (bytes_per_frame_pass(args.atoms, dom)
                                  if dom > 1 else bpp)
    achieved = (launch_bytes / (kern_ms * 1e-3)) / 1e9 if kern_ms > 0 else None
    # 16 candidates: a dense contraction, 18 A flop per frame x candidate pair
    launch_flops = float(n_local) * min(dom, 16) * 18 * args.atoms
    tflops = (launch_flops / (kern_ms * 1e-3)) / 1e12 if kern_ms > 0 else None
    traffic, traffic_src = load_traffic(args, n_local, dom)
    ceiling = copy_ceiling(min(x.nbytes, 4 << 30), local_rank)
    if dom == 32:
        kernel_name = "ek_pass16_kernel<true, 1> + <true, 2>"
    elif dom == 16:
        kernel_name = "ek_pass16_kernel<true, 0>"
    elif dom > 1:
        kernel_name = "ek_pass2_kernel<%d, true, true>" % dom
    else:
        kernel_name = "ek_step_kernel<FPL,0,NT>"
    hbm = {
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
        "measured_copy_GBps": ceiling[0],
        "measured_read_stream_GBps": ceiling[1],
        "frac_of_measured_read_stream": (achieved / ceiling[1])
                                        if achieved and ceiling[1] else None,
        "algorithmic_bytes_per_launch": launch_bytes,
    }
    if dom >= 16:
        roof = {"bound": "mfma", "kernel": kernel_name, "achieved": tflops,
                "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (tflops / MFMA_F32_PEAK_TFLOPS) if tflops else None,
                "algorithmic_flops_per_launch": launch_flops,
                "why": "16 candidates per frame read: 24 flop per byte, the f32 "
                       "matrix pipe (86 GFLOP per pass) and HBM (3.7 GB) are "
                       "both ~60-70 % busy, and with both the chip is at its power "
                       "limit: it holds ~2.3 GHz under this kernel, not 2.4 "
                       "(DESIGN.md 4a)",
                "hbm": hbm}
    else:
        roof = dict(hbm)
        roof.update({"bound": "hbm", "kernel": kernel_name})
    if dom >= 16:
        # (the driver's parser keeps top-level keys of `roofline`: the HBM side
        # of the dominant kernel beside its matrix-core side)
        roof.update({"hbm_frac": hbm["frac"], "hbm_achieved_GBps": hbm["achieved"],
                     "hbm_algorithmic_bytes_per_launch": launch_bytes})
    roof.update({
        "bytes_per_pair_one_center_pass": bpp,
        "pairs_per_launch": n_local * min(dom, 16),
        "avg_launch_ms": kern_ms,
        "launches_per_round": launches_per_round,
        "avg_round_stream_ms": round_ms,
        "launches_sampled": n_samp,
        "traffic": traffic,
        "traffic_source": traffic_src,
    })

    out = {
        "metric": "frame x center RMSD pairs/sec in k-centers assign",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if (world > 1 and args.scaling == "weak")
                   else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "k-centers RMSD, %d frames x %d atoms, %d centers "
                        "(BASELINE.json configs[1]), whole fit timed, "
                        "%d GPU(s)" % (n_total, args.atoms, centers_total,
                                       world),
            "frames_per_gpu": n_local, "frames_total": n_total,
            "atoms": args.atoms, "centers": centers_total,
            "centers_per_step": cps,
            "warmup_centers": warm_centers,
            "candidates_per_pass": cands,
            "passes_by_candidates": {str(T): {"passes": p * (2 if T == 32 else 1),
                                              "rounds": p, "centers": k}
                                     for T, (p, k) in sorted(mix.items())},
            "algorithm": "k-centers, up to %d candidate centers per pass over "
                         "the frames (1, 8, 16 or 32 by measured centers/ms; a "
                         "round of 32 streams the frames twice), "
                         "results identical to one pass per center" % cands
                         if cands > 1 else
                         "k-centers, one pass over the frames per center",
            "templates": args.templates if args.data == "templates" else None,
            "data": args.data, "seed": args.seed,
            "world_size": (dist.get_world_size() if use_dist else 1),
            **({"one_device_check": "EK_BENCH_ONE_DEVICE=1: every rank on device 0, gloo "
                                    "rendezvous -- the N > 1 code path exercised on a "
                                    "one-GPU box, not a measurement"} if one_device else {}),
            "sharding": ("contiguous frame blocks; per round of ~15 centers ONE "
                         "message per rank (per-prefix maxima + its farthest "
                         "frames as records), transport: %s" % transport)
                        if use_dist else "single shard",
        },
        "roofline": roof,
        "passes_over_frames": rounds,
        "centers_per_pass": centers_total / rounds if rounds else None,
        "pairs_computed": float(n_total) * sum(T * p for T, (p, _) in
                                               mix.items()),
        "triangle_inequality": ti_report,
        "per_rank": per_rank,
        "setup": {"synth_s": t_gen, "context_and_hbm_allocation_s": t_alloc,
                  "upload_center_layout_s": t_load,
                  "upload_center_layout_first_s": t_load_first,
                  "device_wake_fit_s": t_wake,
                  "host_to_hbm_GBps": x.nbytes / t_load / 1e9,
                  "host_to_hbm_first_GBps": x.nbytes / t_load_first / 1e9,
                  "how": "pageable numpy array -> two pinned 256 MiB buffers "
                         "(8 host threads) -> DMA -> centring + three layouts "
                         "on the device, double-buffered (csrc/ek_api.hip "
                         "ek_load_frames); PCIe-inclusive, never part of value"},
    }

    # ---- SURVEY.md 8(d) / BASELINE.md: the one-center pass against HBM ---------
    if world == 1 and not use_dist:
        out["roofline_one_center"] = one_center_leg(store, args, n_local)
        out["roofline"]["one_center_hbm_frac"] = out["roofline_one_center"]["frac"]

    # ---- k-hybrid refinement (configs[2]), outside the timed region -----------
    if args.pam_sweeps is None:
        args.pam_sweeps = 1 if world == 1 else 0
    if args.pam_sweeps > 0:
        # Each run: the fit again from the untouched state (untimed; also puts
        # the GPU back at its working clocks after the idle seconds of the CPU
        # cross-check -- thousands of 10 us kernels do not), then the timed sweeps.
        # Every run starts from the same state and makes the same proposals.
        runs = []
        from enspara_amd.cluster import kmedoids as km     # (not inside the timing)
        start = None
        for _ in range(max(1, args.pam_runs)):
            med = [int(i) for i in run(centers_total)]
            if start is None and world == 1:
                d0, a0 = store.download_state()
                start = {"medoids": list(med), "dist": d0, "assign": a0}
            rs = np.random.RandomState(args.seed)
            base = store.pam_prefetch_stats() + store.pam_prefetch_passes()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter()
            if use_dist:
                with torch.cuda.stream(tstream):
                    for _ in range(args.pam_sweeps):
                        med = sharded.pam_sweep_sharded(shard, med,
                                                        random_state=rs)
            else:
                for _ in range(args.pam_sweeps):
                    med = km._pam_sweep_device(store, med, None, rs)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            runs.append((time.perf_counter() - t0) / args.pam_sweeps)
        t_pam = min(runs) * args.pam_sweeps
        hits, misses, pf_restricted, pf_full = (
            int(v - b) for v, b in zip(store.pam_prefetch_stats() +
                                       store.pam_prefetch_passes(), base))
        out["khybrid"] = {
            "workload": "PAM sweeps over the %d centers of the run above "
                        "(k-hybrid = k-centers + k-medoids, BASELINE.json "
                        "configs[2])" % len(med),
            "sweeps": args.pam_sweeps,
            "s_per_sweep": t_pam / args.pam_sweeps,
            "s_per_sweep_runs": runs,
            "ms_per_proposal": t_pam / args.pam_sweeps / len(med) * 1e3,
            "proposals_per_window": km_width(use_dist),
            "prefetched_proposals_used": hits,
            "proposals_with_own_pass": misses,
            "prefetch_passes_over_touched_frames_only": pf_restricted,
            "prefetch_passes_over_all_frames": pf_full,
            # (windows worked through by one workgroup, no launch
            # between two proposals; the rest take three launches per proposal)
            "windows_in_one_workgroup": store.pam_sparse_stats()[0] if world == 1 else None,
            "of_them_ended_early": store.pam_sparse_stats()[1] if world == 1 else None,
        }
        # ---- configs[2] end to end: the fit + KHybrid's five default sweeps
        # (hybrid.py:65), frames resident; then KHybrid.fit itself from the host
        # array (upload, centring, result arrays and the centers' list included)
        if world == 1:
            n_it = 5
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            med5 = [int(i) for i in run(centers_total)]
            t_fit = time.perf_counter() - t0
            rs5 = np.random.RandomState(args.seed)
            per = []
            for _ in range(n_it):
                t1 = time.perf_counter()
                med5 = km._pam_sweep_device(store, med5, None, rs5)
                per.append(time.perf_counter() - t1)
            store.sync()
            t_all = time.perf_counter() - t0
            from enspara_amd.cluster import KHybrid
            est = KHybrid("rmsd", n_clusters=centers_total, kmedoids_updates=n_it,
                          random_state=np.random.RandomState(args.seed),
                          device=local_rank).fit(x)
            out["khybrid"]["five_sweeps"] = {
                "what": "KHybrid('rmsd', n_clusters=%d).fit: k-centers + %d PAM sweeps "
                        "(the reference's default kmedoids_updates, hybrid.py:65)"
                        % (centers_total, n_it),
                "frames_resident_total_s": t_all, "kcenters_s": t_fit,
                "sweep_s": per,
                "pairs_per_s_of_the_kcenters_part_alone": pairs / t_fit,
                "estimator_fit_from_host_array_s": est.runtime_,
                "estimator_medoids_equal_resident_run": [int(i) for i in
                                                         est.center_indices_] == med5,
                "medoids_moved_by_the_sweeps": sum(
                    1 for a_, b_ in zip(med5, start["medoids"]) if a_ != b_)}
            del est
        second = None
        if (start is not None and args.pam_sweeps == 1 and not args.no_cpu_baseline
                and args.cpu_seconds <= 0):
            # two consecutive sweeps on the GPU from the fit's state; the first one's
            # state stays on `store` for the check, the second's is handed over
            med = [int(i) for i in run(centers_total)]
            rs = np.random.RandomState(args.seed)
            med = km._pam_sweep_device(store, med, None, rs)
            d1, a1 = store.download_state()
            med2 = km._pam_sweep_device(store, list(med), None, rs)
            d2, a2 = store.download_state()
            store.upload_state(d1, a1)
            second = {"medoids": med2, "dist": d2, "assign": a2}
        if start is not None and args.pam_sweeps == 1 and not args.no_cpu_baseline:
            if second is None and world == 1:
                # (`store` must hold the state of ONE sweep for the sampled check)
                med = [int(i) for i in run(centers_total)]
                med = km._pam_sweep_device(store, med, None,
                                           np.random.RandomState(args.seed))
            out["khybrid"]["parity"] = khybrid_check(x, store, start, med, args.seed,
                                                     min(40.0, 3 * args.cpu_seconds),
                                                     second_sweep=second)
    gpu_state = start if (args.pam_sweeps > 0 and world == 1) else None
    if gpu_state is None and world == 1 and args.cpu_seconds <= 0 \
            and not args.no_cpu_baseline:
        run(centers_total)
        d0, a0 = store.download_state()
        gpu_state = {"dist": d0, "assign": a0}

    if rank == 0 and world == 1 and not args.no_msm:
        out["msm"] = msm_block(args)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(x, idx, args.cpu_seconds, gpu_state)
    elif rank == 0:
        out["cpu_baseline"] = None

    store.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
