#!/usr/bin/env python3
"""bench.py -- frame x center RMSD pairs/s of the k-centers hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one k-centers iteration: one new center, its RMSD to every frame of
the shard, the strict-< update and the farthest-point reduction, i.e.
n_frames RMSD pairs per GPU.  Workload at N=1 is BASELINE.json configs[1]:
1,000,000 synthetic frames x 300 atoms, 5000 centers (steps default to 5000,
the whole fit).  For N>1 (launched by torch.distributed.run, one rank per
GPU, RCCL) every rank holds its own 1,000,000-frame shard (weak scaling).

By default the frames are streamed once per ROUND against --candidates (8)
candidate centers and further centers are accepted from the stored distances
while the farthest point is one of them (csrc/ek_spec.hip): the
same sequential algorithm and bit-identical results with fewer passes over
HBM.  `--candidates 1` runs one pass per center (the HBM roofline case of
BASELINE.md).  Every reported pair is a distance that was computed; guesses
that were never used are not counted ("pairs_computed" has the total).

Inputs are resident in HBM (already centred and laid out frame-minor) when the
timed region starts; generation, upload and layout are reported separately in
"setup".  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E nominal (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBS = 6290.0  # measured float4 copy on MI355X (same guide)


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
    p.add_argument("--steps", type=int, default=5000)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--frames", type=int, default=1_000_000,
                   help="frames per GPU (weak scaling)")
    p.add_argument("--atoms", type=int, default=300)
    p.add_argument("--templates", type=int, default=5000)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--fpl", type=int, default=0,
                   help="frames per lane of the one-center kernel (0 = auto)")
    p.add_argument("--candidates", type=int, default=-1,
                   help="candidate centers per pass: -1 auto (8), 1, 4 or 8")
    p.add_argument("--cpu-seconds", type=float, default=15.0,
                   help="time budget of the CPU baseline leg (rank 0, N=1)")
    p.add_argument("--no-cpu-baseline", action="store_true")
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


def make_shard(args, rank):
    """This rank's frames, float32 [frames, atoms, 3], and its global offset.
    Shards start on synth.CHUNK boundaries so the data of rank r does not
    depend on the number of ranks."""
    from enspara_amd import synth
    chunks_per_rank = (args.frames + synth.CHUNK - 1) // synth.CHUNK
    first = rank * chunks_per_rank * synth.CHUNK
    x = synth.synth(args.frames, args.atoms, args.templates, args.seed,
                    first_frame=first)
    return x


def cpu_baseline(x, gpu_centers, seconds):
    """Time the CPU oracle (oracle/qcp_oracle.c: OpenMP, frames centred once,
    frame-minor tiles, AVX2 FMA across frames) on the same frames for as many
    leading k-centers iterations as fit the time budget, and check the GPU's
    centers for those iterations against it.  The oracle is the checker and
    the reported baseline, never the product path."""
    from oracle import qcp
    t0 = time.perf_counter()
    P = qcp.Prepared(x)
    _ = P.tiled
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
        if time.perf_counter() - t0 > seconds:
            break
    wall = time.perf_counter() - t0
    k = len(centers)
    ok = [int(c) for c in gpu_centers[:k]] == centers
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
        "centers_match_gpu": bool(ok),
    }
    if one:
        out["one_thread"] = one
    return out


def km_width():
    from enspara_amd.cluster import kmedoids as km
    return int(km.PAM_PREFETCH)


def load_traffic(args, cands):
    """HBM bytes per distance-kernel launch from committed rocprofv3 --pmc
    runs of this same command (profiles/traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
        key = "cands%d" % cands
        t = t.get(key, t if cands == 1 else {})
        if (t.get("frames") == args.frames and t.get("atoms") == args.atoms):
            return t.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    args = parse()
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
    if args.gpus > 1 and world == 1:
        raise SystemExit(
            "launch N>1 with: python -m torch.distributed.run --nnodes=1 "
            "--nproc-per-node N --master-addr 127.0.0.1 --master-port P "
            "bench.py --gpus N ...")

    import torch
    import torch.distributed as dist
    from enspara_amd.device import FrameStore
    from enspara_amd import sharded

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU path")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.sharded
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(
            "nccl", rank=rank, world_size=world,
            device_id=torch.device("cuda", local_rank))

    if args.warmup + args.steps > args.frames:
        raise SystemExit("warmup + steps exceeds the number of frames")

    # ---- setup: synthetic frames -> HBM (centred, frame-minor) ------------
    t0 = time.perf_counter()
    x = make_shard(args, rank)
    t_gen = time.perf_counter() - t0
    n_local = x.shape[0]
    offset = rank * n_local
    tstream = torch.cuda.Stream(device=local_rank) if use_dist else None
    stream = tstream.cuda_stream if use_dist else None
    t0 = time.perf_counter()
    store = FrameStore(n_local, args.atoms, device=local_rank,
                       global_offset=offset, stream=stream)
    store.load(x)
    store.sync()
    t_load = time.perf_counter() - t0
    store.set_frames_per_lane(args.fpl)
    store.set_option(4, args.candidates)
    cands = store.candidates
    store.reset_state()

    shard = sharded.DeviceShard(store) if use_dist else None

    def run(first_label, count, fresh):
        if use_dist:
            with torch.cuda.stream(tstream):
                idx, _ = sharded.kcenters_sharded(shard, first_label, count,
                                                  0.0, fresh=fresh)
        else:
            idx, _, _ = store.kcenters_run(first_label, count, 0.0)
        return idx

    # ---- warmup ---------------------------------------------------------------
    warm_idx = run(0, args.warmup, True)

    # ---- timed region: exactly --steps iterations ---------------------------
    store.timing_begin(sample_every=max(1, args.steps // (64 * max(cands, 4))),
                       max_samples=512)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx = run(args.warmup, args.steps, False)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if len(idx) != args.steps:
        raise SystemExit("only %d of %d steps ran" % (len(idx), args.steps))
    kern_ms, n_samp = store.timing_end()
    rounds = store.spec_rounds() if cands > 1 else args.steps

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        n_total = int(tot.item())
    else:
        n_total = n_local

    pairs = float(n_total) * args.steps
    value = pairs / elapsed
    bpp = bytes_per_pair(args.atoms)
    launch_bytes = n_local * (bytes_per_frame_pass(args.atoms, cands)
                              if cands > 1 else bpp)
    achieved = (launch_bytes / (kern_ms * 1e-3)) / 1e9 if kern_ms > 0 else None

    out = {
        "metric": "frame x center RMSD pairs/sec in k-centers assign",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "k-centers RMSD, %d frames x %d atoms per GPU, "
                        "%d centers (BASELINE.json configs[1]), %d GPU(s)"
                        % (n_local, args.atoms, args.warmup + args.steps,
                           world),
            "frames_per_gpu": n_local, "frames_total": n_total,
            "atoms": args.atoms, "centers": args.warmup + args.steps,
            "candidates_per_pass": cands,
            "algorithm": ("k-centers, %d candidate centers per pass over the "
                          "frames, results identical to one pass per center"
                          % cands) if cands > 1 else
                         "k-centers, one pass over the frames per center",
            "templates": args.templates, "seed": args.seed,
            "sharding": ("contiguous frame blocks; per round of ~7 centers: "
                         "all-gather of 8 candidate records + 320 B + 128 B "
                         "per rank") if use_dist else "single shard",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("ek_pass_kernel<%d>" % cands) if cands > 1
                      else "ek_step_kernel<FPL,0,NT>",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
            "frac_of_measured_copy_ceiling":
                (achieved / HBM_COPY_CEILING_GBS) if achieved else None,
            "algorithmic_bytes_per_launch": launch_bytes,
            "bytes_per_pair_one_center_pass": bpp,
            "pairs_per_launch": n_local * cands,
            "avg_launch_ms": kern_ms,
            "launches_sampled": n_samp,
            "traffic": load_traffic(args, cands),
        },
        "passes_over_frames": rounds,
        "centers_per_pass": args.steps / rounds if rounds else None,
        "pairs_computed": float(n_total) * rounds * cands,
        "setup": {"synth_s": t_gen, "upload_center_layout_s": t_load,
                  "host_to_hbm_GBps": x.nbytes / t_load / 1e9},
    }

    # ---- k-hybrid refinement (configs[2]), outside the timed region -----------
    if args.pam_sweeps is None:
        args.pam_sweeps = 1 if world == 1 else 0
    if args.pam_sweeps > 0:
        med = [int(i) for i in np.concatenate([warm_idx, idx])]
        rs = np.random.RandomState(args.seed)
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
            from enspara_amd.cluster import kmedoids as km
            for _ in range(args.pam_sweeps):
                med = km._pam_sweep_device(store, med, None, rs)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t_pam = time.perf_counter() - t0
        hits, misses = store.pam_prefetch_stats()
        pf_restricted, pf_full = store.pam_prefetch_passes()
        out["khybrid"] = {
            "workload": "PAM sweeps over the %d centers of the run above "
                        "(k-hybrid = k-centers + k-medoids, BASELINE.json "
                        "configs[2])" % len(med),
            "sweeps": args.pam_sweeps,
            "s_per_sweep": t_pam / args.pam_sweeps,
            "ms_per_proposal": t_pam / args.pam_sweeps / len(med) * 1e3,
            "proposals_per_pass_over_frames": km_width(),
            "prefetched_proposals_used": hits,
            "proposals_with_own_pass": misses,
            "prefetch_passes_over_touched_frames_only": pf_restricted,
            "prefetch_passes_over_all_frames": pf_full,
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        all_idx = np.concatenate([warm_idx, idx])
        out["cpu_baseline"] = cpu_baseline(x, all_idx, args.cpu_seconds)
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
