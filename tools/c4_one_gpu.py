"""BASELINE.json configs[3] -- k-centers, 10 M frames x 500 atoms, 20 000 centers,
frame-sharded over 8 GPUs -- with its WHOLE data set on ONE MI355X: the eight
shards are eight contexts of this process (1.25 M frames each, three layouts
each: 180 GB of the 288 GB), connected by peer mailboxes, every shard's
ek_ms_run in its own host thread -- the 8-rank message layout and protocol of
csrc/ek_mshard.hip, the GPUs' streams time-sharing one chip instead of running
side by side.  Reference: enspara/cluster/kcenters.py:314-378 (the MPI
iteration) at the shape the reference names for it.

  python3 tools/c4_one_gpu.py [--shards 8] [--frames 10000000]
        [--atoms 500] [--centers 20000] [--templates 20000]
        [--check-centers 200] [--sample 2000] [--out summary.json]

What is checked (the oracle is the checker; the frames are regenerated shard
by shard on the device -- tools/c4/c4gen.hip -- and copied to the host for it):
  1. the first --check-centers centers against the oracle's loop over ALL frames,
     shard by shard: every shard's labels and float32 distances after them, and
     every center as the first-index arg-max over the shards (lowest shard wins
     ties, kcenters.py:337) with its pre-update distance;
  2. after ALL centers, on --sample frames of every shard: the distance is, bit
     for bit, the RMSD to the center the label names; every center's own frame
     is at distance 0 under its own label; the centers are distinct; their
     pre-update distances never increase; the run's final maximum is the
     maximum of the downloaded distances.
"""
import argparse
import json
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # shards that wait for one another
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "c4"))
import numpy as np


def threads():
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--shards", type=int, default=8)
    p.add_argument("--frames", type=int, default=10_000_000,
                   help="frames in all: split into tile-aligned contiguous shards "
                        "(sharded.shard_bounds, as the ranks of a job split them)")
    p.add_argument("--frames-per-shard", type=int, default=0,
                   help="instead of --frames: this many frames per shard")
    p.add_argument("--atoms", type=int, default=500)
    p.add_argument("--centers", type=int, default=20_000)
    p.add_argument("--templates", type=int, default=20_000)
    p.add_argument("--check-centers", type=int, default=200)
    p.add_argument("--sample", type=int, default=2000)
    p.add_argument("--candidates", type=int, default=-1)
    p.add_argument("--seed", type=int, default=2)
    p.add_argument("--out", default=None)
    p.add_argument("--pick-cap", type=int, default=0)
    p.add_argument("--small-shards", type=int, default=-1)
    p.add_argument("--no-oracle", action="store_true",
                   help="the run and its size-independent properties only (debugging)")
    a = p.parse_args()

    # (ENSPARA_NO_TORCH=1: no torch in the process -- the library then binds to the
    # system's HIP runtime instead of the one torch bundles)
    if os.environ.get("ENSPARA_NO_TORCH") != "1":
        import torch  # noqa: F401
    import c4gen
    from enspara_amd.device import FrameStore
    from oracle import qcp

    from enspara_amd import sharded
    S, A, K = a.shards, a.atoms, a.centers
    n_all = a.frames_per_shard * S if a.frames_per_shard else a.frames
    n_per = sharded.shard_bounds(n_all, S, 0)[1]        # (the largest shard)
    C1 = min(a.check_centers, K)
    from enspara_amd import _lib
    _lib.load()
    hip = c4gen.Hip()
    free0, total = hip.mem_info()
    raw_bytes = n_per * A * 12
    per_shard = 3 * raw_bytes + n_per * 200
    while S > 1 and S * per_shard + raw_bytes > 0.94 * free0:
        S //= 2         # (the same shards, fewer of them: a part of the data set)
    bounds = [sharded.shard_bounds(n_all, a.shards, r) for r in range(S)]
    n = sum(cnt for _, cnt in bounds)
    report = {"what": "BASELINE.json configs[3] on one MI355X: %d shards x %d frames x %d "
                      "atoms as contexts of one process, %d centers through ek_ms_run "
                      "(peer mailboxes, the %d-rank message layout)" % (S, n_per, A, K, S),
              "shard_frames": [cnt for _, cnt in bounds],
              "shards": S, "shards_asked": a.shards, "frames": n, "atoms": A, "centers": K,
              "templates": a.templates, "seed": a.seed,
              "hbm_free_before_GB": free0 / 1e9, "hbm_total_GB": total / 1e9}
    print("HBM free %.1f GB of %.1f; %d shards x %.1f GB" %
          (free0 / 1e9, total / 1e9, S, per_shard / 1e9), flush=True)

    # ---- the frames: generated on the device, shard by shard --------------------
    t0 = time.perf_counter()
    tmpl = c4gen.templates_int(a.templates, A, a.seed)
    gen = c4gen.DeviceGenerator(tmpl, hip)
    raw = hip.malloc(n_per * A * 12)
    stores = []
    t_gen = t_load = 0.0
    for r in range(S):
        lo, cnt = bounds[r]
        st = FrameStore(cnt, A, device=0, global_offset=lo)
        t1 = time.perf_counter()
        gen.fill(raw, lo, cnt, a.seed)
        hip.sync()
        t_gen += time.perf_counter() - t1
        t1 = time.perf_counter()
        st.load_device(raw, cnt)
        st.sync()
        t_load += time.perf_counter() - t1
        st.set_option("candidates", a.candidates)
        st.set_option("small_shards", (1 if n_per < 300000 else 0)
                      if a.small_shards < 0 else a.small_shards)
        st.set_option("pick_cap", a.pick_cap)
        stores.append(st)
    for r, st in enumerate(stores):
        if a.candidates == -1 or a.candidates >= 16:
            if not st.quad_copy_ready():
                raise SystemExit("shard %d: no room for the quad copy" % r)
        st.ms_setup(S, r)
        st.reserve_centers(K)       # (no allocation -- no device-wide wait -- inside ms_run)
    boxes = [st.ms_mailbox() for st in stores]
    for st in stores:
        for q in range(S):
            st.ms_connect(q, boxes[q][0], boxes[q][1])
    free1, _ = hip.mem_info()
    report.update({"setup_s": time.perf_counter() - t0, "device_generation_s": t_gen,
                   "layout_s": t_load, "hbm_in_use_GB": (total - free1) / 1e9})
    print("setup %.1f s (generation %.2f, layouts %.2f); HBM in use %.1f GB"
          % (report["setup_s"], t_gen, t_load, report["hbm_in_use_GB"]), flush=True)

    out = [None] * S
    err = [None] * S

    def part(r, first, count):
        try:
            out[r] = stores[r].ms_run(first, count, 0.0)
        except Exception as e:      # noqa: BLE001 (reported below)
            err[r] = "%s: %s" % (type(e).__name__, e)

    def run(first, count):
        th = [threading.Thread(target=part, args=(r, first, count)) for r in range(S)]
        t1 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t1
        if any(err):
            for r_, st_ in enumerate(stores):      # what every shard knows of the run
                try:
                    print("shard %d: %s | state (mode, exchanges, err) %s | rounds %s | "
                          "centers in its history %d | diag %s"
                          % (r_, err[r_], st_.ms_state(),
                             {k: v for k, v in st_.run_stats().items() if v[0]},
                             st_.history(0, 1)[2], st_.ms_diag()), flush=True)
                except Exception as e:      # noqa: BLE001
                    print("shard %d: %s (%s)" % (r_, err[r_], e), flush=True)
            raise SystemExit("ek_ms_run failed")
        for o in out[1:]:
            if not (np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1])):
                raise SystemExit("the shards disagree on the centers")
        return dt, out[0][0].copy(), out[0][1].copy(), out[0][2]

    for st in stores:
        st.reset_state()
        st.reset_history()
        st.sync()
    # ---- phase 1: the centers the oracle replays; phase 2: the rest --------------
    dt1, idx1, cd1, _ = run(0, C1)
    state1 = [st.download_state() for st in stores]
    mix1 = {k: v for k, v in stores[0].run_stats().items() if v[0]}
    dt2, idx2, cd2, fmax = run(C1, K - C1) if K > C1 else (0.0, idx1[:0], cd1[:0], None)
    mix2 = {k: v for k, v in stores[0].run_stats().items() if v[0]} if K > C1 else {}
    centers = np.concatenate([idx1, idx2])
    cdist = np.concatenate([cd1, cd2])
    final = [st.download_state() for st in stores]
    rounds2 = sum(v[0] for v in mix2.values())
    diag = stores[0].ms_diag()
    report["run"] = {
        "centers_found": int(len(centers)),
        "first_centers_s": dt1, "first_centers": C1,
        "rest_s": dt2, "rest_centers": int(len(idx2)),
        "rounds_by_candidates_first": {str(k): list(v) for k, v in mix1.items()},
        "rounds_by_candidates_rest": {str(k): list(v) for k, v in mix2.items()},
        "us_per_center_rest": dt2 / max(len(idx2), 1) * 1e6,
        "ms_per_round_all_shards_rest": dt2 / max(rounds2, 1) * 1e3,
        "pairs_per_s_rest_one_gpu_time_shared": float(n) * len(idx2) / dt2 if dt2 else None,
        "exchanges": diag["exchanges"], "exchanges_without_a_pass": diag["reoffers"],
        "sampled_round_us_shard0": {"pass": diag["pass_us"],
                                    "chain_with_exchange": diag["chain_with_exchange_us"],
                                    "plan": diag["plan_us"]},
        "final_max_distance": fmax}
    print("phase 1: %d centers %.2f s; phase 2: %d centers %.2f s (%.1f us/center, %s)"
          % (C1, dt1, len(idx2), dt2, report["run"]["us_per_center_rest"], mix2), flush=True)
    if len(centers) != K:
        raise SystemExit("only %d of %d centers" % (len(centers), K))

    # ---- size-independent properties of the whole run ----------------------------
    allmax = max(float(d.max()) for d, _ in final)
    props = {
        "centers_distinct": bool(len(set(int(c) for c in centers)) == K),
        "first_center_is_frame_0": bool(int(centers[0]) == 0),
        "pre_update_distances_never_increase": bool(np.all(cdist[2:] <= cdist[1:-1])),
        "final_max_is_max_of_distances": bool(fmax is None or np.float32(fmax) == np.float32(allmax)),
        "labels_in_range": bool(all(int(l.min()) >= 0 and int(l.max()) < K for _, l in final)),
    }
    own_ok = True
    for k, g in enumerate(centers):
        r = max(q for q in range(S) if bounds[q][0] <= int(g))
        i = int(g) - bounds[r][0]
        d, l = final[r]
        # (rmsd(x, x) cancels to ~1e-4 nm at worst, not to 0: DESIGN.md 2; the
        # reference's own check of its medoids is < 0.001, kmedoids.py:197)
        own_ok = own_ok and d[i] < 1e-3 and l[i] == k
    props["every_center_within_1e-3_of_itself_under_its_own_label"] = bool(own_ok)
    report["properties"] = props
    print("properties:", props, flush=True)

    if a.no_oracle:
        report["ok"] = bool(all(v for v in props.values()))
        print(json.dumps(report))
        return 0 if report["ok"] else 1
    # ---- the oracle, shard by shard ----------------------------------------------
    qcp.set_num_threads(threads())
    t0 = time.perf_counter()
    ctr_raw = c4gen.frames(tmpl, a.seed, centers)
    cc, cG = qcp.center_and_trace(ctr_raw)
    rec_v = np.empty((S, C1), dtype=np.float32)
    rec_i = np.empty((S, C1), dtype=np.int64)
    chk = {"generator_sample_equals_numpy": True, "phase1_labels_equal": True,
           "phase1_distances_equal": True, "final_sample_is_rmsd_to_own_center": True,
           "sampled_frames": 0, "oracle_pairs": 0}
    rng = np.random.RandomState(11)
    for r in range(S):
        lo, cnt = bounds[r]
        gen.fill(raw, lo, cnt, a.seed)
        hip.sync()
        x = np.empty((cnt, A, 3), dtype=np.float32)
        hip.to_host(x, raw)
        pick = np.sort(rng.choice(cnt, size=min(a.sample, cnt), replace=False))
        chk["generator_sample_equals_numpy"] &= bool(np.array_equal(
            x[pick[:256]], c4gen.frames(tmpl, a.seed, lo + pick[:256])))
        P = qcp.Prepared(x)
        dist = np.full(cnt, np.inf, dtype=np.float32)
        assign = np.full(cnt, -1, dtype=np.int32)
        for k in range(C1):
            rec_v[r, k], am = P.kcenters_step(cc[k], cG[k], k, dist, assign)
            rec_i[r, k] = lo + am
        chk["oracle_pairs"] += cnt * C1
        chk["phase1_labels_equal"] &= bool(np.array_equal(assign, state1[r][1]))
        chk["phase1_distances_equal"] &= bool(np.array_equal(dist, state1[r][0]))
        d, l = final[r]
        for lab in np.unique(l[pick]):
            fr = pick[l[pick] == lab]
            want = qcp.rmsd_centered(np.ascontiguousarray(P.c[fr]),
                                     np.ascontiguousarray(P.G[fr]), cc[lab], float(cG[lab]))
            chk["final_sample_is_rmsd_to_own_center"] &= bool(np.array_equal(want, d[fr]))
        chk["sampled_frames"] += int(len(pick))
        print("shard %d checked (%.0f s)" % (r, time.perf_counter() - t0), flush=True)
        del P, x
    # every center as the first-index arg-max over the shards (lowest shard on ties)
    win = np.argmax(rec_v, axis=0)          # first maximum = lowest shard
    want_next = rec_i[win, np.arange(C1)]
    want_dist = rec_v[win, np.arange(C1)]
    upto = min(C1, K - 1)
    chk["phase1_centers_equal"] = bool(np.array_equal(want_next[:upto], centers[1:upto + 1]))
    chk["phase1_center_distances_equal"] = bool(np.array_equal(want_dist[:upto],
                                                               cdist[1:upto + 1]))
    chk["oracle_s"] = time.perf_counter() - t0
    chk["oracle_threads"] = threads()
    report["oracle"] = chk
    ok = all(v for v in props.values()) and all(
        chk[k] for k in ("generator_sample_equals_numpy", "phase1_labels_equal",
                         "phase1_distances_equal", "final_sample_is_rmsd_to_own_center",
                         "phase1_centers_equal", "phase1_center_distances_equal"))
    report["ok"] = bool(ok)
    for st in stores:
        st.close()
    line = json.dumps(report)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(line + "\n")
    print(line)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
