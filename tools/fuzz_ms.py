"""Scratch: randomized comparison with the oracle, part 4: the rounds ACROSS shards with the
exchange on the device (csrc/ek_mshard.hip, ek_ms_run) -- the shards are contexts of this one
process on the one GPU, mailboxes by address, every shard's loop in its own host thread.
Random shard counts (ragged and empty shards), frame / atom / template counts (duplicates and
ties when the templates are few), center counts or distance cut-offs, round widths (the
ladder, 8, 16, 32), the per-prefix maxima in the pass or in the chain kernel, and the exchange
in two steps (per-prefix maxima first) or in one.  Every center, label and distance against
oracle.cluster.kcenters.  usage: fuzz_ms.py [n_cases] [seed] [first_case]; FUZZ_VERBOSE=1
prints every case before it runs."""
import os
import sys
import threading
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")    # shards that wait for one another: a HW queue each
os.environ.setdefault("OMP_NUM_THREADS", "8")
sys.path.insert(0, os.environ.get("FUZZ_ROOT") or      # (another tree's package: bisecting)
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from enspara_amd import sharded, synth
from enspara_amd.device import FrameStore
from oracle import cluster as oc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = 0
t0 = time.time()
n_reoffer = n_exch = 0
for case in range(first, first + cases):
    if case and case % 50 == 0:
        print("... %d cases, %d mismatches, %.0f s" % (case, bad, time.time() - t0), flush=True)
    rng = np.random.RandomState(base * 100003 + case)
    shards = int(rng.choice([1, 2, 3, 4, 5, 8]))
    n = int(rng.choice([40, 300, 777, 2048, 5000, 12000, 30000]))
    A = int(rng.choice([1, 3, 5, 16, 33, 100]))
    nt = int(rng.choice([1, 3, 11, 40]))
    cands = int(rng.choice([-1, 8, 16, 32]))
    two = int(rng.randint(2))
    sweep = int(rng.choice([0, 1, 2]))
    if os.environ.get("FUZZ_TWO"):      # (bisecting a failure)
        two = int(os.environ["FUZZ_TWO"])
    if os.environ.get("FUZZ_CANDS"):
        cands = int(os.environ["FUZZ_CANDS"])
    if os.environ.get("FUZZ_SHARDS"):
        shards = int(os.environ["FUZZ_SHARDS"])
    if os.environ.get("FUZZ_ATOMS"):        # (e.g. "300,500": the bench shapes' atom counts)
        A = int(rng.choice([int(v) for v in os.environ["FUZZ_ATOMS"].split(",")]))
    if os.environ.get("FUZZ_SHARDS_MIN"):
        shards = max(shards, int(os.environ["FUZZ_SHARDS_MIN"]))
    if os.environ.get("FUZZ_SWEEP"):
        sweep = int(os.environ["FUZZ_SWEEP"])
    x = synth.synth(n, A, nt, seed=int(rng.randint(1 << 30)))
    if rng.randint(4) == 0:
        K, cutoff = 0, float(rng.choice([0.2, 0.5, 1.0]))
    else:
        K, cutoff = int(min(n, rng.choice([1, 2, 17, 64, 200, 700]))), 0.0
    tag = "case %d shards=%d n=%d A=%d nt=%d K=%d cutoff=%g cands=%d two=%d sweep=%d" % (
        case, shards, n, A, nt, K, cutoff, cands, two, sweep)
    if os.environ.get("FUZZ_VERBOSE"):
        print(tag, flush=True)
    if os.environ.get("FUZZ_MEM"):      # (free device memory before the case: a leak shows)
        if case == first:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "c4"))
            import c4gen
            hip = c4gen.Hip()
        print("   free %.3f GB" % (hip.mem_info()[0] / 1e9), flush=True)
    inds, wa, wd = oc.kcenters(x, n_clusters=K or None, dist_cutoff=cutoff or None)
    stores = []
    try:
        for r in range(shards):
            lo, cnt = sharded.shard_bounds(n, shards, r)
            def watch(step):        # (FUZZ_WATCH: which step of a later store spoils an earlier one's frames)
                if not os.environ.get("FUZZ_WATCH"):
                    return
                from oracle import qcp
                Pw = qcp.Prepared(x)
                for q, sq in enumerate(stores):
                    l2, c2 = sharded.shard_bounds(n, shards, q)
                    if c2 == 0:
                        continue
                    got = np.asarray(sq.rmsd_to_xyz(x[0]), dtype=np.float32)
                    want = np.asarray(qcp.rmsd_centered(np.ascontiguousarray(Pw.c[l2:l2 + c2]),
                                                        np.ascontiguousarray(Pw.G[l2:l2 + c2]),
                                                        Pw.c[0], float(Pw.G[0])), dtype=np.float32)
                    if not np.array_equal(got, want):
                        print("WATCH", tag, "store", q, "is wrong after store", r, "did", step,
                              int((got != want).sum()), "of", c2, flush=True)
                    sq.reset_state()
                    sq.sync()
            st = FrameStore(cnt, A, device=0, global_offset=lo)
            watch("create")
            st.load(x[lo:lo + cnt])
            st.sync()
            watch("load")
            st.set_option(4, cands)
            try:
                st.set_option("pass_sweep", sweep)
            except Exception:       # noqa: BLE001 (a tree from before the option)
                pass
            try:
                st.set_option("ms_two_phase", two)
            except Exception:       # noqa: BLE001 (a library from before the option, bisecting)
                pass
            st.ms_setup(shards, r)
            watch("ms_setup")
            if hasattr(st, "reserve_centers"):
                st.reserve_centers(K if K else n)
            watch("reserve_centers")
            st.reset_state()
            st.sync()
            if os.environ.get("FUZZ_CHECK_LOAD") and cnt:
                from oracle import qcp
                Pq = qcp.Prepared(x)
                got = np.asarray(st.rmsd_to_xyz(x[0]), dtype=np.float32)
                want = np.asarray(qcp.rmsd_centered(np.ascontiguousarray(Pq.c[lo:lo + cnt]),
                                                    np.ascontiguousarray(Pq.G[lo:lo + cnt]),
                                                    Pq.c[0], float(Pq.G[0])), dtype=np.float32)
                if not np.array_equal(got, want):
                    print("LOAD0", tag, "shard", r, int((got != want).sum()), "of", cnt,
                          "wrong right after its own load", flush=True)
                st.reset_state()
                st.sync()
            stores.append(st)
        if os.environ.get("FUZZ_CHECK_LOAD"):   # (are the frames on the device the caller's?)
            from oracle import qcp
            P = qcp.Prepared(x)
            for r, st in enumerate(stores):
                lo, cnt = sharded.shard_bounds(n, shards, r)
                if cnt == 0:
                    continue
                got = np.asarray(st.rmsd_to_xyz(x[0]), dtype=np.float32)
                want = qcp.rmsd_centered(np.ascontiguousarray(P.c[lo:lo + cnt]),
                                         np.ascontiguousarray(P.G[lo:lo + cnt]), P.c[0],
                                         float(P.G[0]))
                if not np.array_equal(got, np.asarray(want, dtype=np.float32)):
                    print("LOAD", tag, "shard", r, int((got != want).sum()), "of", cnt,
                          "distances to frame 0 differ right after the load", flush=True)
                st.reset_state()
                st.sync()
        boxes = [st.ms_mailbox() for st in stores]
        for st in stores:
            for p in range(shards):
                st.ms_connect(p, boxes[p][0], boxes[p][1])
        out = [None] * shards
        errs = []

        def work(r):
            try:
                out[r] = stores[r].ms_run(0, K if K else n, cutoff)
            except Exception as e:     # noqa: BLE001 (reported below)
                errs.append("shard %d: %s" % (r, e))

        th = [threading.Thread(target=work, args=(r,)) for r in range(shards)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        ok = not errs
        if ok:
            parts = [st.download_state() for st in stores]
            ok = (all(np.array_equal(o[0], np.array(inds)) for o in out)
                  and np.array_equal(np.concatenate([p[1] for p in parts]), wa)
                  and np.array_equal(np.concatenate([p[0] for p in parts]).astype(np.float64), wd)
                  and out[0][2] == np.float32(wd.max()))
            try:
                dg = stores[0].ms_diag()
                n_reoffer += dg["reoffers"]
                n_exch += dg["exchanges"]
            except Exception:       # noqa: BLE001
                pass
        if os.environ.get("EK_POISON"):
            over = [st.debug_guards() for st in stores]
            if any(over):
                print("OVERRUN", tag, over, flush=True)
                ok = False
        if not ok:
            bad += 1
            print("MISMATCH", tag, errs[:2], flush=True)
            if not errs:
                got = np.asarray(out[0][0])
                want = np.array(inds)
                m = min(len(got), len(want))
                neq = np.nonzero(got[:m] != want[:m])[0]
                print("   centers: %d found, %d expected, first difference at %s; per shard equal: %s"
                      % (len(got), len(want), neq[:1], [bool(np.array_equal(o[0], out[0][0])) for o in out]),
                      flush=True)
                a = np.concatenate([p[1] for p in parts])
                d = np.concatenate([p[0] for p in parts]).astype(np.float64)
                print("   labels differ at %d frames, distances at %d; final max %r / %r"
                      % (int((a != wa).sum()), int((d != wd).sum()), out[0][2], np.float32(wd.max())),
                      flush=True)
    finally:
        for st in stores:
            st.close()
print("%d cases, %d mismatches, %.0f s (%d exchanges, %d of them without a pass)"
      % (cases, bad, time.time() - t0, n_exch, n_reoffer), flush=True)
sys.exit(1 if bad else 0)
