"""Rounds across shards with one exchange each (csrc/ek_mshard.hip) on ONE GPU:
  ms_probe.py <n> <atoms> <centers> [shards=1] [cands=16] [reps=2]
`shards` contexts of one process share the GPU (own stream each), connected by
peer mailboxes (plain addresses); each runs ek_ms_run from its own host thread.
Prints seconds per run, rounds, microseconds per accepted center, and compares
centers / labels / distances with the single-shard ek_kcenters_run of the same
data (which the GPU tests compare with the oracle)."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # shards that wait for one another: a HW queue each
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from enspara_amd import sharded, synth
from enspara_amd.device import FrameStore

n, A, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
T = int(sys.argv[5]) if len(sys.argv) > 5 else 16
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
tm = synth.templates(int(os.environ.get("MS_TEMPLATES", "5000")), A, 1)
x = np.concatenate([synth.synth_chunk(c, min(synth.CHUNK, n - c * synth.CHUNK), tm, 1)
                    for c in range((n + synth.CHUNK - 1) // synth.CHUNK)])

ref = FrameStore.from_array(x)
ref.set_option(4, T)
ref.set_option(8, 0)
ref.reset_state()
t0 = time.perf_counter()
ridx, rcd, rmx = ref.kcenters_run(0, K, 0.0)
t_ref = time.perf_counter() - t0
ref.reset_state()
t0 = time.perf_counter()
ridx, rcd, rmx = ref.kcenters_run(0, K, 0.0)
t_ref = time.perf_counter() - t0
rd, ra = ref.download_state()
rstats = {k: v for k, v in ref.run_stats().items() if v[0]}
ref.close()
print("single shard, %d candidates: %.4f s  %.2f us/center  %s" % (T, t_ref, t_ref / K * 1e6, rstats), flush=True)

stores = []
for r in range(S):
    lo, cnt = sharded.shard_bounds(n, S, r)
    st = FrameStore(cnt, A, device=0, global_offset=lo)
    st.load(x[lo:lo + cnt])
    st.set_option(4, T)
    st.set_option("pass_sweep", int(os.environ.get("MS_SWEEP", "1")))
    st.set_option("ms_two_phase", int(os.environ.get("MS_TWO_PHASE", "1")))
    st.set_option(18, int(os.environ.get("MS_SMALL", 1 if cnt < 300000 else 0)))     # (what sharded.kcenters_sharded sets)
    st.ms_setup(S, r)
    st.reserve_centers(K)
    stores.append(st)
boxes = [st.ms_mailbox() for st in stores]
for st in stores:
    for p in range(S):
        st.ms_connect(p, boxes[p][0], boxes[p][1])
out = [None] * S


def work(r):
    try:
        out[r] = stores[r].ms_run(0, K, 0.0)
    except Exception as e:
        print("shard %d: %s" % (r, e), flush=True)
        out[r] = (np.zeros(0, np.int64), np.zeros(0, np.float32), 0.0)


for rep in range(reps):
    for st in stores:
        st.reset_state()
        st.reset_history()
        st.sync()
    th = [threading.Thread(target=work, args=(r,)) for r in range(S)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    mix = {k: v for k, v in stores[0].run_stats().items() if v[0]}
    rounds = sum(v[0] for v in mix.values())
    print("   passes by candidates:", mix, flush=True)
    print("   exchanges per shard:", [st.ms_state() for st in stores], " without a pass:",
          [st.ms_diag()["reoffers"] for st in stores], flush=True)
    print("%d shard(s) of %d frames, mailboxes: %.4f s  %d passes  %.2f centers/pass  %.2f us/center  %.1f us/round"
          % (S, stores[0].n, dt, rounds, K / max(rounds, 1), dt / K * 1e6, dt / max(rounds, 1) * 1e6), flush=True)
ok = all(np.array_equal(o[0], ridx) and np.array_equal(o[1], rcd) for o in out)
d = np.concatenate([st.download_state()[0] for st in stores])
a = np.concatenate([st.download_state()[1] for st in stores])
print("centers equal: %s  distances equal: %s  labels equal: %s  final max %r / %r"
      % (ok, np.array_equal(d, rd), np.array_equal(a, ra), out[0][2], rmx), flush=True)
# (a measurement build, -DEK_MS_STAMPS: where the chain and plan kernels' single workgroups
# spend their time; ENSPARA_HIP_LIB names the variant library)
import ctypes as C
from enspara_amd import _lib
try:
    fn = _lib.load().ek_ms_stamps
except AttributeError:
    fn = None
if fn is not None:
    mean = (C.c_double * 32)()
    cnt = (C.c_int64 * 32)()
    fn(mean, cnt, 1)
    names = {0: "chain tail: per-prefix maxima reduced", 1: "chain tail: headers out",
             2: "chain tail: look at the flags (+ walk)", 3: "chain tail: pick(s) (+ wait, walk)",
             4: "chain tail: list + head published", 5: "last helper: flags out, peers' flags seen",
             6: "helper 0: start -> go-ahead", 7: "helper 0: records written",
             8: "plan wg 0: counts, values, ranks", 9: "plan wg 0: its pairs",
             10: "plan last wg: D into LDS, values, maxima", 11: "plan last wg: decision",
             12: "plan last wg: greedy choice", 13: "plan last wg: records, plan"}
    for k in sorted(names):
        print("   %-48s %7.2f us  x %d" % (names[k], mean[k], cnt[k]), flush=True)
for st in stores:
    st.close()
