"""How far ahead can a round's guesses hold?  (CPU study, round 5.)

The true k-centers sequence on a scaled copy of the bench's data (one template
per center, 200 frames per template), and at every `stride`-th state the
device's candidate pick restated in numpy: the per-256-frame block maxima, at
most PER of them per current label, the M largest = the list; greedy order on
the list with exact pairwise distances; count how many of the next true centers
the greedy order predicts before its first miss, for lists of 64 / 128 / 256
and rounds of 16 / 32.  Prints the mean accepted per round (1 + hits)."""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from enspara_amd import synth            # noqa: E402
from oracle import qcp                   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
A = int(sys.argv[2]) if len(sys.argv) > 2 else 30
K = int(sys.argv[3]) if len(sys.argv) > 3 else 500
tmpl = int(sys.argv[4]) if len(sys.argv) > 4 else K
x = synth.synth(n, A, tmpl, 1)
c, G = qcp.center_and_trace(x)
dist = np.full(n, np.inf, dtype=np.float32)
assign = np.zeros(n, dtype=np.int64)
centers = []
states = []
for k in range(K):
    i = 0 if k == 0 else int(np.argmax(dist))
    centers.append(i)
    d = qcp.rmsd_centered(c, G, c[i], G[i])
    upd = d < dist
    dist[upd] = d[upd]
    assign[upd] = k
    states.append((dist.copy(), assign.copy()))
print("fit done", flush=True)
nb = (n + 255) // 256


def pick(dist, assign, M, PER, B=256):
    nb = (n + B - 1) // B
    pad = np.full(nb * B, -np.inf, dtype=np.float32)
    pad[:n] = dist
    blk = pad.reshape(nb, B)
    bi = blk.argmax(1) + np.arange(nb) * B
    bv = blk.max(1)
    order = np.lexsort((bi, -bv))
    taken = {}
    lst = []
    for o in order:
        lab = assign[bi[o]]
        if taken.get(lab, 0) >= PER:
            continue
        taken[lab] = taken.get(lab, 0) + 1
        lst.append(bi[o])
        if len(lst) >= M:
            break
    return np.array(lst)


def greedy(lst, dist, T):
    cur = dist[lst].copy()
    D = np.empty((len(lst), len(lst)), dtype=np.float32)
    for a, i in enumerate(lst):
        D[a] = qcp.rmsd_centered(c[lst], G[lst], c[i], G[i])
    open_ = np.ones(len(lst), bool)
    out = []
    for _ in range(T):
        if not open_.any():
            break
        v = np.where(open_, cur, -np.inf)
        b = int(np.lexsort((lst, -v))[0])
        out.append(int(lst[b]))
        open_[b] = False
        cur = np.minimum(cur, D[b])
    return out


stride = 7
for M, PER, B in ((64, 8, 64), (96, 8, 64), (128, 8, 64), (128, 16, 64), (128, 6, 64), (128, 8, 32),
                  (128, 100, 64)):
    for T in (16, 32):
        acc = []
        for k in range(40, K - T - 1, stride):
            d, a = states[k - 1]        # state after centers[0..k-1]
            lst = pick(d, a, M, PER, B)
            g = greedy(lst, d, T)
            hits = 0
            for j, gi in enumerate(g):
                if k + j < K and centers[k + j] == gi:
                    hits += 1
                else:
                    break
            acc.append(hits)
        acc = np.array(acc)
        print("block %3d list %3d per-label %d rounds of %2d: accepted %.2f mean, whole %.0f%%, "
              "hist(quartiles) %s" % (B, M, PER, T, acc.mean(), 100 * (acc == T).mean(),
                                      np.percentile(acc, [10, 25, 50, 75, 90])), flush=True)
