"""Checker-backed shard for the multi-rank driver tests (TEST CODE).

Implements the shard protocol of enspara_amd/sharded.py on CPU with the
oracle, with the same record layout as the device (include/enspara_hip.h), so
that the product's driver loop -- collectives, winner rule, stop rule, history
-- can be exercised with the gloo backend and world_size 2 on a machine
without GPUs."""
import numpy as np
import torch

from oracle import qcp

HDR = np.dtype([("maxdist", "<f4"), ("valid", "<i4"), ("gidx", "<i8"),
                ("trace", "<f8"), ("reserved", "<i8")])
assert HDR.itemsize == 32


MAXHDR = np.dtype([("maxdist", "<f4"), ("valid", "<i4"), ("gidx", "<i8")])
assert MAXHDR.itemsize == 16


PAM_OUT = np.dtype([("sum_old", "<f8"), ("sum_new", "<f8"),
                    ("n_frames", "<i8"), ("n_amb", "<u4"), ("moved", "<u4")])


class _PamMixin:
    """Checker-backed PAM shard protocol (ek_centered_frames,
    ek_pam_begin_table, ek_pam_count/select[_batch], ek_pam_prefetch_centers,
    ek_pam_propose_center, ek_pam_commit)."""

    @property
    def n_atoms(self):
        return self.A

    @property
    def n_local(self):
        return self.n

    def host_to_buffer(self, arr):
        return torch.from_numpy(np.array(arr, dtype=np.int64))

    def new_table(self, rows):
        return (torch.zeros((rows, 3 * self.A), dtype=torch.float32),
                torch.zeros(2 * rows, dtype=torch.int64))

    @staticmethod
    def _traces(meta):
        return meta.numpy()[:len(meta) // 2].view(np.float64)

    def fill_rows(self, local_frames, rows, coords, meta):
        for f, r in zip(local_frames, rows):
            coords.numpy()[r] = self.P.c[f].ravel()
            self._traces(meta)[r] = self.P.G[f]

    def _vec(self, c, g):
        if not self.n:
            return np.zeros(0, dtype=np.float32)
        return qcp.rmsd_centered(self.P.c, self.P.G,
                                 np.ascontiguousarray(c.reshape(self.A, 3)),
                                 float(g))

    def pam_begin_table(self, coords, meta, K):
        self.med_c = coords.numpy()[:K].copy()
        self.med_G = self._traces(meta)[:K].copy()
        self.pending = None
        self.pf = []

    def pam_count_batch(self, cid0, count):
        return np.array([np.count_nonzero(self.assign == cid0 + i)
                         for i in range(count)], dtype=np.int64)

    def pam_select_batch(self, cid0, js):
        return np.array([-1 if j < 0 else
                         np.flatnonzero(self.assign == cid0 + i)[j]
                         for i, j in enumerate(js)], dtype=np.int64)

    def pam_count(self, cid):
        return int(np.count_nonzero(self.assign == cid))

    def pam_select(self, cid, j):
        return int(np.flatnonzero(self.assign == cid)[j])

    def pam_prefetch_centers(self, coords, meta, count, win_lo=0, win_count=0):
        g = self._traces(meta)
        self.pf = [self._vec(coords.numpy()[j], g[j]) for j in range(count)]

    def pam_propose_center(self, cid, slot, coords, meta, row,
                           n_members_local, win_lo, win_count, out):
        c = coords.numpy()[row].copy()
        g = float(self._traces(meta)[row])
        nd = self.pf[slot] if slot >= 0 else self._vec(c, g)
        d, a = self.dist, self.assign
        nd_, na_ = d.copy(), a.copy()
        down = d > nd                                   # kmedoids.py:644
        nd_[down] = nd[down]
        na_[down] = cid
        amb = np.flatnonzero((d <= nd) & (a == cid))    # :658
        trial_c, trial_G = self.med_c.copy(), self.med_G.copy()
        trial_c[cid], trial_G[cid] = c, g
        if len(amb):
            best = np.full(len(amb), np.inf, dtype=np.float32)
            lab = np.zeros(len(amb), dtype=np.int32)
            sub_c = np.ascontiguousarray(self.P.c[amb])
            sub_G = np.ascontiguousarray(self.P.G[amb])
            for k in range(len(trial_G)):               # util.py:199-203
                v = qcp.rmsd_centered(
                    sub_c, sub_G,
                    np.ascontiguousarray(trial_c[k].reshape(self.A, 3)),
                    float(trial_G[k]))
                closer = v < best
                best[closer] = v[closer]
                lab[closer] = k
            nd_[amb] = best
            na_[amb] = lab
        rec = np.zeros(1, dtype=PAM_OUT)
        rec["sum_old"] = np.square(d.astype(np.float64)).sum()
        rec["sum_new"] = np.square(nd_.astype(np.float64)).sum()
        rec["n_frames"] = self.n
        rec["n_amb"] = len(amb)
        moved = 0
        ch = a != na_
        for i in range(win_count):
            k = win_lo + i
            if np.any(ch & ((a == k) | (na_ == k))):
                moved |= 1 << i
        rec["moved"] = moved
        out.numpy()[:32] = rec.view(np.uint8)
        self.pending = (nd_, na_, trial_c, trial_G)

    def pam_commit(self, accept):
        if accept:
            self.dist, self.assign, self.med_c, self.med_G = self.pending
        self.pending = None


class HostShard(_PamMixin):
    candidates = 1          # one-center protocol (ek_kcenters_step)

    def __init__(self, xyz, global_offset):
        self.P = qcp.Prepared(xyz) if len(xyz) else None
        self.n = len(xyz)
        self.A = xyz.shape[1]
        self.offset = int(global_offset)
        self.dist = np.full(self.n, np.inf, dtype=np.float32)
        self.assign = np.full(self.n, -1, dtype=np.int32)
        self.hist = {}
        self.n_done = 0
        self.stopped = False

    @property
    def record_bytes(self):
        return (32 + 12 * self.A + 15) // 16 * 16

    def new_buffer(self, nbytes):
        return torch.zeros(nbytes, dtype=torch.uint8)

    def _write(self, rec, maxdist, valid, gidx, trace, coords):
        buf = rec.numpy()
        h = np.zeros(1, dtype=HDR)
        h["maxdist"], h["valid"], h["gidx"], h["trace"] = (maxdist, valid,
                                                           gidx, trace)
        buf[:32] = h.view(np.uint8)
        if coords is not None:
            buf[32:32 + 12 * self.A] = np.ascontiguousarray(
                coords, dtype=np.float32).view(np.uint8).ravel()

    def local_candidate(self, rec):
        if self.stopped:
            return
        if self.n == 0:
            self._write(rec, -np.inf, 0, -1, 0.0, None)
            return
        i = int(np.argmax(self.dist))
        self._write(rec, self.dist[i], 1, self.offset + i, self.P.G[i],
                    self.P.c[i])

    def step(self, all_recs, n_recs, label, cutoff, own_rec):
        buf = all_recs.numpy()
        rb = self.record_bytes
        best, wmax = 0, -np.inf
        for r in range(n_recs):
            h = buf[r * rb:r * rb + 32].view(HDR)[0]
            m = h["maxdist"] if h["valid"] else -np.inf
            if r == 0 or m > wmax:
                if r == 0 or m > wmax:
                    best, wmax = r, m
        h = buf[best * rb:best * rb + 32].view(HDR)[0]
        if not (float(wmax) > cutoff):
            self.stopped = True
            return
        coords = buf[best * rb + 32:best * rb + 32 + 12 * self.A].view(
            np.float32).reshape(self.A, 3).copy()
        self.hist[label] = (int(h["gidx"]), float(wmax))
        self.n_done = label + 1
        if self.n:
            self.P.kcenters_step(coords, float(h["trace"]), label, self.dist,
                                 self.assign)
        self.local_candidate(own_rec)

    def progress(self):
        return self.n_done

    # -- warm start (sharded.warm_start_sharded) ---------------------------------
    def assign_nearest(self, centers_xyz):
        if self.n == 0:
            return
        cc, Gc = qcp.center_and_trace(centers_xyz)
        a, d = qcp.assign_nearest(self.P.c, self.P.G, cc, Gc)
        self.dist[:] = d
        self.assign[:] = a

    def state(self):
        return self.dist, self.assign

    def set_state(self, distances, assignments):
        self.dist[:] = np.asarray(distances, dtype=np.float32)
        self.assign[:] = np.asarray(assignments, dtype=np.int32)

    @property
    def n_atoms(self):
        return self.A

    def history(self, first, count):
        idx = np.full(max(count, 1), -1, dtype=np.int64)
        cd = np.zeros(max(count, 1), dtype=np.float32)
        for i in range(count):
            if first + i in self.hist:
                idx[i], cd[i] = self.hist[first + i]
        return idx[:count], cd[:count], self.n_done

    def reset_history(self):
        self.hist = {}
        self.n_done = 0
        self.stopped = False


class HostShardRounds(HostShard):
    far_as_inf = False
    """Same shard speaking the multi-candidate round protocol
    (ek_spec_begin / _round / _localmax / _apply / _round_end / _progress)."""

    def __init__(self, xyz, global_offset, candidates=4):
        super().__init__(xyz, global_offset)
        self.candidates = candidates
        self.limit = 0
        self.plan = None

    def _hdr(self, buf, r):
        rb = self.record_bytes
        return buf[r * rb:r * rb + 32].view(HDR)[0]

    def _records(self, recs):
        """T candidate records: record 0 the local first-index arg-max, the
        others simply the next largest distances"""
        buf = recs.numpy()
        rb = self.record_bytes
        T = self.candidates
        order = []
        if self.n:
            first = int(np.argmax(self.dist))
            rest = [int(i) for i in np.argsort(-self.dist, kind="stable")
                    if int(i) != first]
            order = [first] + rest[:T - 1]
        for j in range(T):
            sub = recs[j * rb:(j + 1) * rb]
            if j < len(order):
                i = order[j]
                self._write(sub, self.dist[i], 1, self.offset + i, self.P.G[i],
                            self.P.c[i])
            else:
                self._write(sub, -np.inf, 0, -1, 0.0, None)

    def spec_begin(self, first_label, limit, recs):
        self.n_done = first_label
        self.limit = limit
        self.stopped = False
        self._records(recs)

    def spec_round(self, recs_all, n_recs, cutoff):
        buf = recs_all.numpy()
        rb = self.record_bytes
        self.plan = None
        if self.stopped or self.n_done >= self.limit:
            return
        hs = [(r, self._hdr(buf, r)) for r in range(n_recs)]
        hs = [(r, h) for r, h in hs if h["valid"]]
        hs.sort(key=lambda t: (-float(t[1]["maxdist"]), int(t[1]["gidx"])))
        hs = hs[:self.candidates]
        if not hs:
            return
        if not (float(hs[0][1]["maxdist"]) > cutoff):
            self.stopped = True
            return
        cands = []
        for r, h in hs:
            coords = buf[r * rb + 32:r * rb + 32 + 12 * self.A].view(
                np.float32).reshape(self.A, 3).copy()
            vec = (qcp.rmsd_centered(self.P.c, self.P.G, coords,
                                     float(h["trace"]))
                   if self.n else np.zeros(0, np.float32))
            cands.append(dict(gidx=int(h["gidx"]), vec=vec, used=False))
        self.plan = dict(cands=cands, miss=False)
        self._accept(cands[0], float(hs[0][1]["maxdist"]))
        if self.far_as_inf:
            # what the device pass stores (ek_qcp.h ek_rmsd_from_S_below): a
            # kept distance that is not below the frame's current one is +inf
            for cand in cands[1:]:
                cand["vec"] = np.where(cand["vec"] < self.dist, cand["vec"],
                                       np.float32(np.inf)).astype(np.float32)

    def _accept(self, cand, maxdist):
        label = self.n_done
        cand["used"] = True
        upd = cand["vec"] < self.dist
        self.dist[upd] = cand["vec"][upd]
        self.assign[upd] = label
        self.hist[label] = (cand["gidx"], maxdist)
        self.n_done = label + 1

    def spec_localmax(self, hdr):
        h = np.zeros(1, dtype=MAXHDR)
        if self.n:
            i = int(np.argmax(self.dist))
            h["maxdist"], h["valid"], h["gidx"] = self.dist[i], 1, self.offset + i
        else:
            h["maxdist"], h["valid"], h["gidx"] = -np.inf, 0, -1
        hdr.numpy()[:16] = h.view(np.uint8)

    def spec_apply(self, hdrs_all, n_hdrs, cutoff):
        if self.plan is None or self.plan["miss"]:
            return
        self.plan["miss"] = True
        if self.stopped or self.n_done >= self.limit:
            return
        hs = hdrs_all.numpy()[:16 * n_hdrs].view(MAXHDR)
        hs = [h for h in hs if h["valid"]]
        if not hs:
            return
        hs.sort(key=lambda h: (-float(h["maxdist"]), int(h["gidx"])))
        w = hs[0]
        if not (float(w["maxdist"]) > cutoff):
            self.stopped = True
            return
        for c in self.plan["cands"][1:]:
            if c["gidx"] == int(w["gidx"]) and not c["used"]:
                self.plan["miss"] = False
                self._accept(c, float(w["maxdist"]))
                return

    def spec_round_end(self, recs):
        self._records(recs)

    def spec_progress(self):
        return self.n_done, self.stopped


CHAIN_ROW = np.dtype([("cur", "<f4"), ("valid", "<i4"), ("d", "<f4", (8,))])
assert CHAIN_ROW.itemsize == 40


class HostShardChain(HostShardRounds):
    """HostShardRounds plus the chained cheap steps (ek_spec_chain_rows / _max
    / _apply): same record and header layouts as the device."""

    def spec_chain_bytes(self):
        return 8 * CHAIN_ROW.itemsize, 8 * MAXHDR.itemsize

    def spec_chain_rows(self, rows):
        r = np.zeros(8, dtype=CHAIN_ROW)
        if self.plan is not None:
            cands = self.plan["cands"]
            for j in range(1, len(cands)):
                local = cands[j]["gidx"] - self.offset
                if 0 <= local < self.n:
                    r[j]["valid"] = 1
                    r[j]["cur"] = self.dist[local]
                    for u in range(1, len(cands)):
                        r[j]["d"][u] = cands[u]["vec"][local]
        rows.numpy()[:r.nbytes] = r.view(np.uint8)

    def spec_chain_max(self, rows_all, n_shards, hdrs):
        out = np.zeros(8, dtype=MAXHDR)
        out["maxdist"], out["gidx"] = -np.inf, -1
        if self.plan is not None:
            cands = self.plan["cands"]
            allr = rows_all.numpy()[:n_shards * 8 * CHAIN_ROW.itemsize].view(
                CHAIN_ROW).reshape(n_shards, 8)
            row = {}
            for j in range(1, len(cands)):
                for s in range(n_shards):
                    if allr[s, j]["valid"]:
                        row[j] = allr[s, j]
                        break
            cur = {j: np.float32(row[j]["cur"]) for j in row}
            chain = []
            while cur:
                best = sorted(cur, key=lambda j: (-float(cur[j]),
                                                  cands[j]["gidx"]))[0]
                chain.append(best)
                del cur[best]
                for j in cur:
                    d = np.float32(row[j]["d"][best])
                    if d < cur[j]:
                        cur[j] = d
            self.plan["chain"] = chain
            if self.n:
                run = self.dist.copy()
                for k in range(len(chain)):
                    if k > 0:
                        run = np.minimum(run, cands[chain[k - 1]]["vec"])
                    i = int(np.argmax(run))
                    out[k]["maxdist"], out[k]["valid"] = run[i], 1
                    out[k]["gidx"] = self.offset + i
        hdrs.numpy()[:out.nbytes] = out.view(np.uint8)

    def spec_chain_apply(self, hdrs_all, n_shards, cutoff):
        if self.plan is None:
            return
        cands = self.plan["cands"]
        allh = hdrs_all.numpy()[:n_shards * 8 * MAXHDR.itemsize].view(
            MAXHDR).reshape(n_shards, 8)
        for k, j in enumerate(self.plan.get("chain", [])):
            hs = [h for h in allh[:, k] if h["valid"]]
            if self.stopped or self.n_done >= self.limit or not hs:
                break
            hs.sort(key=lambda h: (-float(h["maxdist"]), int(h["gidx"])))
            w = hs[0]
            if not (float(w["maxdist"]) > cutoff):
                self.stopped = True
                break
            if int(w["gidx"]) != cands[j]["gidx"]:
                break
            self._accept(cands[j], float(w["maxdist"]))


# (state: the state the offered records are the far frames of -- round 6; with the caller's
# collective as the exchange that is the state the whole chain would leave, or -1 without one)
MS_MSG = np.dtype([("n_recs", "<i4"), ("cn", "<i4"), ("state", "<i4"), ("pad", "<i4")])
MAX_CANDS = 32      # EK_MAX_CANDS: the per-prefix headers of a message


class HostShardMs(HostShard):
    """The shard speaking the one-exchange-per-round protocol of
    csrc/ek_mshard.hip (ek_ms_setup / _begin / _local / _global / _end), with
    the device's message layout: EkMsMsg | 32 x EkMaxHdr | `offer` records.
    Restated on the CPU with the checker's distances so that the product's
    driver loop (enspara_amd/sharded.py, gather transport) and the protocol's
    decisions -- per-prefix global maxima, lowest global index on ties, the
    chain that breaks and is re-offered -- run under gloo without a GPU.  The
    guesses are simpler than the device's (largest distances, no pairwise
    table); correctness never depends on them."""

    far_as_inf = True

    def __init__(self, xyz, global_offset, candidates=8):
        super().__init__(xyz, global_offset)
        self.candidates = candidates
        self.mode = 0
        self.exchanges = 0
        self.repicks = 0

    # -- layout ---------------------------------------------------------------
    def ms_setup(self, world, rank):
        self.world, self.rank = world, rank
        self.offer = max(1, 64 // world)
        self.msg_bytes = 16 + 16 * MAX_CANDS + self.offer * self.record_bytes
        return self.msg_bytes

    def ms_begin(self, first_label, limit):
        self.n_done = first_label
        self.limit = limit
        self.stopped = False
        self.mode = 2               # offer the records of the state as it stands
        self.pick_state = 0
        self.cands = []             # the round's candidates (dicts), [0] applied
        self.order = []             # presumed order of candidates 1..
        self.states = [self.dist]   # state k: after the first k of the order
        self.pending = 0
        self.rounds = 0

    def _state(self, k):
        return self.states[k]

    # -- pass + chain -> message ----------------------------------------------
    def ms_local(self, cutoff, msg):
        if self.mode == 0:
            return
        buf = msg.numpy()
        buf[:self.msg_bytes] = 0
        if self.mode == 1:
            # the pass: the pending chain of the round before is the state
            # already (applied when it was accepted, below); candidate 0 ...
            c0 = self.cands[0]
            for c in self.cands:
                c["vec"] = (qcp.rmsd_centered(self.P.c, self.P.G, c["coords"],
                                              c["trace"])
                            if self.n else np.zeros(0, np.float32))
            upd = c0["vec"] < self.dist
            self.dist[upd] = c0["vec"][upd]
            self.assign[upd] = c0["label"]
            # ... and the states every prefix of the presumed order would leave
            self.states = [self.dist.copy()]
            for j in self.order:
                v = self.cands[j]["vec"]
                if self.far_as_inf:     # what the device keeps of a far distance
                    v = np.where(v < self.states[0], v, np.float32(np.inf))
                self.states.append(np.minimum(self.states[-1], v))
        cn = len(self.order) if self.mode == 1 else 0
        hdrs = np.zeros(MAX_CANDS, dtype=MAXHDR)
        hdrs["maxdist"], hdrs["gidx"] = -np.inf, -1
        for k in range(cn):             # states 0 .. cn - 1
            if self.n:
                s = self.states[k]
                i = int(np.argmax(s))
                hdrs[k] = (s[i], 1, self.offset + i)
        ps = cn if self.mode == 1 else self.pick_state
        s = self.states[ps] if self.n else np.zeros(0, np.float32)
        order = []
        if self.n:
            first = int(np.argmax(s))
            rest = [int(i) for i in np.argsort(-s, kind="stable")
                    if int(i) != first]
            order = [first] + rest[:self.offer - 1]
        head = np.zeros(1, dtype=MS_MSG)
        head["n_recs"], head["cn"] = len(order), cn
        head["state"] = ps if self.mode == 1 else -1
        buf[:16] = head.view(np.uint8)
        buf[16:16 + hdrs.nbytes] = hdrs.view(np.uint8)
        rb = self.record_bytes
        base = 16 + 16 * MAX_CANDS
        for j, i in enumerate(order):
            sub = msg[base + j * rb:base + (j + 1) * rb]
            self._write(sub, s[i], 1, self.offset + i, self.P.G[i], self.P.c[i])

    # -- all messages -> decision, next plan -----------------------------------
    def ms_global(self, cutoff, msgs):
        if self.mode == 0:
            return
        buf = msgs.numpy()
        mb, rb = self.msg_bytes, self.record_bytes
        base = 16 + 16 * MAX_CANDS
        heads = [buf[r * mb:r * mb + 16].view(MS_MSG)[0]
                 for r in range(self.world)]
        cn = len(self.order) if self.mode == 1 else 0
        na = 0
        short = False
        if self.mode == 1:
            # candidate 0 of the pass that has just run is a center now
            c0 = self.cands[0]
            self.hist[c0["label"]] = (c0["gidx"], c0["maxdist"])
            self.n_done = c0["label"] + 1
            self.rounds += 1
            label0 = self.n_done
            for k in range(cn):
                hs = [buf[r * mb + 16:r * mb + 16 + 16 * MAX_CANDS].view(MAXHDR)[k]
                      for r in range(self.world)]
                hs = [h for h in hs if h["valid"]]
                if self.stopped or self.n_done >= self.limit or not hs:
                    break
                hs.sort(key=lambda h: (-float(h["maxdist"]), int(h["gidx"])))
                w = hs[0]
                if not (float(w["maxdist"]) > cutoff):      # kcenters.py:217
                    self.stopped = True
                    break
                cand = self.cands[self.order[k]]
                if int(w["gidx"]) != cand["gidx"]:
                    break
                self.hist[self.n_done] = (cand["gidx"], float(w["maxdist"]))
                self.n_done += 1
                na += 1
            short = na < cn
            # the accepted prefix becomes the state (labels in order)
            for k in range(na):
                v = self.cands[self.order[k]]["vec"]
                upd = v < self.dist
                self.dist[upd] = v[upd]
                self.assign[upd] = label0 + k
        self.exchanges += 1
        over = self.stopped or self.n_done >= self.limit
        if over:
            self.mode = 0
            return
        # (the offers describe the state this chain left on every shard, or they are
        # offered again: ek_ms_plan_kernel's `agree`)
        if self.mode == 1 and not all(int(h["state"]) == na for h in heads):
            assert short
            self.mode, self.pick_state = 2, na
            self.repicks += 1
            return
        # the next round's candidates: the largest offered distances, lowest
        # global index on ties
        offers = []
        for r in range(self.world):
            for j in range(int(heads[r]["n_recs"])):
                o = r * mb + base + j * rb
                h = buf[o:o + 32].view(HDR)[0]
                offers.append((-float(h["maxdist"]), int(h["gidx"]), o))
        offers.sort()
        offers = offers[:self.candidates]
        if not offers:
            self.mode = 0
            return
        if not (-offers[0][0] > cutoff):
            self.stopped = True
            self.mode = 0
            return
        self.cands = []
        for nd, g, o in offers:
            h = buf[o:o + 32].view(HDR)[0]
            self.cands.append(dict(
                gidx=g, trace=float(h["trace"]),
                coords=buf[o + 32:o + 32 + 12 * self.A].view(np.float32)
                .reshape(self.A, 3).copy()))
        self.cands[0]["label"] = self.n_done     # (counted once its pass has run)
        self.cands[0]["maxdist"] = -offers[0][0]
        self.order = list(range(1, len(self.cands)))
        self.mode = 1

    def ms_end(self):
        self.mode = 0

    def spec_progress(self):
        return self.n_done, self.stopped
