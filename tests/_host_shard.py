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


class HostShard:
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
