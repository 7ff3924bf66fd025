"""Minimal GROMACS XTC reader -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Purpose: the reference pins its clustering hot path with known-answer values
computed by mdtraj on ``enspara/test/data/frame0.xtc``
(enspara/test/test_cluster.py:209-218, :231-238, :547).  mdtraj is not
available here, so to check the oracle's RMSD against those values the
fixture has to be decoded independently.  This module restates the published
XTC "xdr3dfcoord" compressed-coordinate format (GROMACS xdrfile library,
libxdrf.c: magic 1995, big-endian XDR header, mixed-radix packed integer
triplets with run-length coded small displacements and the water-molecule
first/second atom swap).  Pure Python, sized for the reference's small test
files.  Not used by the product.
"""
import struct

import numpy as np

_MAGICINTS = [
    0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64,
    80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024, 1290,
    1625, 2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003,
    16384, 20642, 26007, 32768, 41285, 52015, 65536, 82570, 104031,
    131072, 165140, 208063, 262144, 330280, 416127, 524287, 660561,
    832255, 1048576, 1321122, 1664510, 2097152, 2642245, 3329021,
    4194304, 5284491, 6658042, 8388607, 10568983, 13316085, 16777216]
_FIRSTIDX = 9


class _Bits:
    """MSB-first bit reader over a byte string."""

    def __init__(self, data):
        self.big = int.from_bytes(data, "big")
        self.total = 8 * len(data)
        self.pos = 0

    def take(self, nbits):
        if nbits == 0:
            return 0
        if self.pos + nbits > self.total:
            raise ValueError("XTC: compressed block exhausted")
        shift = self.total - self.pos - nbits
        self.pos += nbits
        return (self.big >> shift) & ((1 << nbits) - 1)

    def take_ints(self, nbits, sizes):
        """Unpack three integers stored as one mixed-radix number whose bytes
        were written least-significant first."""
        v = 0
        shift = 0
        left = nbits
        while left > 8:
            v |= self.take(8) << shift
            shift += 8
            left -= 8
        if left > 0:
            v |= self.take(left) << shift
        c2 = v % sizes[2]
        v //= sizes[2]
        c1 = v % sizes[1]
        c0 = v // sizes[1]
        return [c0, c1, c2]


def _bits_for(size):
    n, bits = 1, 0
    while size >= n and bits < 32:
        bits += 1
        n <<= 1
    return bits


def _decode_coords(natoms, buf, off):
    """Decode one xdr3dfcoord block starting at ``off``.
    Returns (float32 [natoms, 3], new offset)."""
    (lsize,) = struct.unpack_from(">i", buf, off)
    off += 4
    if lsize != natoms:
        raise ValueError("XTC: atom count mismatch")
    if lsize <= 9:
        xyz = np.frombuffer(buf, dtype=">f4", count=3 * lsize, offset=off)
        return xyz.astype(np.float32).reshape(lsize, 3), off + 12 * lsize
    (precision,) = struct.unpack_from(">f", buf, off)
    off += 4
    minint = struct.unpack_from(">3i", buf, off)
    off += 12
    maxint = struct.unpack_from(">3i", buf, off)
    off += 12
    (smallidx,) = struct.unpack_from(">i", buf, off)
    off += 4
    (nbytes,) = struct.unpack_from(">i", buf, off)
    off += 4
    data = bytes(buf[off:off + nbytes])
    off += (nbytes + 3) // 4 * 4

    sizeint = [maxint[i] - minint[i] + 1 for i in range(3)]
    if any(s > 0xFFFFFF for s in sizeint):
        bitsizeint = [_bits_for(s) for s in sizeint]
        bitsize = 0
    else:
        bitsizeint = None
        bitsize = (sizeint[0] * sizeint[1] * sizeint[2]).bit_length()

    smaller = _MAGICINTS[max(_FIRSTIDX, smallidx - 1)] // 2
    smallnum = _MAGICINTS[smallidx] // 2
    sizesmall = [_MAGICINTS[smallidx]] * 3

    bits = _Bits(data)
    out = np.empty((lsize, 3), dtype=np.int64)
    w = 0  # atoms written
    i = 0  # atoms decoded
    run = 0
    while i < lsize:
        if bitsize == 0:
            this = [bits.take(bitsizeint[0]), bits.take(bitsizeint[1]),
                    bits.take(bitsizeint[2])]
        else:
            this = bits.take_ints(bitsize, sizeint)
        i += 1
        this = [this[j] + minint[j] for j in range(3)]
        prev = list(this)

        is_smaller = 0
        if bits.take(1) == 1:
            run = bits.take(5)
            is_smaller = run % 3
            run -= is_smaller
            is_smaller -= 1
        if run > 0:
            for k in range(0, run, 3):
                small = bits.take_ints(smallidx, sizesmall)
                i += 1
                this = [small[j] + prev[j] - smallnum for j in range(3)]
                if k == 0:
                    # first two atoms of a run are stored swapped
                    this, prev = prev, this
                    out[w] = prev
                    w += 1
                else:
                    prev = list(this)
                out[w] = this
                w += 1
        else:
            out[w] = this
            w += 1

        smallidx += is_smaller
        if is_smaller < 0:
            smallnum = smaller
            smaller = (_MAGICINTS[smallidx - 1] // 2
                       if smallidx > _FIRSTIDX else 0)
        elif is_smaller > 0:
            smaller = smallnum
            smallnum = _MAGICINTS[smallidx] // 2
        sizesmall = [_MAGICINTS[smallidx]] * 3
    if w != lsize:
        raise ValueError("XTC: decoded %d of %d atoms" % (w, lsize))
    inv = np.float32(1.0 / float(precision))  # double divide, rounded once
    return (out.astype(np.float32) * inv).astype(np.float32), off


def read_xtc(path):
    """-> dict(xyz float32 [n_frames, n_atoms, 3] in nm, step, time, box)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    off = 0
    frames, steps, times, boxes = [], [], [], []
    while off < len(buf):
        magic, natoms, step = struct.unpack_from(">3i", buf, off)
        if magic != 1995:
            raise ValueError("XTC: unsupported magic %d" % magic)
        off += 12
        (time,) = struct.unpack_from(">f", buf, off)
        off += 4
        box = np.frombuffer(buf, dtype=">f4", count=9, offset=off)
        off += 36
        xyz, off = _decode_coords(natoms, buf, off)
        frames.append(xyz)
        steps.append(step)
        times.append(time)
        boxes.append(box.astype(np.float32).reshape(3, 3))
    return dict(xyz=np.stack(frames), step=np.array(steps),
                time=np.array(times, dtype=np.float32), box=np.stack(boxes))
