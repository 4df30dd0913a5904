#!/usr/bin/env python3
"""Search an XOR swizzle (GF(2)-linear map of the low 5 bits of an 8-byte LDS slot index) that makes
every access pattern of the in-place FFT exchange conflict free on gfx950:
  ds_read_b64 : two 32-lane groups, 32 distinct slots mod 32 needed
  ds_write_b64: four 16-lane groups, 16 distinct slots mod 16 needed
(MI355X_MICROARCH.md, LDS table).  Output: 5-bit column constants col[j], j < LOGP; physical slot =
(a & ~31) | XOR_{j : bit j of a set} col[j].
"""
import itertools
import random
import sys


def rank(vecs, bits):
    vecs = [v & ((1 << bits) - 1) for v in vecs]
    r = 0
    for b in range(bits):
        piv = next((i for i in range(r, len(vecs)) if (vecs[i] >> b) & 1), None)
        if piv is None:
            continue
        vecs[r], vecs[piv] = vecs[piv], vecs[r]
        for i in range(len(vecs)):
            if i != r and (vecs[i] >> b) & 1:
                vecs[i] ^= vecs[r]
        r += 1
    return r


def lane_bit_maps(logp, logr):
    """For every LDS layout used: list of address-bit index per lane bit (lane bit i -> address bit)."""
    full = logp // logr
    loglast = logp - full * logr
    np_ = full + (1 if loglast else 0)
    logt = logp - logr
    layouts = []
    for s in range(np_):
        lr = logr if s < full else loglast
        ls = logp - min(s, full) * logr
        ls1 = ls - lr
        groups_log = logr - lr
        m = []
        for i in range(logt):
            if groups_log == 0:
                m.append(i if i < ls1 else i + lr)          # tau = Q*S1 + t'
            else:
                m.append(i + logr)                           # pi = tau*g + gi ; a = pi*r + m
        layouts.append(m)
    layouts.append(list(range(logt)))                        # spectrum publish [rho][tau]
    return layouts


def ok(cols, layouts):
    for m in layouts:
        if len(m) >= 5 and rank([cols[j] for j in m[:5]], 5) < 5:
            return False
        if len(m) >= 4 and rank([cols[j] for j in m[:4]], 4) < 4:
            return False
    return True


def search(logp, logr, tries=2_000_00, seed=1):
    layouts = lane_bit_maps(logp, logr)
    ident = [1 << j if j < 5 else 0 for j in range(logp)]
    if ok(ident, layouts):
        return ident
    rng = random.Random(seed)
    best = None
    for _ in range(tries):
        cols = [rng.randrange(32) for _ in range(logp)]
        if rank(cols[:5], 5) < 5:
            continue
        if ok(cols, layouts):
            cost = sum(bin(c).count("1") for c in cols)
            if best is None or cost < best[0]:
                best = (cost, cols)
    return best[1] if best else None


if __name__ == "__main__":
    pairs = [(10, 2), (10, 3), (10, 4), (9, 3), (9, 2), (8, 2), (11, 3), (11, 4)]
    if len(sys.argv) > 2:           # find_swizzle.py LOGP LOGR
        pairs = [(int(sys.argv[1]), int(sys.argv[2]))]
    for logp, logr in pairs:
        cols = search(logp, logr)
        print(f"LOGP={logp} LOGR={logr}: {cols}")
