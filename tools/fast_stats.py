#!/usr/bin/env python3
"""Survivor and corner fractions of the benchmark frames per pyramid level (numpy, CPU): what fraction of the tested pixels
passes the round-1 compass pre-test (two of the four compass pixels), the round-2 opposite-pair test on two / four / eight
pairs, how many are FAST-9 corners before NMS, and how many survivors allow both polarities.  These fractions size the phases of
k_fast_cells.  usage: python tools/fast_stats.py [seed ...]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from eao_fusion_amd import synth
from oracle import oracle as O

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def stats(img, t):
    h, w = img.shape
    I = img.astype(np.int32)
    c = I[3:h - 3, 3:w - 3]
    d = np.stack([c - I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])
    dark, bright = d > t, (-d) > t
    compass = ((dark[[0, 4, 8, 12]].sum(0) >= 2) | (bright[[0, 4, 8, 12]].sum(0) >= 2)).mean()
    pd = np.ones_like(c, bool); pb = np.ones_like(c, bool)
    out = {}
    for n, ks in ((2, (0, 4)), (4, (2, 6)), (8, (1, 3, 5, 7))):
        for k in ks:
            pd &= dark[k] | dark[k + 8]; pb &= bright[k] | bright[k + 8]
        out[n] = ((pd | pb).mean(), (pd & pb).sum() / max((pd | pb).sum(), 1))
    def run9(m):
        mm = np.concatenate([m, m[:8]])
        r = np.zeros_like(c, bool)
        for s in range(16):
            r |= mm[s:s + 9].all(0)
        return r
    corner = (run9(dark) | run9(bright)).mean()
    return compass, out, corner


for seed in [int(a) for a in sys.argv[1:]] or [1000, 1001]:
    lv = synth.synth_frame(seed)
    print("frame seed %d: level, size, compass(r1), pairs2(r2) [both-polarity share], pairs4, pairs8, corners before NMS" % seed)
    for l in range(8):
        if l:
            w = int(np.rint(np.float32(640) * np.float32(1 / np.float32(1.2) ** l))); h = int(np.rint(np.float32(480) * np.float32(1 / np.float32(1.2) ** l)))
            lv = O.resize_linear(lv, w, h)
        cmp_, o, cor = stats(lv, 20)
        print("  %d %4dx%-4d  %.4f  %.4f [%.3f]  %.4f  %.4f  %.4f" % (l, lv.shape[1], lv.shape[0], cmp_, o[2][0], o[2][1], o[4][0], o[8][0], cor))
