"""Invariants of the guided-matching oracle (oracle/match_cpu.cpp; reference src/ORBmatcher.cc:45-129, 1328-1472).
No upstream fixtures exist, so the sequential restatement is checked against an independent numpy brute force of the
same rules (dense masks + greedy replay)."""
import numpy as np

from eao_fusion_amd import synth


def _grid_cells(F):
    iw = np.float32(64) / np.float32(F["max_x"] - F["min_x"])
    ih = np.float32(48) / np.float32(F["max_y"] - F["min_y"])
    # C round(): half away from zero
    px = np.floor(np.abs((F["kp_x"] - F["min_x"]) * iw) + np.float32(0.5)) * np.sign((F["kp_x"] - F["min_x"]) * iw)
    py = np.floor(np.abs((F["kp_y"] - F["min_y"]) * ih) + np.float32(0.5)) * np.sign((F["kp_y"] - F["min_y"]) * ih)
    return px.astype(int), py.astype(int), iw, ih


def _brute_points(F, M, th, ratio):
    px, py, iw, ih = _grid_cells(F)
    N = len(px)
    ingrid = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    order = sorted(np.nonzero(ingrid)[0], key=lambda i: (px[i], py[i], i))
    occ = F["occupied"].astype(bool).copy()
    out = np.full(len(M["level"]), -1)
    for m in range(len(out)):
        if M["skip"][m]:
            continue
        lvl = int(M["level"][m])
        r = np.float32(2.5 if M["view_cos"][m] > 0.998 else 4.0)
        if th != 1.0:
            r = np.float32(r * np.float32(th))
        rs = np.float32(r * F["scale_factors"][lvl])
        x, y = M["proj_x"][m], M["proj_y"][m]
        x0 = max(0, int(np.floor((x - F["min_x"] - rs) * iw))); x1 = min(63, int(np.ceil((x - F["min_x"] + rs) * iw)))
        y0 = max(0, int(np.floor((y - F["min_y"] - rs) * ih))); y1 = min(47, int(np.ceil((y - F["min_y"] + rs) * ih)))
        if x0 >= 64 or x1 < 0 or y0 >= 48 or y1 < 0:
            continue
        best, best2, bl, bl2, bi = 256, 256, -1, -1, -1
        for i in order:
            if not (x0 <= px[i] <= x1 and y0 <= py[i] <= y1):
                continue
            o = F["kp_octave"][i]
            if o < lvl - 1 or o > lvl:
                continue
            if not (abs(F["kp_x"][i] - x) < rs and abs(F["kp_y"][i] - y) < rs):
                continue
            if occ[i]:
                continue
            if F["u_right"][i] > 0 and abs(M["proj_xr"][m] - F["u_right"][i]) > rs:
                continue
            d = int(np.unpackbits(M["descriptors"][m] ^ F["descriptors"][i]).sum())
            if d < best:
                best2, best, bl2, bl, bi = best, d, bl, o, i
            elif d < best2:
                bl2, best2 = o, d
        if best <= 100:
            if bl == bl2 and best > np.float32(ratio) * best2:
                continue
            out[m] = bi
            occ[bi] = True
    return out


def test_points_search_vs_bruteforce(oracle):
    cur, last, mps = synth.synth_tracking(n=300, seed=7001)
    for th in (1.0, 3.0):
        nm, got = oracle.search_by_projection_points(cur, mps, th, 0.8)
        ref = _brute_points(cur, mps, th, 0.8)
        assert np.array_equal(got, ref)
        assert nm == (ref >= 0).sum() and nm > 100
        m = got[got >= 0]
        assert len(set(m)) == len(m)                    # a keypoint is claimed at most once
        assert not cur["occupied"][m].any()             # and never one that already held a map point
        assert not (got[mps["skip"] > 0] >= 0).any()


def test_frames_search_properties(oracle):
    cur, last, mps = synth.synth_tracking(n=400, seed=7002)
    nm, cm = oracle.search_by_projection_frames(cur, last, 7.0, False, True)
    nm0, cm0 = oracle.search_by_projection_frames(cur, last, 7.0, False, False)
    assert nm == (cm >= 0).sum() and nm0 == (cm0 >= 0).sum()
    assert 150 < nm <= nm0                                # the rotation histogram only removes matches
    k = np.nonzero(cm >= 0)[0]
    assert last["valid"][cm[k]].all() and not cur["occupied"][k].any()
    assert len(set(cm[k])) == len(k)
    # planted rotation is +12 deg: every surviving match sits in one of (at most) three histogram bins
    rot = (last["angle"][cm[k]] - cur["kp_angle"][k]) % 360
    bins = np.round(rot * np.float32(30 / 360.0)).astype(int) % 30
    assert len(set(bins)) <= 3
    # monocular flag widens the level window (no forward/backward test)
    nm_m, _ = oracle.search_by_projection_frames(cur, last, 7.0, True, True)
    assert nm_m > 0
