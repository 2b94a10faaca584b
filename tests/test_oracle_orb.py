"""Pins for the CPU ORB oracle (oracle/orb_cpu.cpp).  The reference has no tests of its own, so
these are the known answers this build can state: reference-derived constants, the pattern table compared
number-for-number with the reference file when it is present, and independent numpy re-derivations of the
OpenCV primitives (documented generic 3.3.x algorithms)."""
import hashlib
import json
import os

import numpy as np
import pytest

from eao_fusion_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_tables_match_reference_derivation(oracle):
    t = oracle.OrbOracle(1000, 1.2, 8, 20, 7).tables()
    # SURVEY.md s8a (derived from reference src/ORBextractor.cc:436-447,455-469)
    assert list(t["quota"]) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(t["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    sc = np.float32(1.0)
    for i in range(8):
        assert t["scale"][i] == sc
        assert t["sigma2"][i] == np.float32(sc * sc)
        assert t["inv_scale"][i] == np.float32(1.0) / sc
        sc = np.float32(np.float64(sc) * np.float64(np.float32(1.2)))


def test_level_sizes(oracle):
    e = oracle.OrbOracle()
    e.extract(synth.synth_frame(1000))
    dims = [e.level_dims(l) for l in range(8)]
    assert dims == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]


def test_pattern_matches_fixture_and_reference(oracle):
    fx = json.load(open(os.path.join(GOLD, "orb_pattern.json")))
    vals = fx["values"]
    assert len(vals) == 1024
    assert hashlib.sha256(bytes((v & 0xFF) for v in vals)).hexdigest() == fx["sha256_int8"]
    p = oracle.lib().orc_orb_pattern()
    assert [p[i] for i in range(1024)] == vals
    ref = "/root/reference/src/ORBextractor.cc"
    if os.path.exists(ref):  # build container only; the GPU box has the fixture
        import importlib.util
        spec = importlib.util.spec_from_file_location("gen", os.path.join(os.path.dirname(GOLD), "..", "tools", "gen_orb_pattern.py"))
        gen = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gen)
        assert gen.parse_reference_pattern(ref) == vals


def test_gaussian_taps(oracle):
    taps = np.zeros(7, np.int32)
    oracle.lib().orc_gaussian_taps(taps.ctypes.data)
    assert list(taps) == [18, 34, 49, 55, 49, 34, 18]


def test_gaussian_blur_vs_numpy(oracle):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    rows = sum(k[t] * pad[:, t:t + 53] for t in range(7))
    out = sum(k[t] * rows[t:t + 37, :] for t in range(7))
    ref = np.clip((out + 32768) >> 16, 0, 255).astype(np.uint8)
    assert np.array_equal(oracle.gaussian_blur7(img), ref)


def _resize_numpy(src, dw, dh):
    sh, sw = src.shape
    def coeffs(d, s):
        scale = 1.0 / (float(d) / s)
        f = (((np.arange(d) + 0.5) * scale) - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(np.float32)).astype(np.float32)
        return i, f
    ix, fx = coeffs(dw, sw)
    fx[ix < 0] = 0; ix[ix < 0] = 0
    fx[ix >= sw - 1] = 0; ix[ix >= sw - 1] = sw - 1
    iy, fy = coeffs(dh, sh)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    S = src.astype(np.int64)
    ix1 = np.minimum(ix + 1, sw - 1)
    H = S[:, ix] * a0 + S[:, ix1] * a1
    y0 = np.clip(iy, 0, sh - 1); y1 = np.clip(iy + 1, 0, sh - 1)
    v = (((b0[:, None] * (H[y0] >> 4)) >> 16) + ((b1[:, None] * (H[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape,dst", [((480, 640), (533, 400)), ((134, 179), (149, 112)), ((40, 33), (27, 31))])
def test_resize_vs_numpy(oracle, shape, dst):
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(img, dst[0], dst[1]), _resize_numpy(img, dst[0], dst[1]))


def _fast_bruteforce(img, th):
    """Literal segment test + score + NMS, written independently of the oracle's bit tricks."""
    h, w = img.shape
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
            (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    score = np.zeros((h, w), np.int32)
    corner = np.zeros((h, w), bool)
    I = img.astype(np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = I[y, x]
            d = [v - I[y + dy, x + dx] for dx, dy in ring]
            best = -999
            ok = False
            for s in range(16):
                arc = [d[(s + j) % 16] for j in range(9)]
                if min(arc) > th or -max(arc) > th:
                    ok = True
                best = max(best, min(arc), -max(arc))
            if ok:
                corner[y, x] = True
                score[y, x] = max(best, th) - 1
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if corner[y, x]:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if score[y, x] > nb.max():
                    out.append((x, y, score[y, x]))
    return np.array(out, np.float32).reshape(-1, 3)


def test_fast_vs_bruteforce(oracle):
    img = synth.synth_frame(11, 96, 80, n_rect=20, n_small=60)
    for th in (20, 7):
        got = oracle.fast(img, th)
        ref = _fast_bruteforce(img, th)
        assert len(ref) > 5
        assert np.array_equal(got, ref)


def test_fast_atan2(oracle):
    rng = np.random.default_rng(7)
    ys = rng.integers(-200000, 200000, 2000).astype(np.float32)
    xs = rng.integers(-200000, 200000, 2000).astype(np.float32)
    L = oracle.lib()
    got = np.array([L.orc_fast_atan2(float(y), float(x)) for y, x in zip(ys, xs)])
    ref = np.degrees(np.arctan2(ys.astype(np.float64), xs.astype(np.float64))) % 360.0
    err = np.abs(((got - ref) + 180) % 360 - 180)
    assert err.max() < 0.02  # OpenCV documents ~0.3 deg worst case for this polynomial; measured far tighter
    assert L.orc_fast_atan2(0.0, 0.0) == 0.0
    assert L.orc_fast_atan2(0.0, -5.0) == 180.0


def test_distribute_small_known_answer(oracle):
    # 4 well separated clusters in a 100x100 box, N = 4: one best-response point per quadrant, list order
    # = children pushed to the front in quadrant order UL,UR,BL,BR => output BR,BL,UR,UL
    pts = np.array([[10, 10, 5], [12, 11, 9], [80, 12, 7], [81, 15, 3], [15, 85, 4], [11, 80, 6], [90, 90, 8], [85, 88, 2]], np.float32)
    sel = oracle.distribute(pts, 0, 100, 0, 100, 4)
    assert list(sel) == [6, 5, 2, 1]


def test_extract_end_to_end_properties(oracle):
    e = oracle.OrbOracle()
    img = synth.synth_frame(1000)
    kps, desc = e.extract(img)
    assert 1000 <= len(kps) <= 1000 + 8 * 2
    assert desc.shape == (len(kps), 32)
    assert np.all(np.diff(kps["octave"]) >= 0)           # concatenated by level
    assert np.all(kps["class_id"] == -1)
    assert np.all((kps["angle"] >= 0) & (kps["angle"] < 360))
    t = e.tables()
    for l in range(8):
        m = kps["octave"] == l
        lk = e.level_keypoints(l)
        assert m.sum() == len(lk)
        w, h = e.level_dims(l)
        assert np.all((lk["x"] >= 19) & (lk["x"] < w - 19) & (lk["y"] >= 19) & (lk["y"] < h - 19))  # appendix A.1
        assert np.all(kps["size"][m] == np.float32(int(31 * t["scale"][l])))
        if l:
            assert np.array_equal(kps["x"][m], lk["x"] * t["scale"][l])
    # determinism
    kps2, desc2 = e.extract(img)
    assert np.array_equal(kps, kps2) and np.array_equal(desc, desc2)
    # empty image: no outputs touched (reference :1046-1047)
    assert e.L.orc_orb_run(e.h, None, 0, 0, 0) == 0


def test_low_texture_uses_min_threshold(oracle):
    e = oracle.OrbOracle()
    img = synth.synth_frame(7, n_rect=40, n_small=0)
    kps, _ = e.extract(img)
    assert len(kps) > 50
    assert kps["response"].min() < 20  # a response below iniThFAST can only come from the minThFAST retry


def test_stereo_matches_against_independent_replay(oracle):
    """Frame::ComputeStereoMatches (src/Frame.cc:841-1013): the oracle against a replay written here from the reference
    text over the oracle's own pyramid levels (numpy patches padded with reflect-101, float32 arithmetic)."""
    from eao_fusion_amd import synth
    left, right = synth.synth_stereo_pair(9100, 400, 300, 4)
    cfg = (400, 1.2, 4, 20, 7)
    ol, orr = oracle.OrbOracle(*cfg), oracle.OrbOracle(*cfg)
    kl, dl = ol.extract(left)
    kr, dr = orr.extract(right)
    F = np.float32
    bf, mb = F(40.0), F(40.0 / 535.4)
    got_u, got_d = oracle.stereo_matches(ol, orr, kl, dl, kr, dr, mb, bf)
    tab = ol.tables()
    scale, inv = tab["scale"], tab["inv_scale"]
    pad = 19
    pyrL = [np.pad(ol.level_image(l).astype(np.float32), pad, mode="reflect") for l in range(4)]
    pyrR = [np.pad(orr.level_image(l).astype(np.float32), pad, mode="reflect") for l in range(4)]
    n_rows = left.shape[0]
    rows = [[] for _ in range(n_rows)]
    for i in range(len(kr)):
        r = F(2.0) * scale[kr["octave"][i]]
        for yi in range(int(np.floor(kr["y"][i] - r)), int(np.ceil(kr["y"][i] + r)) + 1):
            if 0 <= yi < n_rows:
                rows[yi].append(i)
    maxD = bf / mb
    exp_u = np.full(len(kl), -1, np.float32)
    exp_d = np.full(len(kl), -1, np.float32)
    dist_idx = []
    for iL in range(len(kl)):
        lvl, uL, vL = int(kl["octave"][iL]), kl["x"][iL], kl["y"][iL]
        cands = rows[int(vL)]
        minU, maxU = uL - maxD, uL + F(3)
        if not cands or maxU < 0:
            continue
        best, bi = 100, 0
        for iR in cands:
            if abs(int(kr["octave"][iR]) - lvl) > 1 or not (minU <= kr["x"][iR] <= maxU):
                continue
            d = int(np.unpackbits(dl[iL] ^ dr[iR]).sum())
            if d < best:
                best, bi = d, iR
        if best >= 100:
            continue
        sf = inv[lvl]
        rnd = lambda v: int(np.floor(abs(v) + F(0.5)) * np.sign(v))          # C round(): half away from zero
        su, sv, sr0 = rnd(uL * sf), rnd(vL * sf), rnd(kr["x"][bi] * sf)
        w = 5
        if sr0 + 5 - w < 0 or sr0 + 5 + w + 1 >= pyrR[lvl].shape[1] - 2 * pad:
            continue
        IL = pyrL[lvl][pad + sv - w:pad + sv + w + 1, pad + su - w:pad + su + w + 1]
        IL = IL - IL[w, w]
        dists = []
        for inc in range(-5, 6):
            IR = pyrR[lvl][pad + sv - w:pad + sv + w + 1, pad + sr0 + inc - w:pad + sr0 + inc + w + 1]
            dists.append(F(np.abs(IL - (IR - IR[w, w])).sum()))
        binc = int(np.argmin(dists)) - 5                                      # first minimum, like the strict '<'
        if binc in (-5, 5):
            continue
        d1, d2, d3 = dists[5 + binc - 1], dists[5 + binc], dists[5 + binc + 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            delta = (d1 - d3) / (F(2.0) * (d1 + d3 - F(2.0) * d2))
        if not (-1 <= delta <= 1):
            continue
        bu = scale[lvl] * (F(sr0) + F(binc) + delta)
        disp = uL - bu
        if 0 <= disp < maxD:
            if disp <= 0:
                disp, bu = F(0.01), uL - F(0.01)
            exp_d[iL], exp_u[iL] = bf / disp, bu
            dist_idx.append((int(dists[5 + binc]), iL))
    dist_idx.sort()
    th = F(1.5) * F(1.4) * F(dist_idx[len(dist_idx) // 2][0])
    for d, i in reversed(dist_idx):
        if d < th:
            break
        exp_u[i] = exp_d[i] = -1
    assert (exp_u >= 0).sum() > 50
    assert np.array_equal(got_u, exp_u) and np.array_equal(got_d, exp_d)
