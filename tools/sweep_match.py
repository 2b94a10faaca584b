"""Randomised bit-exactness sweep of SearchByProjection(Frame, vector<MapPoint*>) / (Frame, Frame), AssignFeaturesToGrid, isInFrustum and
ComputeStereoFromRGBD against the CPU oracle: frame sizes, radii, ratio thresholds, forward / backward motion (the level windows),
monocular frames, occupied keypoints, grid shapes.  Not part of the test suite: run by hand on a GPU box.
    python tools/sweep_match.py [seed] [cases]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth, frame as FR
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fb = O.frame_binding()
bad = calls = 0
def check(name, ok, kw):
    global bad, calls
    calls += 1
    if not ok:
        bad += 1
        print("MISMATCH %s %s" % (name, kw), flush=True)
for it in range(N):
    kw = dict(n=int(rng.choice([rng.integers(20, 300), rng.integers(300, 1500), rng.integers(1500, 4000)])), seed=int(rng.integers(0, 1 << 30)),
              moved=float(rng.choice([0.0, 0.03, 0.3, -0.3])), mono_frac=float(rng.choice([0.0, 0.3, 1.0])), occupied_frac=float(rng.choice([0.0, 0.2, 0.7])))
    try:
        cur, last, mps = synth.synth_tracking(**kw)
        th, ratio = float(rng.choice([1.0, 3.0, 5.0, 8.0])), float(rng.choice([0.6, 0.8, 0.9]))
        nm, got = E.ORBmatcher(ratio, True).SearchByProjectionPoints(cur, mps, th)
        onm, ref = O.search_by_projection_points(cur, mps, th, ratio)
        check("points", nm == onm and np.array_equal(got, ref), kw)
        th2, mono, chk = float(rng.choice([7.0, 15.0, 30.0])), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        nm, got = E.ORBmatcher(ratio, chk).SearchByProjectionFrames(cur, last, th2, mono)
        onm, ref = O.search_by_projection_frames(cur, last, th2, mono, chk)
        check("frames", nm == onm and np.array_equal(got, ref), kw)
        # grid
        n = int(rng.integers(0, 5000)); spread = float(rng.choice([0.0, 20.0, 300.0]))
        kx = rng.uniform(-spread, 640 + spread, n).astype(np.float32); ky = rng.uniform(-spread, 480 + spread, n).astype(np.float32)
        kx[: n // 8] = np.round(kx[: n // 8] / 10) * 10 + rng.choice([0.0, 5.0])
        b = (0, 0, 640, 480) if rng.random() < 0.5 else (float(rng.uniform(-9, 0)), float(rng.uniform(-9, 0)), float(rng.uniform(640, 660)), float(rng.uniform(480, 495)))
        shape = (64, 48) if rng.random() < 0.6 else (int(rng.integers(4, 80)), int(rng.integers(4, 60)))
        a1, a2 = FR.assign_features_to_grid(kx, ky, *b, *shape), fb.assign_features_to_grid(kx, ky, *b, *shape)
        check("grid", np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1]), dict(n=n, bounds=b, shape=shape))
        # RGB-D stereo
        depth = rng.uniform(0.3, 8, (480, 640)).astype(np.float32)
        depth[rng.random((480, 640)) < 0.2] = 0
        depth[rng.random((480, 640)) < 0.02] = -1
        kx = rng.uniform(0, 639.9, n).astype(np.float32); ky = rng.uniform(0, 479.9, n).astype(np.float32); ku = (kx + rng.normal(0, 0.5, n)).astype(np.float32)
        mbf = float(rng.choice([40.0, 35.5, 386.1]))
        a1, a2 = FR.compute_stereo_from_rgbd(kx, ky, ku, depth, mbf), fb.compute_stereo_from_rgbd(kx, ky, ku, depth, mbf)
        check("rgbd", np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1]), dict(n=n, mbf=mbf))
        # frustum
        m = int(rng.integers(1, 30000))
        X = rng.uniform([-8, -5, -2], [8, 5, 14], (m, 3)).astype(np.float32)
        nrm = X / np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-3)
        nrm = (nrm + rng.normal(0, 0.4, (m, 3))).astype(np.float32); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        d = np.linalg.norm(X, axis=1).astype(np.float32)
        pts = dict(active=(rng.random(m) < 0.9).astype(np.uint8), Xw=X, normal=nrm.astype(np.float32), min_dist_inv=(d * rng.uniform(0.3, 1.1, m)).astype(np.float32),
                   max_dist_inv=(d * rng.uniform(0.9, 3.0, m)).astype(np.float32), max_dist=(d * rng.uniform(1.0, 4.0, m)).astype(np.float32),
                   descriptors=np.zeros((m, 32), np.uint8))
        Tm = np.eye(4, dtype=np.float32); a = rng.normal(0, 0.2, 3)
        Tm[:3, :3] = synth._rot(*a).astype(np.float32); Tm[:3, 3] = rng.normal(0, 0.5, 3).astype(np.float32)
        Ow = (-(Tm[:3, :3].astype(np.float64).T @ Tm[:3, 3].astype(np.float64))).astype(np.float32)
        fr = dict(Tcw=Tm, Ow=Ow, fx=517.3, fy=516.5, cx=318.6, cy=255.3, mbf=40.0, min_x=0, max_x=640, min_y=0, max_y=480, log_scale_factor=np.log(np.float32(1.2)))
        lim = float(rng.choice([0.5, 0.8]))
        a1, a2 = FR.is_in_frustum(fr, pts, lim), fb.is_in_frustum(fr, pts, lim)
        check("frustum", all(np.array_equal(a1[k], a2[k]) for k in a2), dict(m=m, lim=lim))
    except Exception as e:
        bad += 1
        print("EXCEPTION %s: %s: %s" % (kw, type(e).__name__, e), flush=True)
print("match / frame sweep: %d cases, %d calls, %d mismatches" % (N, calls, bad))
