"""Golden cases: the hot path's entry points on fixed seeded inputs, written ONCE by tools/gen_golden.py (CPU oracle, build
container) into tests/golden/*.npz and from then on compared with BOTH implementations:

  tests/test_golden.py::test_oracle_matches_golden   (CPU suite)  -- the oracle of the current tree against the frozen vectors
  tests/test_golden.py::test_gpu_matches_golden      (-m gpu)      -- the HIP path against the same frozen vectors

so that an edit which moves oracle and kernels TOGETHER (VERDICT r2 missing #2: "they can drift together unnoticed") fails
here.  Parity is still unpinned by the reference (no tests / fixtures upstream, sources unbuildable here -- DESIGN.md section 2):
these vectors freeze this repository's restatement, they do not come from a run of the reference.

Every case is a function of an `api` object (OracleApi / ProductApi below: the same calls on the two implementations) that
returns a flat dict of numpy arrays.  Keys that start with "f:" are floating-point LM results compared within the LM bound
(1e-4 of the update, tests/test_gpu_lm.py); "t:" keys are LM traces (trials exact, chi2 1e-6, lambda 5e-4 relative, over the
well-conditioned prefix -- tests/test_gpu_lm.py::_check_trace); "in:" keys are a
digest of the generated inputs (a mismatch there means the GENERATOR changed, not the path); everything else is bit-exact."""
import hashlib

import numpy as np

from eao_fusion_amd import synth

ORB_CFG = (1000, 1.2, 8, 20, 7)


def _digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def _dict_digest(d):
    return _digest(*[np.asarray(d[k]) for k in sorted(d) if isinstance(d[k], (np.ndarray, np.generic, float, int))])


# --------------------------------------------------------------------------------------------------------------- the two APIs
class OracleApi:
    name = "oracle"

    def __init__(self, O):
        self.O = O
        self.search = O.search_binding()
        self.frame = O.frame_binding()

    def orb(self, img, cfg=ORB_CFG):
        return self.O.OrbOracle(*cfg).extract(img)

    def hamming_best2(self, a, b, mask=None):
        return self.O.hamming_best2(a, b, mask)

    def hamming_matrix(self, a, b):
        return self.O.hamming_matrix(a, b)

    def pose(self, p):
        return self.O.pose_optimization(p)

    def local_ba(self, p):
        return self.O.local_ba(p)

    def bundle_adjustment(self, p, iterations, robust):
        return self.O.bundle_adjustment(p, iterations, robust)

    def search_points(self, cur, mps, th, nnratio):
        return self.O.search_by_projection_points(cur, mps, th, nnratio)

    def search_frames(self, cur, last, th, mono, check, nnratio):
        return self.O.search_by_projection_frames(cur, last, th, mono, check)

    def distinct(self, sets):
        return self.O.distinctive_descriptors(sets)

    def stereo(self, left, right, mb, mbf, cfg=ORB_CFG):
        ol, orr = self.O.OrbOracle(*cfg), self.O.OrbOracle(*cfg)
        kl, dl = ol.extract(left)
        kr, dr = orr.extract(right)
        return self.O.stereo_matches(ol, orr, kl, dl, kr, dr, mb, mbf)


class ProductApi:
    name = "gpu"

    def __init__(self, E):
        from eao_fusion_amd import frame as FR
        from eao_fusion_amd import search as S
        self.E = E
        self.search = S.product()
        self.frame = FR.product()

    def orb(self, img, cfg=ORB_CFG):
        return self.E.ORBextractor(*cfg)(img)

    def hamming_best2(self, a, b, mask=None):
        return self.E.hamming_best2(a, b, mask)

    def hamming_matrix(self, a, b):
        return self.E.hamming_matrix(a, b)

    def pose(self, p):
        return self.E.Optimizer.PoseOptimization(p)

    def local_ba(self, p):
        return self.E.Optimizer.LocalBundleAdjustment(p)

    def bundle_adjustment(self, p, iterations, robust):
        return self.E.Optimizer.BundleAdjustment(p, iterations, bRobust=robust)

    def search_points(self, cur, mps, th, nnratio):
        return self.E.ORBmatcher(nnratio, True).SearchByProjectionPoints(cur, mps, th)

    def search_frames(self, cur, last, th, mono, check, nnratio):
        return self.E.ORBmatcher(nnratio, check).SearchByProjectionFrames(cur, last, th, mono)

    def distinct(self, sets):
        return self.E.distinctive_descriptors(sets)

    def stereo(self, left, right, mb, mbf, cfg=ORB_CFG):
        el, er = self.E.ORBextractor(*cfg), self.E.ORBextractor(*cfg)
        kl, dl = el(left)
        kr, dr = er(right)
        return self.E.compute_stereo_matches(el, er, kl, dl, kr, dr, mb, mbf)


# --------------------------------------------------------------------------------------------------------------- the cases
def _orb_case(seed, n_rect, n_small):
    def run(api):
        img = synth.synth_frame(seed, n_rect=n_rect, n_small=n_small)
        kps, desc = api.orb(img)
        return {"in:image": _digest(img), "keypoints": np.ascontiguousarray(kps).view(np.uint8).reshape(len(kps), -1), "descriptors": desc}
    return run


def _trace(prefix, tr, out):
    out["t:%schi2" % prefix] = np.asarray(tr["chi2"], np.float64)
    out["t:%slam" % prefix] = np.asarray(tr["lam"], np.float64)
    out["t:%strials" % prefix] = np.asarray(tr["trials"], np.int32)


def _pose_case(**kw):
    def run(api):
        p = synth.synth_pose(**kw)
        r = api.pose(p)
        out = {"in:problem": _dict_digest(p), "f:Tcw": r["Tcw"], "f0:Tcw": np.asarray(p["Tcw"], np.float32), "outlier": np.asarray(r["outlier"], np.uint8),
               "n_inliers": np.array([r["n_inliers"]], np.int32)}
        if "plane_outlier" in r:
            out["plane_outlier"] = np.asarray(r["plane_outlier"], np.uint8)
        _trace("", r["trace"], out)
        return out
    return run


def _lba_case(**kw):
    def run(api):
        p = synth.synth_ba(**kw)
        r = api.local_ba(p)
        out = {"in:problem": _dict_digest(p), "f:poses": r["poses"], "f0:poses": p["poses"], "f:points": r["points"], "f0:points": p["points"],
               "iters": np.asarray(r["iters"], np.int32), "edge_outlier": np.asarray(r["edge_outlier"], np.uint8)}
        _trace("", r["trace"], out)
        return out
    return run


def _gba_case(planes, iterations, robust, **kw):
    def run(api):
        p = synth.synth_ba(**kw)
        p["fixed"] = (np.arange(len(p["poses"])) == 0).astype(np.uint8)      # only keyframe 0 is fixed (src/Optimizer.cc:85)
        if planes:
            p = synth.add_ba_planes(p, n_planes=planes, seed=7001)
        r = api.bundle_adjustment(p, iterations, robust)
        out = {"in:problem": _dict_digest(p), "f:poses": r["poses"], "f0:poses": p["poses"], "f:points": r["points"], "f0:points": p["points"],
               "iters": np.asarray(r["iters"], np.int32)}
        if planes:
            out["f:planes"] = r["planes"]
            out["f0:planes"] = np.asarray(p["planes"], np.float32)
        _trace("", r["trace"], out)
        return out
    return run


def _hamming(api):
    a, b, perm = synth.synth_descriptors_planted(1000)
    rng = np.random.default_rng(9)
    mask = (rng.random((300, 1000)) < 0.05).astype(np.uint8)
    mask[5] = 0
    D = api.hamming_matrix(a, b)
    return {"in:descriptors": _digest(a, b, mask), "best2": np.ascontiguousarray(api.hamming_best2(a, b)).view(np.uint8),
            "best2_masked": np.ascontiguousarray(api.hamming_best2(a[:300], b, mask)).view(np.uint8),
            "matrix_digest": _digest(np.asarray(D, np.uint16)), "matrix_corner": np.asarray(D, np.uint16)[:48, :48].copy(),
            "matrix_row_sums": np.asarray(D, np.int64).sum(1)}


def _tracking_searches(api):
    out = {}
    cur, last, mps = synth.synth_tracking(n=1000, seed=7003)
    nm, m = api.search_points(cur, mps, 3.0, 0.8)
    out.update({"in:tracking": _digest(cur["kp_x"], cur["descriptors"], mps["proj_x"], mps["descriptors"]), "points_n": np.array([nm], np.int32), "points_match": m})
    cur, last, mps = synth.synth_tracking(n=1000, seed=7006, moved=0.3)
    for check in (True, False):
        nm, m = api.search_frames(cur, last, 15.0, False, check, 0.9)
        out["frames_n_%d" % check] = np.array([nm], np.int32)
        out["frames_match_%d" % check] = m
    return out


def _guided_searches(api):
    g = api.search
    sc = synth.synth_search_scene(n=1200, seed=8001, flip=0.09, mono_frac=0.6)
    out = {"in:scene": _digest(sc["K1"]["kp_x"], sc["K1"]["descriptors"], sc["K2"]["kp_x"], sc["K2"]["descriptors"], sc["points"]["Xw"], sc["points"]["descriptors"])}

    def put(name, res):
        out[name + "_n"] = np.array([res[0]], np.int32)
        for i, a in enumerate(res[1:]):
            out["%s_%d" % (name, i)] = np.asarray(a)
    kf = dict(sc["K2"])
    kf["occupied"] = (np.arange(len(kf["kp_x"])) % 13 == 0).astype(np.uint8)
    put("projection_sim3", g.search_by_projection_sim3(kf, sc["Scw"], sc["K"], sc["points"], 10))
    cur = dict(sc["K2"])
    cur["occupied"] = (np.arange(len(cur["kp_x"])) % 11 == 0).astype(np.uint8)
    P = sc["points"]
    ang = ((np.arange(len(P["active"])) * 37) % 360).astype(np.float32)
    put("projection_kf", g.search_by_projection_kf(cur, sc["T2w"], sc["K"], P, ang, 15, 100, True))

    def side(K, mp, fv):
        return dict(descriptors=K["descriptors"], angle=K["kp_angle"], valid=(mp >= 0).astype(np.uint8), fv=fv)
    s1, s2 = side(sc["K1"], sc["mp1"], sc["fv1"]), side(sc["K2"], sc["mp2"], sc["fv2"])
    put("bow_kf_frame", g.search_by_bow(0, s1, s2, 0.7, True))
    put("bow_kf_kf", g.search_by_bow(1, s1, s2, 0.75, True))
    k1, k2 = dict(sc["K1"]), dict(sc["K2"])
    k1["occupied"] = ((sc["mp1"] >= 0) & (np.arange(len(sc["mp1"])) % 2 == 0)).astype(np.uint8)
    k2["occupied"] = ((sc["mp2"] >= 0) & (np.arange(len(sc["mp2"])) % 3 == 0)).astype(np.uint8)
    put("triangulation", g.search_for_triangulation(k1, sc["fv1"], k2, sc["fv2"], sc["F12"], sc["ex"], sc["ey"], 0, True))
    pm = np.stack([sc["K1"]["kp_x"], sc["K1"]["kp_y"]], 1)
    put("initialization", g.search_for_initialization(sc["K1"], sc["K2"], pm, 100, 0.9, True))
    T = sc["T2w"].astype(np.float64)
    pose = np.concatenate([T[:3, :3].ravel(), T[:3, 3], -T[:3, :3].T @ T[:3, 3]]).astype(np.float32)
    put("fuse_pose", g.fuse_search(sc["K2"], 0, pose, sc["K"], sc["bf"], sc["points"], 3.0))
    put("fuse_sim3", g.fuse_search(sc["K2"], 1, sc["Scw"], sc["K"], sc["bf"], sc["points"], 3.0))

    def pts_of(mp):
        idx = np.maximum(mp, 0)
        d = {k: np.ascontiguousarray(P[k][idx]) for k in ("Xw", "normal", "min_dist_inv", "max_dist_inv", "max_dist", "descriptors")}
        d["active"] = ((mp >= 0) & (P["active"][idx] > 0)).astype(np.uint8)
        return d
    put("sim3", g.search_by_sim3(sc["K1"], sc["T1w"], pts_of(sc["mp1"]), sc["K2"], sc["T2w"], pts_of(sc["mp2"]), sc["K"], 1.0, sc["R12"], sc["t12"], 7.5))
    return out


def _frame_glue(api):
    rng = np.random.default_rng(8800)
    sc = synth.synth_search_scene(n=900, seed=8003)
    P = sc["points"]
    T = np.ascontiguousarray(sc["T2w"], np.float32)
    Ow = (-(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))).astype(np.float32)
    K = sc["K"]
    fr = dict(Tcw=T, Ow=Ow, fx=K[0], fy=K[1], cx=K[2], cy=K[3], mbf=sc["bf"], min_x=0.0, max_x=640.0, min_y=0.0, max_y=480.0,
              log_scale_factor=np.float32(np.log(np.float32(1.2))))
    fo = api.frame.is_in_frustum(fr, P, 0.5)
    out = {"in:scene": _digest(P["Xw"], P["normal"], T)}
    inv = fo["in_view"].astype(bool)
    out["frustum_in_view"] = np.asarray(fo["in_view"], np.uint8)
    for k in ("proj_x", "proj_y", "proj_xr", "view_cos", "pred_level"):
        out["frustum_" + k] = np.where(inv, fo[k], 0).astype(fo[k].dtype)      # only defined for points in view
    kx, ky = sc["K2"]["kp_x"], sc["K2"]["kp_y"]
    start, items = api.frame.assign_features_to_grid(kx, ky, -3.5, -2.25, 644.0, 482.5)
    out["grid_start"], out["grid_items"] = np.asarray(start), np.asarray(items)
    depth = rng.uniform(0.5, 6.0, (480, 640)).astype(np.float32)
    depth[rng.random((480, 640)) < 0.2] = 0
    ur, dz = api.frame.compute_stereo_from_rgbd(kx, ky, kx, depth, np.float32(40.0))
    out["rgbd_u_right"], out["rgbd_depth"] = ur, dz
    return out


# the reference's own camera files: ros_test/config/TUM1.yaml:8-17, TUM2.yaml:8-17 (fx, fy, cx, cy | k1, k2, p1, p2, k3)
TUM_CAMERAS = {"TUM1": ((517.306408, 516.469215, 318.643040, 255.313989), (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)),
               "TUM2": ((520.908620, 521.007327, 325.141442, 249.701764), (0.231222, -0.784899, -0.003257, -0.000105, 0.917205))}


def _undistort(api):
    """Frame::UndistortKeyPoints / ComputeImageBounds with the reference's two distorted cameras: random keypoints, the image border, and the
    four-coefficient form (k3 = 0: src/Tracking.cc:95-100 only appends a non-zero k3)."""
    rng = np.random.default_rng(8900)
    x = np.concatenate([rng.uniform(0, 640, 1500), np.linspace(0, 640, 33), np.zeros(9), np.full(9, 640.0)]).astype(np.float32)
    y = np.concatenate([rng.uniform(0, 480, 1500), np.zeros(33), np.linspace(0, 480, 9), np.linspace(0, 480, 9)]).astype(np.float32)
    out = {"in:points": _digest(x, y)}
    for name, (K, D) in TUM_CAMERAS.items():
        for nc in (4, 5):
            ux, uy = api.frame.undistort_keypoints(x, y, *K, D[:nc])
            out["%s_k%d_x" % (name, nc)], out["%s_k%d_y" % (name, nc)] = ux, uy
            out["%s_k%d_bounds" % (name, nc)] = api.frame.compute_image_bounds(640, 480, *K, D[:nc])
    ux, uy = api.frame.undistort_keypoints(x, y, *TUM_CAMERAS["TUM1"][0], (0.0, 0.1, 0.0, 0.0))      # k1 == 0: upstream copies mvKeys whatever the rest says
    out["k1_zero_x"], out["k1_zero_y"] = ux, uy
    out["k1_zero_bounds"] = api.frame.compute_image_bounds(640, 480, *TUM_CAMERAS["TUM1"][0], (0.0, 0.1, 0.0, 0.0))
    return out


def _distinct_and_stereo(api):
    rng = np.random.default_rng(78)
    sets = []
    for n in list(rng.integers(0, 40, 120)) + [1, 64, 65, 257]:
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        flips = (rng.random((int(n), 256)) < 0.1)
        sets.append(np.bitwise_xor(np.packbits(flips, axis=1), base) if n else np.zeros((0, 32), np.uint8))
    sl, sr = synth.synth_stereo_pair(9000)
    ur, dp = api.stereo(sl, sr, np.float32(40.0) / np.float32(535.4), np.float32(40.0))
    return {"in:sets": _digest(*sets), "in:stereo": _digest(sl, sr), "distinct_best": api.distinct(sets), "stereo_u_right": ur, "stereo_depth": dp}


CASES = {
    # keypoints + descriptors of four seeded frames, incl. the low-texture ones that exercise the minThFAST retry
    "orb_frame_1000": _orb_case(1000, 400, 1000),
    "orb_frame_1001": _orb_case(1001, 400, 1000),
    "orb_frame_lowtex_7": _orb_case(7, 40, 0),
    "orb_frame_lowtex_8": _orb_case(8, 5, 0),
    "hamming_planted_1000": _hamming,
    "pose_default": _pose_case(),
    "pose_planes": _pose_case(n=300, seed=4200, n_planes=6),
    "pose_mono": _pose_case(n=50, seed=4001, mono_frac=1.0),
    # BASELINE configs[3]: 20 free + 4 fixed keyframes x 3000 map points, 5 + 10 iterations with the outlier pass
    "lba_configs3": _lba_case(),
    "lba_small_mono": _lba_case(n_free=5, n_fixed=2, n_points=300, mono_frac=0.4, seed=3001),
    "gba_points": _gba_case(0, 10, False, n_free=6, n_fixed=3, n_points=400, seed=3100, mono_frac=0.2),
    "gba_planes": _gba_case(4, 10, True, n_free=6, n_fixed=3, n_points=400, seed=3100, mono_frac=0.2),
    # round 6: a map beyond the register-tile solver -- 60 keyframes on a band, the map-scale path (block-sparse tiles, nested-dissection order) against the oracle's dense solve
    "gba_map_band60": _gba_case(0, 6, False, n_free=60, n_fixed=1, n_points=2400, seed=5780, band=7),
    "search_tracking": _tracking_searches,
    "search_guided": _guided_searches,
    "frame_glue": _frame_glue,
    "undistort": _undistort,
    "distinct_and_stereo": _distinct_and_stereo,
}


# --------------------------------------------------------------------------------------------------------------- comparison
def compare(name, got, want, lm_rel=1e-4, exact_floats=False):
    """got: a fresh run; want: the frozen vectors (np.load of the .npz).  Raises AssertionError with the first difference."""
    assert set(got) == set(want.keys()), "%s: keys differ: %s" % (name, sorted(set(got) ^ set(want.keys())))
    for k in sorted(got):
        g, w = np.asarray(got[k]), np.asarray(want[k])
        if k.startswith("in:"):
            assert np.array_equal(g, w), "%s/%s: the GENERATED INPUT differs from the one the vectors were made from (numpy / synth.py changed?)" % (name, k)
    for k in sorted(got):
        g, w = np.asarray(got[k]), np.asarray(want[k])
        if k.startswith("in:") or k.startswith("f0:"):
            continue
        assert g.shape == w.shape, "%s/%s: shape %s vs %s" % (name, k, g.shape, w.shape)
        if k.startswith("f:") and not exact_floats:
            old = np.asarray(want["f0:" + k[2:]], np.float64)
            upd = max(np.abs(w.astype(np.float64) - old).max(), 1e-6)
            err = np.abs(g.astype(np.float64) - w.astype(np.float64)).max()
            ulp = np.spacing(np.abs(w).max().astype(np.float32))
            assert err <= lm_rel * upd + 2 * ulp, "%s/%s: |new - golden| %.3e vs update %.3e" % (name, k, err, upd)
        elif k.startswith("t:") and not exact_floats:
            # LM trace (lambda, chi2, trials per iteration), compared on the well-conditioned prefix exactly as
            # tests/test_gpu_lm.py::_check_trace does: once chi2 stalls at the float32 noise floor, accept / reject is a coin flip
            base = k[2:]
            for suffix in ("lam", "trials", "chi2"):
                if base.endswith(suffix):
                    base = base[:-len(suffix)]
                    break
            chi = np.asarray(want["t:" + base + "chi2"], np.float64)
            n = len(chi)
            for i in range(len(chi)):
                if (i > 0 and abs(chi[i - 1] - chi[i]) <= 1e-6 * max(abs(chi[i - 1]), 1e-12)) or chi[i] < 1e-6:
                    n = i
                    break
            if k.endswith("trials"):
                assert np.array_equal(g[:n], w[:n]), "%s/%s: %s vs %s" % (name, k, g[:n], w[:n])
            else:
                assert np.allclose(g[:n], w[:n], rtol=5e-4 if k.endswith("lam") else 1e-6, atol=0), "%s/%s: %s vs %s" % (name, k, g[:n], w[:n])
        else:
            assert np.array_equal(g, w), "%s/%s differs from the golden vector (%d of %d entries)" % (name, k, int((g != w).sum()) if g.shape == w.shape else -1, g.size)
