"""CPU: the oracle's Frame glue (oracle/frame_cpu.cpp: isInFrustum, AssignFeaturesToGrid, ComputeStereoFromRGBD) against
independent numpy restatements of reference src/Frame.cc:597-614, 638-695, 751-761, 1016-1037 and hand-made cases."""
import numpy as np
import pytest

from eao_fusion_amd import frame as F
from eao_fusion_amd import synth


@pytest.fixture(scope="module")
def orc(oracle):
    return oracle.frame_binding()


def frustum_frame(scene, which="T2w"):
    T = np.ascontiguousarray(scene[which], np.float32)
    R, t = T[:3, :3], T[:3, 3]
    Ow = (-(R.astype(np.float64).T @ t.astype(np.float64))).astype(np.float32)   # UpdatePoseMatrices: mOw = -mRcw.t() * mtcw
    K = scene["K"]
    k2 = scene["K2"]
    return dict(Tcw=T, Ow=Ow, fx=K[0], fy=K[1], cx=K[2], cy=K[3], mbf=scene["bf"], min_x=k2["min_x"], max_x=k2["max_x"],
                min_y=k2["min_y"], max_y=k2["max_y"], log_scale_factor=k2["log_scale_factor"])


def frustum_case(scene, seed=0, turn=20.0):
    """the scene's second camera turned by 20 degrees (a third of the points leave the image), some normals turned away,
    some invariance ranges shrunk"""
    rng = np.random.default_rng(seed)
    a = np.deg2rad(turn)
    Ry = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
    sc = dict(scene)
    sc["T2w"] = (Ry @ scene["T2w"].astype(np.float64)).astype(np.float32)
    fr = frustum_frame(sc)
    pts = {k: np.array(v, copy=True) for k, v in scene["points"].items()}
    n = len(pts["Xw"])
    flip = rng.random(n) < 0.15
    pts["normal"][flip] *= -1
    pts["max_dist_inv"][rng.random(n) < 0.1] *= 0.4
    pts["min_dist_inv"][rng.random(n) < 0.1] *= 3.0
    return fr, pts


def np_frustum(fr, pts, lim):
    """numpy restatement, one point at a time, float32 where upstream is float"""
    f32 = np.float32
    T = fr["Tcw"]
    n = len(pts["Xw"])
    out = dict(in_view=np.zeros(n, np.uint8), proj_x=np.full(n, -1, f32), proj_y=np.full(n, -1, f32), proj_xr=np.full(n, -1, f32),
               view_cos=np.zeros(n, f32), pred_level=np.full(n, -1, np.int32))
    for i in range(n):
        P = pts["Xw"][i].astype(f32)
        Pc = (T[:3, :3].astype(np.float64) @ P.astype(np.float64) + T[:3, 3].astype(np.float64)).astype(f32)
        if Pc[2] < 0:
            continue
        invz = f32(1) / Pc[2]
        u = f32(f32(f32(fr["fx"]) * Pc[0]) * invz) + f32(fr["cx"])
        v = f32(f32(f32(fr["fy"]) * Pc[1]) * invz) + f32(fr["cy"])
        if u < fr["min_x"] or u > fr["max_x"] or v < fr["min_y"] or v > fr["max_y"]:
            continue
        PO = P - fr["Ow"].astype(f32)
        dist = f32(np.sqrt(np.sum(PO.astype(np.float64) ** 2)))
        if dist < pts["min_dist_inv"][i] or dist > pts["max_dist_inv"][i]:
            continue
        vc = f32(np.dot(PO.astype(np.float64), pts["normal"][i].astype(np.float64)) / np.float64(dist))
        if vc < f32(lim):
            continue
        ratio = f32(pts["max_dist"][i]) / dist
        lvl = int(np.ceil(f32(np.log(ratio)) / f32(fr["log_scale_factor"])))
        out["in_view"][i] = 1
        out["proj_x"][i], out["proj_y"][i], out["proj_xr"][i] = u, v, u - f32(f32(fr["mbf"]) * invz)
        out["view_cos"][i], out["pred_level"][i] = vc, lvl
    return out


@pytest.mark.parametrize("kw", [dict(), dict(n=300, seed=8002, clutter=0.5, n_nodes=12)])
def test_is_in_frustum_vs_numpy(orc, kw):
    scene = synth.synth_search_scene(**kw)
    fr, pts = frustum_case(scene)
    for lim in (0.5, 0.9):
        a = orc.is_in_frustum(fr, pts, lim)
        b = np_frustum(fr, pts, lim)
        assert a["in_view"].sum() > 20 and a["in_view"].sum() < len(a["in_view"])
        for k in a:
            assert np.array_equal(a[k], b[k]), k


def test_is_in_frustum_rejections(orc):
    """one point per early return of src/Frame.cc:652-680, identity pose"""
    fr = dict(Tcw=np.eye(4, dtype=np.float32), Ow=np.zeros(3, np.float32), fx=500, fy=500, cx=320, cy=240, mbf=40, min_x=0, max_x=640,
              min_y=0, max_y=480, log_scale_factor=np.log(np.float32(1.2)))
    X = np.array([[0, 0, -1], [5, 0, 1], [0, 5, 1], [0, 0, 0.05], [0, 0, 50], [0, 0, 2], [0, 0, 2]], np.float32)
    nrm = np.array([[0, 0, 1]] * 6 + [[1, 0, 0]], np.float32)
    pts = dict(active=np.ones(7, np.uint8), Xw=X, normal=nrm, min_dist_inv=np.full(7, 0.1, np.float32), max_dist_inv=np.full(7, 20, np.float32),
               max_dist=np.full(7, 16, np.float32), descriptors=np.zeros((7, 32), np.uint8))
    r = orc.is_in_frustum(fr, pts, 0.5)
    assert r["in_view"].tolist() == [0, 0, 0, 0, 0, 1, 0]      # behind, left/right, top/bottom, too close, too far, ok, grazing
    assert r["proj_x"][5] == 320 and r["proj_y"][5] == 240 and r["proj_xr"][5] == 300 and r["view_cos"][5] == 1
    assert r["pred_level"][5] == int(np.ceil(np.float32(np.log(np.float32(8.0))) / np.float32(np.log(np.float32(1.2)))))
    assert r["pred_level"][0] == -1 and r["proj_x"][1] == -1   # untouched


def test_assign_features_to_grid(orc):
    rng = np.random.default_rng(5)
    kx = rng.uniform(-20, 660, 3000).astype(np.float32)          # some keypoints fall outside after "undistortion"
    ky = rng.uniform(-20, 500, 3000).astype(np.float32)
    start, items = orc.assign_features_to_grid(kx, ky, 0, 0, 640, 480)
    inv_w, inv_h = np.float32(64) / np.float32(640), np.float32(48) / np.float32(480)
    grid = {}
    for i in range(len(kx)):
        px = int(np.floor(abs((kx[i] - np.float32(0)) * inv_w) + np.float32(0.5)) * np.sign((kx[i]) * inv_w))   # C round(): half away from zero
        py = int(np.floor(abs((ky[i] - np.float32(0)) * inv_h) + np.float32(0.5)) * np.sign((ky[i]) * inv_h))
        if 0 <= px < 64 and 0 <= py < 48:
            grid.setdefault(px * 48 + py, []).append(i)
    assert start[-1] == sum(len(v) for v in grid.values()) == len(items)
    for c in range(64 * 48):
        assert items[start[c]:start[c + 1]].tolist() == grid.get(c, [])


def test_stereo_from_rgbd(orc):
    rng = np.random.default_rng(6)
    depth = rng.uniform(0.3, 8, (480, 640)).astype(np.float32)
    depth[rng.random((480, 640)) < 0.2] = 0                      # invalid depth
    depth[10, 10] = -1
    kx = np.concatenate([rng.uniform(0, 639.9, 500), [10.7]]).astype(np.float32)
    ky = np.concatenate([rng.uniform(0, 479.9, 500), [10.2]]).astype(np.float32)
    ku = (kx + rng.normal(0, 0.5, len(kx))).astype(np.float32)
    ur, dz = orc.compute_stereo_from_rgbd(kx, ky, ku, depth, 40.0)
    d = depth[ky.astype(np.int32), kx.astype(np.int32)]          # at<float>(v, u): truncation
    ok = d > 0
    assert np.array_equal(dz, np.where(ok, d, np.float32(-1)))
    assert np.array_equal(ur, np.where(ok, ku - np.float32(40.0) / np.where(ok, d, 1).astype(np.float32), np.float32(-1)))
    assert dz[-1] == -1 and ok.sum() > 300


def _distort(xn, yn, D):
    """the forward model cv::undistortPoints inverts (OpenCV calib3d documentation): normalised undistorted -> normalised distorted"""
    k1, k2, p1, p2, k3 = (list(D) + [0.0])[:5]
    r2 = xn * xn + yn * yn
    rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    return xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn), yn * rad + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn


@pytest.mark.parametrize("cam", ["TUM1", "TUM2"])
@pytest.mark.parametrize("nc", [4, 5])
def test_undistort_keypoints(orc, cam, nc):
    """Frame::UndistortKeyPoints = cv::undistortPoints(., mK, mDistCoef, Mat(), mK) with the reference's own distorted cameras (ros_test/config/TUM1.yaml,
    TUM2.yaml): (1) an independent float64 numpy restatement of the five fixed-point iterations gives the same floats; (2) the forward distortion model
    applied to the result lands back on the keypoint (the iteration has converged to a few thousandths of a pixel inside the image's inner disc)."""
    from golden_cases import TUM_CAMERAS
    (fx, fy, cx, cy), D = TUM_CAMERAS[cam]
    D = D[:nc]
    rng = np.random.default_rng(31)
    x = rng.uniform(0, 640, 4000).astype(np.float32); y = rng.uniform(0, 480, 4000).astype(np.float32)
    ux, uy = orc.undistort_keypoints(x, y, fx, fy, cx, cy, D)
    fxd, fyd, cxd, cyd = (np.float64(np.float32(v)) for v in (fx, fy, cx, cy))
    k = [np.float64(np.float32(v)) for v in (list(D) + [0.0])[:5]]
    xn = (x.astype(np.float64) - cxd) * (1.0 / fxd); yn = (y.astype(np.float64) - cyd) * (1.0 / fyd)
    x0, y0 = xn.copy(), yn.copy()
    for _ in range(5):
        r2 = xn * xn + yn * yn
        ic = 1.0 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * xn * yn + k[3] * (r2 + 2 * xn * xn)
        dy = k[2] * (r2 + 2 * yn * yn) + 2 * k[3] * xn * yn
        xn, yn = (x0 - dx) * ic, (y0 - dy) * ic
    assert np.array_equal(ux, (fxd * xn + cxd).astype(np.float32)) and np.array_equal(uy, (fyd * yn + cyd).astype(np.float32))
    # round trip through the forward model, where the fixed point is well inside its basin (r < 0.45: three quarters of the image)
    un, vn = (ux.astype(np.float64) - cxd) / fxd, (uy.astype(np.float64) - cyd) / fyd
    xd, yd = _distort(un, vn, k)
    inner = (x0 * x0 + y0 * y0) < 0.45 ** 2
    assert inner.sum() > 2000
    assert np.abs(xd * fxd + cxd - x)[inner].max() < 0.02 and np.abs(yd * fyd + cyd - y)[inner].max() < 0.02
    assert np.abs(ux - x).max() > 1.0          # (the cameras do distort)


def test_undistort_identity_and_bounds(orc):
    from golden_cases import TUM_CAMERAS
    K, D = TUM_CAMERAS["TUM1"]
    x = np.array([0, 10.5, 639.25], np.float32); y = np.array([0, 200.75, 479.5], np.float32)
    for d in ((), (0.0, 0.3, 0.01, 0.01), (0.0, 0.0, 0.0, 0.0, 0.5)):      # mDistCoef.at<float>(0) == 0.0: mvKeysUn = mvKeys (src/Frame.cc:775-779)
        ux, uy = orc.undistort_keypoints(x, y, *K, d)
        assert np.array_equal(ux, x) and np.array_equal(uy, y)
        assert orc.compute_image_bounds(640, 480, *K, d).tolist() == [0.0, 640.0, 0.0, 480.0]
    b = orc.compute_image_bounds(640, 480, *K, D)                           # src/Frame.cc:808-833
    cx_, cy_ = orc.undistort_keypoints(np.array([0, 640, 0, 640], np.float32), np.array([0, 0, 480, 480], np.float32), *K, D)
    want = [max(min(cx_[0], cx_[2]), 0), min(max(cx_[1], cx_[3]), 640), max(min(cy_[0], cy_[1]), 0), min(max(cy_[2], cy_[3]), 480)]
    assert b.tolist() == [float(np.float32(v)) for v in want] and 0 < b[0] < 30 and 600 < b[1] < 640
