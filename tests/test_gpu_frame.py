"""GPU: the Frame glue (eao_frame_is_in_frustum, eao_assign_features_to_grid, eao_compute_stereo_from_rgbd) against the CPU
oracle -- bit-exact tables (reference src/Frame.cc:597-614, 638-695, 751-761, 1016-1037)."""
import numpy as np
import pytest

from eao_fusion_amd import frame as F
from eao_fusion_amd import synth

from test_oracle_frame import frustum_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def both(oracle):
    import torch  # noqa: F401  (first, so that the library resolves the same HIP runtime)
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    return F.product(), oracle.frame_binding(), E


@pytest.mark.parametrize("kw", [dict(), dict(n=1200, seed=8001, flip=0.09, mono_frac=0.6), dict(n=300, seed=8002, clutter=0.5, n_nodes=12)])
def test_is_in_frustum(both, kw):
    g, o, _ = both
    scene = synth.synth_search_scene(**kw)
    for seed in (0, 1):
        fr, pts = frustum_case(scene, seed)
        for lim in (0.5, 0.8):
            a, b = g.is_in_frustum(fr, pts, lim), o.is_in_frustum(fr, pts, lim)
            assert 20 < int(b["in_view"].sum()) < len(b["in_view"])
            for k in b:
                assert np.array_equal(a[k], b[k]), k


def test_is_in_frustum_many_points(both):
    """a local map of 20 000 points (several workgroups, ragged tail)"""
    g, o, _ = both
    rng = np.random.default_rng(11)
    n = 20001
    X = rng.uniform([-8, -5, -2], [8, 5, 14], (n, 3)).astype(np.float32)
    nrm = X / np.linalg.norm(X, axis=1, keepdims=True)          # mean viewing direction: from the camera to the point
    nrm = (nrm + rng.normal(0, 0.4, (n, 3))).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    d = np.linalg.norm(X, axis=1).astype(np.float32)
    pts = dict(active=np.ones(n, np.uint8), Xw=X, normal=nrm.astype(np.float32), min_dist_inv=(d * rng.uniform(0.3, 1.1, n)).astype(np.float32),
               max_dist_inv=(d * rng.uniform(0.9, 3.0, n)).astype(np.float32), max_dist=(d * rng.uniform(1.0, 4.0, n)).astype(np.float32),
               descriptors=np.zeros((n, 32), np.uint8))
    fr = dict(Tcw=np.eye(4, dtype=np.float32), Ow=np.zeros(3, np.float32), fx=517.3, fy=516.5, cx=318.6, cy=255.3, mbf=40.0, min_x=0, max_x=640,
              min_y=0, max_y=480, log_scale_factor=np.log(np.float32(1.2)))
    a, b = g.is_in_frustum(fr, pts, 0.5), o.is_in_frustum(fr, pts, 0.5)
    assert 500 < int(b["in_view"].sum()) < n
    for k in b:
        assert np.array_equal(a[k], b[k]), k
    e = g.is_in_frustum(fr, {k: v[:0] for k, v in pts.items()}, 0.5)
    assert len(e["in_view"]) == 0


def test_frustum_feeds_search_by_projection(both, oracle):
    """Tracking::SearchLocalPoints end to end: isInFrustum -> SearchByProjection(F, vpMapPoints, th)"""
    g, o, E = both
    scene = synth.synth_search_scene()
    fr, pts = frustum_case(scene, 3, turn=0.0)      # the scene's own second camera: its keypoints observe these points
    outs = [b.is_in_frustum(fr, pts, 0.5) for b in (g, o)]
    nl = len(scene["K2"]["scale_factors"])
    res = []
    for r, backend in zip(outs, ("gpu", "cpu")):
        skip = ((r["in_view"] == 0) | (r["pred_level"] < 0) | (r["pred_level"] >= nl)).astype(np.uint8)
        mps = dict(proj_x=r["proj_x"], proj_y=r["proj_y"], proj_xr=r["proj_xr"], view_cos=r["view_cos"], level=np.clip(r["pred_level"], 0, nl - 1),
                   descriptors=pts["descriptors"], skip=skip)
        if backend == "gpu":
            res.append(E.ORBmatcher(0.8).SearchByProjectionPoints(scene["K2"], mps, 3.0))
        else:
            res.append(oracle.search_by_projection_points(scene["K2"], mps, 3.0, 0.8))
    assert res[0][0] == res[1][0] and res[0][0] > 20
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("n,spread", [(2000, 20), (1, 0), (64, 300), (4097, 5), (30000, 40)])
def test_assign_features_to_grid(both, n, spread):
    g, o, _ = both
    rng = np.random.default_rng(n)
    kx = rng.uniform(-spread, 640 + spread, n).astype(np.float32)
    ky = rng.uniform(-spread, 480 + spread, n).astype(np.float32)
    kx[: n // 10] = np.round(kx[: n // 10] / 10) * 10 + 5          # cell boundaries: round() half away from zero
    a, b = g.assign_features_to_grid(kx, ky, 0, 0, 640, 480), o.assign_features_to_grid(kx, ky, 0, 0, 640, 480)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # other grid shapes / bounds (undistorted image bounds are not integers)
    a, b = g.assign_features_to_grid(kx, ky, -3.7, -2.2, 651.3, 489.9, 32, 24), o.assign_features_to_grid(kx, ky, -3.7, -2.2, 651.3, 489.9, 32, 24)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_assign_features_to_grid_limits(both):
    g, _, E = both
    s, it = g.assign_features_to_grid(np.zeros(0, np.float32), np.zeros(0, np.float32), 0, 0, 640, 480)
    assert s.sum() == 0 and len(it) == 0
    s, it = g.assign_features_to_grid(np.full(10, -500, np.float32), np.full(10, -500, np.float32), 0, 0, 640, 480)   # all outside
    assert s[-1] == 0 and len(it) == 0
    with pytest.raises(E.EaoError):
        g.assign_features_to_grid(np.zeros(32769, np.float32), np.zeros(32769, np.float32), 0, 0, 640, 480)


def test_stereo_from_rgbd(both):
    g, o, E = both
    rng = np.random.default_rng(6)
    depth = rng.uniform(0.3, 8, (480, 640)).astype(np.float32)
    depth[rng.random((480, 640)) < 0.2] = 0
    depth[rng.random((480, 640)) < 0.02] = -1
    for n in (1500, 1, 0):
        kx = rng.uniform(0, 639.9, n).astype(np.float32)
        ky = rng.uniform(0, 479.9, n).astype(np.float32)
        ku = (kx + rng.normal(0, 0.5, n)).astype(np.float32)
        a, b = g.compute_stereo_from_rgbd(kx, ky, ku, depth, 40.0), o.compute_stereo_from_rgbd(kx, ky, ku, depth, 40.0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(E.EaoError):     # cv::Mat::at would read out of bounds: refused
        g.compute_stereo_from_rgbd(np.array([640.5], np.float32), np.array([3], np.float32), np.array([640.5], np.float32), depth, 40.0)


@pytest.mark.parametrize("cam", ["TUM1", "TUM2"])
def test_undistort_keypoints(both, cam):
    """Frame::UndistortKeyPoints / ComputeImageBounds on the device (k_undistort): the same floats as the oracle's restatement of cv::undistortPoints, for the
    reference's two distorted cameras with four and five coefficients, keypoints inside, on and outside the image."""
    from golden_cases import TUM_CAMERAS
    g, o, E = both
    K, D = TUM_CAMERAS[cam]
    rng = np.random.default_rng(32)
    for n in (5000, 257, 1, 0):
        x = rng.uniform(-40, 680, n).astype(np.float32); y = rng.uniform(-40, 520, n).astype(np.float32)
        for d in (D, D[:4], (0.0, 0.2, 0.0, 0.0), ()):
            a, b = g.undistort_keypoints(x, y, *K, d), o.undistort_keypoints(x, y, *K, d)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for d in (D, D[:4], ()):
        assert np.array_equal(g.compute_image_bounds(640, 480, *K, d), o.compute_image_bounds(640, 480, *K, d))
        assert np.array_equal(g.compute_image_bounds(752, 480, *K, d), o.compute_image_bounds(752, 480, *K, d))
    with pytest.raises(E.EaoError):
        g.undistort_keypoints(np.zeros(3, np.float32), np.zeros(3, np.float32), 0.0, 500.0, 320.0, 240.0, D)
