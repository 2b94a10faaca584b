"""Known-answer tests for the LM oracle (oracle/lm_cpu.cpp).  Upstream has none, so the
arithmetic is pinned against maths: analytic Jacobians vs central differences on the manifold, the Schur
solution vs a full dense solve, Huber values, zero-noise convergence to ground truth."""
import ctypes as C

import numpy as np
import pytest

from eao_fusion_amd import synth

# intrinsics cross the boundary as float32 (Optimizer.cc:858-898): use the promoted values
FX, FY, CX, CY, BF = (float(np.float32(v)) for v in (535.4, 539.2, 320.1, 247.6, 40.0))


def _edge_eval(L, cam7, pt, obs, stereo):
    err = np.zeros(3); A = np.zeros((3, 3)); B = np.zeros((3, 6))
    L.orc_ba_edge_eval(cam7.ctypes.data, pt.ctypes.data, obs.ctypes.data, int(stereo), FX, FY, CX, CY, BF,
                       err.ctypes.data, A.ctypes.data, B.ctypes.data)
    return err, A, B


@pytest.mark.parametrize("stereo", [False, True])
def test_jacobians_vs_central_differences(oracle, stereo):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    T = synth.synth_pose()["Tcw_gt"]
    cam7 = np.zeros(7)
    L.orc_Tcw_to_cam7(np.ascontiguousarray(T, np.float32).ctypes.data, cam7.ctypes.data)
    D = 3 if stereo else 2
    for _ in range(10):
        pt = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(2, 5)])
        obs = np.array([300.0, 200.0, 290.0])
        _, A, B = _edge_eval(L, cam7, pt, obs, stereo)
        h = 1e-6 if not stereo else 1e-2   # float32 1/z quantises the stereo projection: needs a coarse step
        for j in range(3):
            dp = np.zeros(3); dp[j] = h
            ep, _, _ = _edge_eval(L, cam7, pt + dp, obs, stereo)
            em, _, _ = _edge_eval(L, cam7, pt - dp, obs, stereo)
            # stereo projection goes through a float32 1/z (types_six_dof_expmap.cpp:150-156): coarse tolerance there
            assert np.allclose((ep - em)[:D] / (2 * h), A[:D, j], rtol=2e-3 if stereo else 0, atol=5e-2 if stereo else 1e-5)
        hh = 1e-6 if not stereo else 1e-2
        for j in range(6):
            up = np.zeros(6); up[j] = hh
            cp = np.zeros(7); cm = np.zeros(7)
            L.orc_se3_oplus(cam7.ctypes.data, up.ctypes.data, cp.ctypes.data)
            L.orc_se3_oplus(cam7.ctypes.data, (-up).ctypes.data, cm.ctypes.data)
            ep, _, _ = _edge_eval(L, cp, pt, obs, stereo)
            em, _, _ = _edge_eval(L, cm, pt, obs, stereo)
            num = (ep - em)[:D] / (2 * hh)
            assert np.allclose(num, B[:D, j], rtol=5e-3 if stereo else 1e-5, atol=2e-1 if stereo else 1e-4)


def test_huber_values(oracle):
    L = oracle.lib()
    d = float(np.float32(np.sqrt(5.991)))
    out = np.zeros(2)
    L.orc_huber(d * d - 1e-9, d, out.ctypes.data)
    assert out[0] == pytest.approx(d * d - 1e-9) and out[1] == 1.0
    L.orc_huber(4 * d * d, d, out.ctypes.data)      # e = (2 delta)^2 : rho = 2*2d*d - d^2 = 3 d^2, rho' = 1/2
    assert out[0] == pytest.approx(3 * d * d) and out[1] == pytest.approx(0.5)


def test_zero_noise_ba_reaches_ground_truth(oracle):
    p = synth.synth_ba(n_free=6, n_fixed=2, n_points=300, sigma=0.0, outlier_frac=0.0)
    r = oracle.local_ba(p)
    assert np.abs(r["poses"] - p["poses_gt"]).max() < 2e-4
    # depth is weakly observable (7.5 cm baseline, float32 observations): compare against the 2 cm start error
    assert np.abs(r["points"] - p["points_gt"]).mean() < 0.1 * np.abs(p["points"] - p["points_gt"]).mean()
    assert r["edge_outlier"].sum() == 0
    assert np.all(np.diff(r["trace"]["chi2"][:5]) <= 0)


def test_zero_noise_pose_reaches_ground_truth(oracle):
    p = synth.synth_pose(n=300, sigma=0.0, outlier_frac=0.0)
    r = oracle.pose_optimization(p)
    assert np.abs(r["Tcw"] - p["Tcw_gt"]).max() < 1e-5
    assert r["n_inliers"] == 300 and r["outlier"].sum() == 0


def test_fixed_cameras_do_not_move_and_abort(oracle):
    p = synth.synth_ba(n_free=5, n_fixed=3, n_points=200)
    r = oracle.local_ba(p)
    f = p["fixed"].astype(bool)
    # fixed cameras pass through the float32 -> quaternion -> float32 boundary only
    assert np.abs(r["poses"][f] - p["poses"][f]).max() < 1e-6
    assert np.abs(r["poses"][~f] - p["poses"][~f]).max() > 1e-4
    r2 = oracle.local_ba(p, stop=True)          # pbStopFlag set on entry: silent return (Optimizer.cc:961-963)
    assert r2["aborted"] and list(r2["iters"]) == [0, 0]
    assert np.abs(r2["points"] - p["points"]).max() == 0


def test_schur_step_equals_full_dense_step(oracle):
    """One LM step of the oracle (Schur) vs a numpy full (poses+points) dense solve built from the oracle's
    own per-edge residuals/Jacobians: checks the reduction/back-substitution algebra."""
    L = oracle.lib()
    p = synth.synth_ba(n_free=3, n_fixed=1, n_points=40, outlier_frac=0.0)
    r = oracle.local_ba(p, its=(1, 0))
    # rebuild the normal equations in numpy at the initial state
    nc, npnt = len(p["poses"]), len(p["points"])
    cam7 = np.zeros((nc, 7))
    for c in range(nc):
        L.orc_Tcw_to_cam7(np.ascontiguousarray(p["poses"][c]).ctypes.data, cam7[c].ctypes.data)
    free = [c for c in range(nc) if not p["fixed"][c]]
    cidx = {c: i for i, c in enumerate(free)}
    n = 6 * len(free) + 3 * npnt
    H = np.zeros((n, n)); b = np.zeros(n)
    d = float(np.float32(np.sqrt(7.815)))
    pts = p["points"].astype(np.float64)
    for k in range(len(p["edge_cam"])):
        c, q = int(p["edge_cam"][k]), int(p["edge_point"][k])
        err, A, B = _edge_eval(L, cam7[c], pts[q], p["obs"][k].astype(np.float64), True)
        info = float(p["inv_sigma2"][k])
        chi = info * err @ err
        w = 1.0 if chi <= d * d else d / np.sqrt(chi)
        W = w * info
        lo = 6 * len(free) + 3 * q
        H[lo:lo + 3, lo:lo + 3] += W * A.T @ A
        b[lo:lo + 3] += -W * A.T @ err
        if c in cidx:
            po = 6 * cidx[c]
            H[po:po + 6, po:po + 6] += W * B.T @ B
            H[po:po + 6, lo:lo + 3] += W * B.T @ A
            H[lo:lo + 3, po:po + 6] += W * A.T @ B
            b[po:po + 6] += -W * B.T @ err
    lam = 1e-5 * np.abs(np.diag(H)).max()
    x = np.linalg.solve(H + lam * np.eye(n), b)
    new_pts = pts + x[6 * len(free):].reshape(-1, 3)
    assert r["trace"]["trials"][0] == 1
    assert np.allclose(r["points_d"], new_pts, rtol=0, atol=1e-9)
    for c in free:
        out = np.zeros(7)
        L.orc_se3_oplus(cam7[c].ctypes.data, x[6 * cidx[c]:6 * cidx[c] + 6].copy().ctypes.data, out.ctypes.data)
        assert np.allclose(r["cams_d"][c], out, rtol=0, atol=1e-10)


def test_pose_outlier_classification_consistent(oracle):
    p = synth.synth_pose()
    r = oracle.pose_optimization(p)
    assert r["n_inliers"] == len(p["points"]) - r["outlier"].sum()
    assert 0.05 < r["outlier"].mean() < 0.3
    assert np.abs(r["Tcw"] - p["Tcw_gt"]).max() < 5e-3
    assert oracle.pose_optimization({**p, "points": p["points"][:2], "obs": p["obs"][:2], "inv_sigma2": p["inv_sigma2"][:2]})["n_inliers"] == 0


def test_bundle_adjustment_restatement(oracle):
    """oracle/lm_cpu.cpp: orc_bundle_adjustment (src/Optimizer.cc:55-323 over keyframes and map points): zero noise reaches
    the ground truth with keyframe 0 alone fixed; it is the first pass of the local-BA restatement when the kernels and
    the Huber width agree (no outliers: every residual far below sqrt(5.99)); unobserved points do not move; robust
    kernels change the result when there are outliers."""
    p = synth.synth_ba(n_free=7, n_fixed=1, n_points=300, sigma=0.0, outlier_frac=0.0)
    r = oracle.bundle_adjustment(p, 10, robust=False)
    assert np.abs(r["poses"] - p["poses_gt"]).max() < 3e-4 and r["iters"][0] >= 3 and r["iters"][1] == 0
    p2 = synth.synth_ba(n_free=5, n_fixed=1, n_points=200, seed=3111, sigma=0.3, outlier_frac=0.0)
    a = oracle.bundle_adjustment(p2, 5, robust=True)
    b = oracle.local_ba(p2, its=(5, 0))
    assert np.abs(a["poses"] - b["poses"]).max() < 1e-5 and np.array_equal(a["iters"][:1], b["iters"][:1])
    p3 = synth.synth_ba(n_free=5, n_fixed=1, n_points=200, seed=3112, outlier_frac=0.1)
    p3["points"] = np.concatenate([p3["points"], np.array([[0.5, 0.5, 3.5]], np.float32)])
    rr, rn = oracle.bundle_adjustment(p3, 10, robust=True), oracle.bundle_adjustment(p3, 10, robust=False)
    assert np.array_equal(rr["points"][-1], p3["points"][-1]) and np.array_equal(rn["points"][-1], p3["points"][-1])
    assert np.abs(rr["poses"] - rn["poses"]).max() > 1e-4
    err_r = np.abs(rr["poses"] - p3["poses_gt"]).max()
    err_n = np.abs(rn["poses"] - p3["poses_gt"]).max()
    assert err_r < err_n                                  # the Huber kernels are what keeps 10 % outliers from dragging the poses


# ------------------------------------------------------------------ map planes in BundleAdjustment (src/Optimizer.cc:203-252)
def _np_plane_rotation(n):
    """Plane3D::rotation (src/g2oAddition/Plane3D.h:65-71) with scipy: Rz(azimuth) * Ry(-elevation)."""
    from scipy.spatial.transform import Rotation as Rot
    az = np.arctan2(n[1], n[0])
    el = np.arctan2(n[2], np.hypot(n[0], n[1]))
    return (Rot.from_rotvec([0, 0, az]) * Rot.from_rotvec([0, -el, 0])).as_matrix()


def _np_normalize(c):
    c = c / np.linalg.norm(c[:3])
    return -c if c[3] < 0 else c


def test_plane_oplus_and_error_against_numpy(oracle):
    O = oracle
    """Independent numpy / scipy restatement of Plane3D::oplus, operator*(Isometry3D, Plane3D) and ominus."""
    import ctypes as C
    L = O.lib()
    rng = np.random.default_rng(77)
    from scipy.spatial.transform import Rotation as Rot
    for _ in range(50):
        c = _np_normalize(np.concatenate([rng.normal(size=3), [rng.uniform(0.5, 4)]]))
        v = rng.normal(size=3) * 0.2
        out = np.zeros(4)
        L.orc_plane_oplus(C.c_void_p(c.ctypes.data), C.c_void_p(v.ctypes.data), C.c_void_p(out.ctypes.data))
        n = np.array([np.cos(v[1]) * np.cos(v[0]), np.cos(v[1]) * np.sin(v[0]), np.sin(v[1])])
        ref = _np_normalize(np.concatenate([_np_plane_rotation(c[:3]) @ n, [-((-c[3]) + v[2])]]))
        assert np.allclose(out, ref, atol=1e-12)
        zero = np.zeros(3)
        L.orc_plane_oplus(C.c_void_p(c.ctypes.data), C.c_void_p(zero.ctypes.data), C.c_void_p(out.ctypes.data))
        assert np.allclose(out, c, atol=1e-12)                       # a zero update leaves the plane where it is
        q = Rot.from_rotvec(rng.normal(size=3) * 0.4)
        t = rng.normal(size=3) * 0.5
        cam7 = np.concatenate([q.as_quat(), t])
        meas = _np_normalize(np.concatenate([rng.normal(size=3), [rng.uniform(0.5, 4)]]))
        err = np.zeros(3)
        L.orc_plane_error(C.c_void_p(cam7.ctypes.data), C.c_void_p(c.ctypes.data), C.c_void_p(meas.ctypes.data), C.c_void_p(err.ctypes.data))
        nl = q.as_matrix() @ c[:3]
        loc = _np_normalize(np.concatenate([nl, [c[3] - t @ nl]]))
        nn = _np_plane_rotation(loc[:3]).T @ meas[:3]
        ref = np.array([np.arctan2(nn[1], nn[0]), np.arctan2(nn[2], np.hypot(nn[0], nn[1])), (-loc[3]) - (-meas[3])])
        assert np.allclose(err, ref, atol=1e-12)
        # the measurement of the plane itself has zero error
        L.orc_plane_error(C.c_void_p(cam7.ctypes.data), C.c_void_p(c.ctypes.data), C.c_void_p(loc.ctypes.data), C.c_void_p(err.ctypes.data))
        assert np.abs(err).max() < 1e-12


def test_bundle_adjustment_with_planes_reaches_ground_truth(oracle):
    O = oracle
    """Noise-free observations of points and planes, perturbed start: BundleAdjustment with the MapPlane vertices brings
    poses, points AND planes back to the truth (error function, oplus and the numeric Jacobians are consistent)."""
    p = synth.synth_ba(n_free=6, n_fixed=2, n_points=300, seed=5401, sigma=0.0, outlier_frac=0.0)
    q = synth.add_ba_planes(p, n_planes=4, seed=7001, angle_noise_deg=0.0, dist_noise=0.0, outlier_edges=0)
    o = O.bundle_adjustment(q, 20, True)
    assert np.abs(o["planes"] - q["planes_gt"]).max() < 2e-5 and np.abs(q["planes"] - q["planes_gt"]).max() > 5e-3
    assert np.abs(o["points"] - p["points_gt"]).max() < 1e-3
    # the planes carry information: without them the same start ends elsewhere after the same iterations
    o0 = O.bundle_adjustment(p, 20, True)
    assert o["trace"]["chi2"][-1] < 1e-3 and o0["trace"]["chi2"][-1] < 1e-3
