"""GPU parity: HIP Levenberg-Marquardt (PoseOptimization, LocalBundleAdjustment) vs the fp64 oracle.
Bar (BASELINE.json north_star): <= 1e-4 relative on the pose / point UPDATES; the LM control flow
(iterations, trials per iteration, lambda) and the inlier/outlier decisions must match."""
import numpy as np
import pytest

from eao_fusion_amd import synth
from lm_tolerances import (CHAOTIC_BAND, CHAOTIC_BANDS_ALLOWED, CHI2_REL, CHI2_REL_BATCH_VS_SINGLE, CHI2_REL_FAR_OFF, CHI2_REL_PLANES, LAMBDA_REL, LAMBDA_REL_FAR_OFF, LAMBDA_REL_FAR_OFF_MAP_SCALE, LEAVES_THE_BAR,
                           SCHEDULE_UNSTABLE, UPDATE_REL)

pytestmark = pytest.mark.gpu
REL = UPDATE_REL      # every bound of this file comes from tests/lm_tolerances.py (one table, the evidence beside each entry)


@pytest.fixture(scope="module")
def gpu():
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    return E


def _check_trace(r, o, rel=CHI2_REL, lam_rel=LAMBDA_REL):
    """LM control flow must match wherever it is well-conditioned: once chi2 stalls at the float32 noise floor
    (relative improvement < 1e-6) rho = (chi_old - chi_new)/scale is rounding noise and accept/reject is a coin flip
    in ANY implementation, so only the well-conditioned prefix of each optimize() call is compared."""
    tg, to = r["trace"], o["trace"]
    n = min(len(tg["chi2"]), len(to["chi2"]))
    prev = None
    for k in range(n):
        c = to["chi2"][k]
        stalled = prev is not None and abs(prev - c) <= 1e-6 * max(abs(prev), 1e-12)
        if stalled or c < 1e-6:
            break
        assert tg["trials"][k] == to["trials"][k], "trials differ at LM iteration %d" % k
        assert tg["chi2"][k] == pytest.approx(c, rel=rel), "chi2 differs at LM iteration %d" % k
        # (profiles/r02_lm_trace_sensitivity.txt: GPU vs oracle <= 5.1e-5 on these problems, where one float32 ulp on the inputs moves the
        #  ORACLE's own lambda by up to 8e-3 -- lambda is a function of rho, a ratio of small differences)
        assert tg["lam"][k] == pytest.approx(to["lam"][k], rel=lam_rel), "lambda differs at LM iteration %d" % k
        prev = c


def _check_updates(new_gpu, new_cpu, old, what):
    upd = np.abs(new_cpu.astype(np.float64) - old.astype(np.float64))
    scale = max(upd.max(), 1e-6)
    err = np.abs(new_gpu.astype(np.float64) - new_cpu.astype(np.float64)).max()
    # outputs are float32 (Converter::toCvMat): allow one float32 ulp of the value on top of the 1e-4 bound
    ulp = np.spacing(np.abs(new_cpu).max().astype(np.float32))
    assert err <= REL * scale + 2 * ulp, "%s: |gpu-cpu| %.3e vs update scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("kw", [dict(), dict(n_free=5, n_fixed=2, n_points=300), dict(mono_frac=0.4, seed=3001),
                                dict(n_free=3, n_fixed=0, n_points=120, seed=3002), dict(n_free=20, n_fixed=4, n_points=3000, sigma=0.0, outlier_frac=0.0),
                                # 42 unknowns (padded to 44 by the tile solver), its 30-keyframe limit, and the map-scale path beyond it
                                dict(n_free=7, n_fixed=2, n_points=400, seed=3003), dict(n_free=30, n_fixed=3, n_points=1500, seed=3004),
                                dict(n_free=34, n_fixed=2, n_points=1500, seed=3005)])
def test_local_ba_parity(gpu, oracle, kw):
    p = synth.synth_ba(**kw)
    if kw.get("n_fixed", 4) == 0:
        p["fixed"][0] = 1          # KeyFrame mnId == 0 is fixed by the reference (src/Optimizer.cc:780)
    r = gpu.Optimizer.LocalBundleAdjustment(p)
    o = oracle.local_ba(p)
    assert list(r["iters"]) == list(o["iters"])
    _check_trace(r, o)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    assert np.array_equal(r["edge_outlier"], o["edge_outlier"])
    f = p["fixed"].astype(bool)
    assert np.array_equal(r["poses"][f], o["poses"][f])


# (CHAOTIC_BAND, SCHEDULE_UNSTABLE, LEAVES_THE_BAR: tests/lm_tolerances.py)


@pytest.mark.parametrize("seed", list(range(3030, 3060)))
def test_local_ba_rejected_trials(gpu, oracle, seed):
    """Far-off starts (25 degrees, 0.8 m, 1 m on the points): LM trials get rejected in the first and / or the second
    optimize() -- the stream freezes and the host replays the iteration trial by trial (up to the 10-trial limit) -- and
    in some windows every edge ends up an outlier, so the second optimize() has nothing to do (g2o returns -1).  Two fixed
    cameras keep the scale observable: with one, the damped system is singular along the gauge and no two
    implementations agree on the step.  All thirty seeds 3030..3059: full parity (LM schedule, trace, updates within 1e-4,
    outlier table) on the 18 well-conditioned ones; on the 12 of CHAOTIC_BAND (points flipping behind cameras in the first
    iterations, chi2 around 1e6 -- one ulp on the inputs moves the oracle's own result by the listed amount and on three
    of them changes its iteration count) the result must be reproducible bit for bit and stay within four of the oracle's
    own one-ulp bands."""
    p = synth.synth_ba(n_free=5, n_fixed=2, n_points=200, seed=seed, rot_noise_deg=25, trans_noise=0.8, point_noise=1.0, mono_frac=0.7)
    r = gpu.Optimizer.LocalBundleAdjustment(p)
    o = oracle.local_ba(p)
    if seed in CHAOTIC_BAND:
        r2 = gpu.Optimizer.LocalBundleAdjustment(p)
        assert np.array_equal(r["poses"].view(np.uint32), r2["poses"].view(np.uint32)) and np.array_equal(r["points"].view(np.uint32), r2["points"].view(np.uint32))
        assert list(r["iters"]) == list(r2["iters"]) and np.array_equal(r["edge_outlier"], r2["edge_outlier"])
        #  the band loosens the UPDATE bound only.  The LM schedule and the outlier table are still compared with the
        # oracle wherever the oracle's own schedule survives the one-ulp perturbation (all but SCHEDULE_UNSTABLE) ...
        if seed not in SCHEDULE_UNSTABLE:
            assert list(r["iters"]) == list(o["iters"]), "LM schedule differs from the oracle's"
            assert np.array_equal(r["edge_outlier"], o["edge_outlier"]), "outlier table differs from the oracle's"
        # ... and a banded seed on which the GPU is in fact within the strict bar stays under the strict bar (the band is the
        # ceiling for the seeds in LEAVES_THE_BAR only): a regression on one of the others is a finding, not noise
        if seed not in LEAVES_THE_BAR:
            _check_updates(r["poses"], o["poses"], p["poses"], "poses")
            _check_updates(r["points"], o["points"], p["points"], "points")
            return
        bound = CHAOTIC_BANDS_ALLOWED * max(CHAOTIC_BAND[seed], REL)
        for key in ("poses", "points"):
            upd = max(np.abs(o[key].astype(np.float64) - p[key].astype(np.float64)).max(), 1e-6)
            err = np.abs(r[key].astype(np.float64) - o[key].astype(np.float64)).max()
            assert err <= bound * upd, "%s: |gpu-cpu| %.3e vs update %.3e (band %.1e)" % (key, err, upd, CHAOTIC_BAND[seed])
        return
    assert list(r["iters"]) == list(o["iters"])
    # monocular-only windows this far from the optimum are ill-conditioned: rounding differences grow.  lambda: 4e-3 since round 4 (2e-3 before) -- the
    # pair assembly works on Cholesky-scaled blocks now (csrc/lba.hip, ba_chol3) where upstream and the oracle multiply by an explicit 3 x 3 inverse, so
    # GPU and oracle differ in rounding from the first iteration on instead of from the first reordered sum: on seed 3037 the lambda of the 14th
    # iteration moved from 1.45e-3 (explicit inverse) to 3.07e-3 (Cholesky) off the oracle's while chi2 agrees to 9e-8 and the final points to 8e-6 of the
    # update (profiles/r04_lm_seed3037.txt); one float32 ulp on the inputs moves the ORACLE's own lambda by up to 8e-3 (profiles/r02_lm_trace_sensitivity.txt)
    _check_trace(r, o, rel=CHI2_REL_FAR_OFF, lam_rel=LAMBDA_REL_FAR_OFF)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    assert np.array_equal(r["edge_outlier"], o["edge_outlier"])


@pytest.mark.parametrize("seed", [3030, 3034, 3043, 3051])
def test_rejected_trials_on_the_map_scale_path(gpu, oracle, seed, monkeypatch):
    """The same far-off windows through the map-scale kernels (EAO_BA_SOLVER=big): rejected trials freeze the stream, the
    host replays them one by one with the dense factorisation, a failed / empty second pass is reported like g2o's."""
    monkeypatch.setenv("EAO_BA_SOLVER", "big")
    p = synth.synth_ba(n_free=5, n_fixed=2, n_points=200, seed=seed, rot_noise_deg=25, trans_noise=0.8, point_noise=1.0, mono_frac=0.7)
    r = gpu.Optimizer.LocalBundleAdjustment(p)
    o = oracle.local_ba(p)
    assert list(r["iters"]) == list(o["iters"])
    _check_trace(r, o, rel=CHI2_REL_FAR_OFF, lam_rel=LAMBDA_REL_FAR_OFF_MAP_SCALE)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    assert np.array_equal(r["edge_outlier"], o["edge_outlier"])


def test_local_ba_abort_flag(gpu):
    p = synth.synth_ba(n_free=5, n_fixed=2, n_points=200)
    r = gpu.Optimizer.LocalBundleAdjustment(p, stop=np.array([1], np.uint8))
    assert r["aborted"] and list(r["iters"]) == [0, 0]
    assert np.array_equal(r["points"], p["points"])


def test_local_ba_rejects_duplicate_edges(gpu):
    p = synth.synth_ba(n_free=3, n_fixed=1, n_points=50)
    p["edge_cam"] = np.concatenate([p["edge_cam"], p["edge_cam"][-1:]])
    p["edge_point"] = np.concatenate([p["edge_point"], p["edge_point"][-1:]])
    p["obs"] = np.concatenate([p["obs"], p["obs"][-1:]])
    p["inv_sigma2"] = np.concatenate([p["inv_sigma2"], p["inv_sigma2"][-1:]])
    if not p["fixed"][p["edge_cam"][-1]]:
        with pytest.raises(gpu.EaoError):
            gpu.Optimizer.LocalBundleAdjustment(p)


@pytest.mark.parametrize("kw", [dict(), dict(n=300, sigma=0.0, outlier_frac=0.0), dict(n=50, seed=4001, mono_frac=1.0),
                                dict(n=2000, seed=4002, mono_frac=0.0), dict(n=8, seed=4003),
                                # 2 and 4 register-resident edges per thread, and the global-memory variant beyond 2048
                                dict(n=1024, seed=4004), dict(n=1025, seed=4005), dict(n=2600, seed=4006)])
def test_pose_optimization_parity(gpu, oracle, kw):
    p = synth.synth_pose(**kw)
    r = gpu.Optimizer.PoseOptimization(p)
    o = oracle.pose_optimization(p)
    assert r["n_inliers"] == o["n_inliers"]
    assert np.array_equal(r["outlier"], o["outlier"])
    _check_trace(r, o)
    _check_updates(r["Tcw"], o["Tcw"], p["Tcw"], "Tcw")


@pytest.mark.parametrize("kw", [dict(n=300, seed=4200, n_planes=6), dict(n=12, seed=4201, n_planes=5, sigma=0.5, outlier_frac=0.0),
                                dict(n=1500, seed=4202, n_planes=32), dict(n=40, seed=4203, n_planes=1, mono_frac=1.0),
                                # fewer threads in the launch than 13 plane variants x planes (found by tools/sweep_pose.py: planes beyond the
                                # launch's thread count were never evaluated)
                                dict(n=54, seed=488437215, sigma=0.5, outlier_frac=0.1, mono_frac=0.0, n_planes=7),
                                dict(n=34, seed=866062170, sigma=0.5, outlier_frac=0.0, mono_frac=1.0, n_planes=6),
                                dict(n=27, seed=226023083, sigma=0.5, outlier_frac=0.3, mono_frac=0.3, n_planes=6),
                                dict(n=100, seed=4204, n_planes=32), dict(n=700, seed=4205, n_planes=25),
                                # beyond 2048 correspondences: the global-memory variant of the kernel carries the plane edges too
                                dict(n=2600, seed=4204, n_planes=6), dict(n=3000, seed=4205, n_planes=32, mono_frac=0.5)])
def test_pose_optimization_with_planes(gpu, oracle, kw):
    """Plane edges (src/Optimizer.cc:456-535): same inlier / outlier tables as the oracle, pose update within 1e-4.  The
    Jacobian of these edges is g2o's central difference with delta = 1e-9, i.e. it carries ~1e-7 of rounding noise in
    any implementation."""
    p = synth.synth_pose(**kw)
    r = gpu.Optimizer.PoseOptimization(p)
    o = oracle.pose_optimization(p)
    assert r["n_inliers"] == o["n_inliers"]
    assert np.array_equal(r["outlier"], o["outlier"]) and np.array_equal(r["plane_outlier"], o["plane_outlier"])
    _check_updates(r["Tcw"], o["Tcw"], p["Tcw"], "Tcw")
    if kw["n_planes"] >= 3:
        assert r["plane_outlier"][-1] == 1 and r["plane_outlier"][:-1].sum() == 0


def test_pose_optimization_batch_equals_single_calls(gpu):
    """eao_pose_optimization_batch (one workgroup per frame -- the candidate loop of Tracking::Relocalization,
    src/Tracking.cc:2786-2940): every frame's pose, outlier tables and return value are BIT-identical to its own
    eao_pose_optimization call, whatever the batch's mix (2 / 4 register edges per thread, planes, mono-only, the
    < 3 correspondences early-out, and a frame beyond 2048 correspondences that the batch hands to the single path)."""
    kws = [dict(), dict(n=300, sigma=0.0, outlier_frac=0.0), dict(n=50, seed=4001, mono_frac=1.0), dict(n=2000, seed=4002, mono_frac=0.0),
           dict(n=8, seed=4003), dict(n=1024, seed=4004), dict(n=1025, seed=4005), dict(n=2, seed=4007), dict(n=2600, seed=4006),
           dict(n=300, seed=4200, n_planes=6), dict(n=1500, seed=4202, n_planes=32), dict(n=40, seed=4203, n_planes=1, mono_frac=1.0)]
    kws += [dict(n=200 + 37 * k, seed=4300 + k) for k in range(20)]
    probs = [synth.synth_pose(**kw) for kw in kws]
    singles = [gpu.Optimizer.PoseOptimization(p) for p in probs]
    for order in (list(range(len(probs))), list(reversed(range(len(probs))))):
        outs = gpu.Optimizer.PoseOptimizationBatch([probs[i] for i in order])
        for o, i in zip(outs, order):
            s = singles[i]
            assert o["n_inliers"] == s["n_inliers"] and o["lm_iterations"] == s["lm_iterations"], kws[i]
            assert np.array_equal(o["outlier"], s["outlier"]), kws[i]
            assert np.array_equal(o["Tcw"].view(np.uint32), s["Tcw"].view(np.uint32)), kws[i]
            if "plane_outlier" in s:
                assert np.array_equal(o["plane_outlier"], s["plane_outlier"])
    assert gpu.Optimizer.PoseOptimizationBatch([]) == []


def test_pose_optimization_batch_parity(gpu, oracle):
    """The batch against the fp64 oracle directly (same bar as the single call)."""
    probs = [synth.synth_pose(n=150 + 61 * k, seed=4400 + k, mono_frac=0.1 * (k % 5)) for k in range(12)]
    outs = gpu.Optimizer.PoseOptimizationBatch(probs)
    for p, r in zip(probs, outs):
        o = oracle.pose_optimization(p)
        assert r["n_inliers"] == o["n_inliers"] and np.array_equal(r["outlier"], o["outlier"])
        _check_updates(r["Tcw"], o["Tcw"], p["Tcw"], "Tcw")


def test_pose_optimization_too_few_points(gpu):
    p = synth.synth_pose(n=2)
    r = gpu.Optimizer.PoseOptimization(p)
    assert r["n_inliers"] == 0 and np.array_equal(r["Tcw"], p["Tcw"])   # src/Optimizer.cc:453-454


@pytest.mark.parametrize("kw,its,robust", [(dict(n_free=11, n_fixed=1, n_points=800, seed=5100), 10, False),
                                           (dict(n_free=5, n_fixed=1, n_points=300, seed=5101, mono_frac=0.3), 20, True),
                                           (dict(n_free=2, n_fixed=1, n_points=150, seed=5103, outlier_frac=0.0), 20, True),
                                           # beyond the register-tile solver (30 free keyframes): the LDS / global-scratch solver
                                           (dict(n_free=40, n_fixed=1, n_points=2000, seed=5102), 10, False)])
def test_bundle_adjustment_parity(gpu, oracle, kw, its, robust):
    """Optimizer::BundleAdjustment over keyframes and map points (src/Optimizer.cc:55-323): keyframe 0 fixed, ONE
    optimize(nIterations), Huber kernels only when bRobust, nothing is erased, a point without observations stays put."""
    p = synth.synth_ba(**kw)
    orphan = np.array([[0.1, 0.2, 3.0], [-0.4, 0.1, 4.0]], np.float32)
    p["points"] = np.concatenate([p["points"], orphan])          # two map points nobody observes (removed from the graph, :193-201)
    r = gpu.Optimizer.BundleAdjustment(p, its, bRobust=robust)
    o = oracle.bundle_adjustment(p, its, robust)
    assert list(r["iters"]) == [int(o["iters"][0]), 0] and r["iters"][0] >= 3
    _check_trace(r, o)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    assert np.array_equal(r["points"][-2:], orphan) and not r["edge_outlier"].any()
    assert np.array_equal(r["poses"][0], o["poses"][0])


@pytest.mark.parametrize("kw,its,robust", [(dict(n_free=80, n_fixed=1, n_points=3000, seed=5200), 10, False),        # n = 480: no padding
                                           (dict(n_free=131, n_fixed=2, n_points=6000, seed=5201, mono_frac=0.3), 12, True),   # n = 786 -> N = 800
                                           (dict(n_free=67, n_fixed=1, n_points=900, seed=5202, outlier_frac=0.0), 8, True)])
def test_bundle_adjustment_map_scale(gpu, oracle, kw, its, robust):
    """More free keyframes than one workgroup factorises: the map-scale path (pair CSR from the host, dense lower
    triangle in HBM, panel / update LDL^T across the chip, k_bal_* in csrc/gba.hip) against the same oracle."""
    p = synth.synth_ba(**kw)
    r = gpu.Optimizer.BundleAdjustment(p, its, bRobust=robust)
    o = oracle.bundle_adjustment(p, its, robust)
    assert list(r["iters"]) == [int(o["iters"][0]), 0] and r["iters"][0] >= 3
    _check_trace(r, o)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    f = p["fixed"].astype(bool)
    assert np.array_equal(r["poses"][f], o["poses"][f])


@pytest.mark.parametrize("kw", [dict(n_free=11, n_fixed=1, n_points=800, seed=5100), dict(n_free=3, n_fixed=1, n_points=150, seed=5203),
                                dict(n_free=34, n_fixed=2, n_points=1500, seed=3005)])
def test_map_scale_path_on_small_windows(gpu, oracle, kw, monkeypatch):
    """EAO_BA_SOLVER=big forces the map-scale kernels onto windows the single-workgroup solvers would take (n = 66, 18 and
    204 unknowns: identity padding up to the 32-column panels, a single panel, odd tile counts): same results as the oracle,
    for BundleAdjustment and for both passes of LocalBundleAdjustment."""
    monkeypatch.setenv("EAO_BA_SOLVER", "big")
    p = synth.synth_ba(**kw)
    r = gpu.Optimizer.BundleAdjustment(p, 10, bRobust=True)
    o = oracle.bundle_adjustment(p, 10, True)
    assert list(r["iters"]) == [int(o["iters"][0]), 0]
    _check_trace(r, o)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    r = gpu.Optimizer.LocalBundleAdjustment(p)
    o = oracle.local_ba(p)
    assert list(r["iters"]) == list(o["iters"])
    _check_trace(r, o)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    assert np.array_equal(r["edge_outlier"], o["edge_outlier"])


def test_local_ba_oversized_window(gpu, oracle):
    """A local window with more than 64 free keyframes goes through the map-scale path too (both optimize() passes, the
    outlier pass between them deactivating edges in the pair lists)."""
    p = synth.synth_ba(n_free=70, n_fixed=3, n_points=2500, seed=5204)
    r = gpu.Optimizer.LocalBundleAdjustment(p)
    o = oracle.local_ba(p)
    assert list(r["iters"]) == list(o["iters"])
    _check_trace(r, o)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    assert np.array_equal(r["edge_outlier"], o["edge_outlier"])


@pytest.mark.parametrize("kw,pk,its,robust", [(dict(n_free=8, n_fixed=1, n_points=500, seed=5400), dict(n_planes=4), 10, True),
                                              (dict(n_free=5, n_fixed=2, n_points=200, seed=5402, mono_frac=0.5), dict(n_planes=1, seed=7002, outlier_edges=0), 10, False),
                                              # beyond the register-tile solver, and on the map-scale path
                                              (dict(n_free=40, n_fixed=1, n_points=2000, seed=5403), dict(n_planes=6, seed=7003, outlier_edges=3), 10, True),
                                              (dict(n_free=90, n_fixed=1, n_points=3000, seed=5404), dict(n_planes=8, seed=7004, outlier_edges=4), 10, False)])
def test_bundle_adjustment_with_planes(gpu, oracle, kw, pk, its, robust):
    """The MapPlane vertices / EdgePlane edges of Optimizer::BundleAdjustment (src/Optimizer.cc:203-252): marginalised 3-dof
    plane landmarks (Plane3D::oplus), information diag(3282.8, 3282.8, 1e4), always a Huber kernel, g2o's central-difference
    Jacobians (delta = 1e-9, i.e. ~1e-7 of rounding noise in any implementation) on both vertices."""
    p = synth.add_ba_planes(synth.synth_ba(**kw), **pk)
    r = gpu.Optimizer.BundleAdjustment(p, its, bRobust=robust)
    o = oracle.bundle_adjustment(p, its, robust)
    assert list(r["iters"]) == [int(o["iters"][0]), 0] and r["iters"][0] >= 3
    _check_trace(r, o, rel=CHI2_REL_PLANES)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    _check_updates(r["planes"], o["planes"], p["planes"], "planes")
    assert np.abs(r["planes"] - p["planes"]).max() > 1e-3            # the planes did move
    f = p["fixed"].astype(bool)
    assert np.array_equal(r["poses"][f], o["poses"][f])
    # aborted before the first iteration: planes come back normalised, nothing else changes
    ra = gpu.Optimizer.BundleAdjustment(p, its, stop=np.array([1], np.uint8), bRobust=robust)
    assert ra["aborted"] and np.allclose(ra["planes"], p["planes"], atol=1e-6) and np.array_equal(ra["points"], p["points"])


def test_bundle_adjustment_limits(gpu):
    p = synth.synth_ba(n_free=8193, n_fixed=1, n_points=60, seed=5104)
    with pytest.raises(gpu.EaoError):            # beyond the map-scale path's 8192 free keyframes (2048 until round 6): refused, not approximated
        gpu.Optimizer.BundleAdjustment(p, 10, bRobust=False)
    p = synth.synth_ba(n_free=4, n_fixed=1, n_points=100, seed=5105)
    r = gpu.Optimizer.BundleAdjustment(p, 10, stop=np.array([1], np.uint8))
    assert r["aborted"] and np.array_equal(r["points"], p["points"])


def _same_result(a, b):
    return (np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"]) and
            np.array_equal(a["edge_outlier"], b["edge_outlier"]) and list(a["iters"]) == list(b["iters"]) and
            np.array_equal(a["chi2"], b["chi2"]))


def _close_result(a, b, p):
    """Same LM schedule and outlier table; poses / points within the parity bar of the UPDATE (the batched Schur assembly adds
    its landmarks in another order than the single-window kernel: rounding-level differences only)."""
    assert list(a["iters"]) == list(b["iters"]) and np.array_equal(a["edge_outlier"], b["edge_outlier"])
    _check_updates(a["poses"], b["poses"], p["poses"], "poses")
    _check_updates(a["points"], b["points"], p["points"], "points")
    assert np.allclose(a["chi2"], b["chi2"], rtol=CHI2_REL_BATCH_VS_SINGLE)


def test_local_ba_batch_equals_single_calls(gpu):
    """eao_local_ba_batch: the window is the z dimension of every launch.  Every window must come out as its own
    eao_local_ba call does -- clean windows, windows of different sizes in one batch, windows whose LM rejects trials (they
    freeze inside the batch and the host finishes them), a window beyond the tile solver (runs on its own inside the call)
    -- and the batch itself must be deterministic and independent of its composition and order.  A batch of ONE window
    runs the single-window kernels: bit-identical."""
    probs = [synth.synth_ba(seed=6000 + w) for w in range(5)]
    probs += [synth.synth_ba(n_free=5, n_fixed=2, n_points=300, seed=6100), synth.synth_ba(n_free=7, n_fixed=2, n_points=400, seed=6101),
              synth.synth_ba(n_free=30, n_fixed=3, n_points=1500, seed=6102)]
    probs += [synth.synth_ba(n_free=5, n_fixed=2, n_points=200, seed=sd, rot_noise_deg=25, trans_noise=0.8, point_noise=1.0, mono_frac=0.7)
              for sd in (3030, 3034, 3046)]
    probs += [synth.synth_ba(n_free=34, n_fixed=2, n_points=1500, seed=6103)]
    single = [gpu.Optimizer.LocalBundleAdjustment(p) for p in probs]
    batch = gpu.Optimizer.LocalBundleAdjustmentBatch(probs)
    assert len(batch) == len(probs)
    for w, (a, b) in enumerate(zip(single, batch)):
        _close_result(b, a, probs[w])
    assert _same_result(single[-1], batch[-1])          # (the 34-keyframe window ran on its own inside the call)
    again = gpu.Optimizer.LocalBundleAdjustmentBatch(probs)
    assert all(_same_result(a, b) for a, b in zip(batch, again)), "eao_local_ba_batch is not deterministic"
    # order and batch composition do not matter
    rev = gpu.Optimizer.LocalBundleAdjustmentBatch(probs[::-1])[::-1]
    assert all(_same_result(a, b) for a, b in zip(batch, rev))
    sub = gpu.Optimizer.LocalBundleAdjustmentBatch(probs[2:7])
    assert all(_same_result(a, b) for a, b in zip(batch[2:7], sub))
    one = gpu.Optimizer.LocalBundleAdjustmentBatch(probs[:1])
    assert _same_result(one[0], single[0])


@pytest.mark.parametrize("n", [2, 7, 8, 9, 12, 17, 21])
def test_local_ba_batch_placement_does_not_change_results(gpu, n, monkeypatch):
    """Round 4: a batch's windows are pinned to XCDs row by row, the windows of an incomplete row of eight are dealt over all eight XCDs (BA_WIN), and the
    window groups hold whole rows (8 + 8 + 9 for 25 windows).  None of it may change a bit: every batch size around the row boundaries against the even
    split of rounds 2-3 (EAO_BA_BATCH_EVEN=1), one group, and a window's own single call (within the LM bound: a single window runs other kernels)."""
    probs = [synth.synth_ba(n_free=4 + (w % 3), n_fixed=2, n_points=150 + 10 * (w % 5), seed=6400 + w) for w in range(n)]
    rows = gpu.Optimizer.LocalBundleAdjustmentBatch(probs)
    monkeypatch.setenv("EAO_BA_BATCH_EVEN", "1")
    even = gpu.Optimizer.LocalBundleAdjustmentBatch(probs)
    monkeypatch.delenv("EAO_BA_BATCH_EVEN")
    assert all(_same_result(a, b) for a, b in zip(rows, even)), "row groups and the even split differ"
    for w in (0, n - 1):
        _close_result(rows[w], gpu.Optimizer.LocalBundleAdjustment(probs[w]), probs[w])


def test_local_ba_batch_parity_configs4(gpu, oracle):
    """The 25 windows of BASELINE configs[4] (seeds 6000 + w) in one call; windows 0 and 24 against the fp64 oracle."""
    probs = [synth.synth_ba(seed=6000 + w) for w in range(25)]
    res = gpu.Optimizer.LocalBundleAdjustmentBatch(probs)
    for w in (0, 24):
        o = oracle.local_ba(probs[w])
        assert list(res[w]["iters"]) == list(o["iters"])
        _check_updates(res[w]["poses"], o["poses"], probs[w]["poses"], "poses of window %d" % w)
        _check_updates(res[w]["points"], o["points"], probs[w]["points"], "points of window %d" % w)
        assert np.array_equal(res[w]["edge_outlier"], o["edge_outlier"])


def test_local_ba_batch_stop_and_empty(gpu):
    probs = [synth.synth_ba(n_free=5, n_fixed=2, n_points=300, seed=6200 + w) for w in range(3)]
    res = gpu.Optimizer.LocalBundleAdjustmentBatch(probs, stop=np.array([1], np.uint8))
    for p, r in zip(probs, res):
        assert r["aborted"] and np.array_equal(r["points"], p["points"]) and not r["edge_outlier"].any()
    assert gpu.Optimizer.LocalBundleAdjustmentBatch([]) == []


def test_local_ba_batch_error_in_one_window_leaves_the_library_usable(gpu):
    """A window that fails validation (two edges on one camera / point pair) fails the whole call loudly -- its group stops,
    the other groups' streams are drained -- and the next call on the same thread is unaffected."""
    probs = [synth.synth_ba(n_free=5, n_fixed=2, n_points=300, seed=6300 + w) for w in range(9)]
    good = gpu.Optimizer.LocalBundleAdjustmentBatch(probs)
    bad = dict(probs[5])
    k = int(np.flatnonzero(bad["fixed"][bad["edge_cam"]] == 0)[-1])       # an edge of a free camera
    for key in ("edge_cam", "edge_point", "obs", "inv_sigma2"):
        bad[key] = np.concatenate([bad[key], bad[key][k:k + 1]])
    with pytest.raises(gpu.EaoError):
        gpu.Optimizer.LocalBundleAdjustmentBatch(probs[:5] + [bad] + probs[6:])
    again = gpu.Optimizer.LocalBundleAdjustmentBatch(probs)
    for a, b in zip(good, again):
        assert np.array_equal(a["poses"].view(np.uint32), b["poses"].view(np.uint32)) and np.array_equal(a["points"].view(np.uint32), b["points"].view(np.uint32))
        assert list(a["iters"]) == list(b["iters"])


def test_map_scale_ba_with_an_abort_flag_that_is_never_raised(gpu):
    """Map-scale runs poll *stop between LM iterations (one iteration per enqueue, like g2o's forceStopFlag) instead of
    submitting the whole optimize() speculatively: an un-raised flag must not change a bit."""
    p = synth.synth_ba(n_free=40, n_fixed=1, n_points=2500, seed=5400)
    a = gpu.Optimizer.BundleAdjustment(p, 8, bRobust=True)
    b = gpu.Optimizer.BundleAdjustment(p, 8, stop=np.zeros(1, np.uint8), bRobust=True)
    assert list(a["iters"]) == list(b["iters"]) and np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"])
    la = gpu.Optimizer.LocalBundleAdjustment(p)
    lb = gpu.Optimizer.LocalBundleAdjustment(p, stop=np.zeros(1, np.uint8))
    assert list(la["iters"]) == list(lb["iters"]) and np.array_equal(la["poses"], lb["poses"]) and np.array_equal(la["edge_outlier"], lb["edge_outlier"])


def _two_maps(kw1, kw2):
    """Two maps that share nothing, as ONE problem (cameras and points of the second renumbered behind the first): the reduced camera system falls into two
    diagonal blocks -- whole tile rows of the map-scale path's structure are dead in the tile columns of the other map."""
    a, b = synth.synth_ba(**kw1), synth.synth_ba(**kw2)
    nc, npt = len(a["poses"]), len(a["points"])
    out = dict(a)
    for k in ("poses", "fixed", "points", "poses_gt", "points_gt"):
        out[k] = np.concatenate([a[k], b[k]])
    out["edge_cam"] = np.concatenate([a["edge_cam"], b["edge_cam"] + nc]).astype(np.int32)
    out["edge_point"] = np.concatenate([a["edge_point"], b["edge_point"] + npt]).astype(np.int32)
    out["obs"] = np.concatenate([a["obs"], b["obs"]]); out["inv_sigma2"] = np.concatenate([a["inv_sigma2"], b["inv_sigma2"]])
    return out


@pytest.mark.parametrize("name", ["band 7 of 60", "band 3 of 110", "band 11 of 150, robust", "two maps", "two maps, the second inside a tile"])
def test_bundle_adjustment_on_sparse_maps(gpu, oracle, name):
    """Round 5: the map-scale path stores and factors the reduced camera system as 64 x 64 TILES of its non-zero structure (covisibility + the fill-in of the
    elimination, worked out on the host) -- a band for a trajectory whose keyframes see their neighbours only, two blocks for two maps that share nothing.  Same LM
    schedule and updates as the oracle (which factors the dense matrix) on all of them."""
    robust = "robust" in name
    if name.startswith("band"):
        band, n = int(name.split()[1]), int(name.split()[3].rstrip(","))
        p = synth.synth_ba(n_free=n, n_fixed=1, n_points=40 * n, seed=5500 + n, band=band)
    elif name == "two maps":
        p = _two_maps(dict(n_free=40, n_fixed=1, n_points=1500, seed=5601, band=5), dict(n_free=50, n_fixed=2, n_points=1800, seed=5602, band=6))
    else:      # the second map starts in the middle of a tile row (36 free cameras = 216 rows = 3.4 tiles): its first tile row is shared with the first map's last cameras
        p = _two_maps(dict(n_free=36, n_fixed=1, n_points=1400, seed=5603, band=4), dict(n_free=33, n_fixed=1, n_points=1300, seed=5604))
    r = gpu.Optimizer.BundleAdjustment(p, 8, bRobust=robust)
    o = oracle.bundle_adjustment(p, 8, robust)
    assert list(r["iters"]) == [int(o["iters"][0]), 0] and r["iters"][0] >= 3
    _check_trace(r, o, rel=CHI2_REL_PLANES)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")


@pytest.mark.parametrize("name,segments", [("band 3 of 110", 1), ("band 3 of 110", 4), ("band 3 of 110", 9), ("band 7 of 60", 3), ("band 11 of 150, robust", 5), ("band 11 of 150, robust", 0),
                                           ("two maps", 4), ("two maps, the second inside a tile", 6), ("dense 80", 3), ("ring 5 of 120", 0), ("ring 5 of 120", 7)])
def test_bundle_adjustment_in_nested_dissection_order(gpu, oracle, name, segments, monkeypatch):
    """Round 6: the map-scale path eliminates the keyframes in a nested-dissection order (segments of the trajectory first, the separator keyframes between them last;
    csrc/gba.hip gba_build_plan) and runs the panels of independent segments in ONE launch per level of the schedule.  EAO_BA_ND forces the number of segments onto maps
    of test size (0: the library's choice, 1: natural order): same LM schedule and updates as the oracle -- which factors the dense matrix in natural order -- whatever the
    order, on bands, on a ring (the last keyframes see the first: they all join the separators), on two maps that share nothing and on a map where everybody sees everybody."""
    robust = "robust" in name
    if name.startswith("band"):
        band, n = int(name.split()[1]), int(name.split()[3].rstrip(","))
        p = synth.synth_ba(n_free=n, n_fixed=1, n_points=40 * n, seed=5500 + n, band=band)
    elif name.startswith("ring"):      # (a band whose points also wrap around: point i of the last keyframes is seen by the first ones)
        band, n = int(name.split()[1]), int(name.split()[3])
        p = synth.synth_ba(n_free=n, n_fixed=1, n_points=30 * n, seed=5700 + n, band=band)
        ec = p["edge_cam"].copy()
        far = ec >= n - 3
        ec[far] = 1 + (ec[far] - (n - 3)) * 2          # the last three cameras' observations go to cameras 1, 3, 5: the trajectory closes on itself
        # (keep at most one edge per (camera, point))
        key = ec.astype(np.int64) * (len(p["points"]) + 1) + p["edge_point"]
        _, first = np.unique(key, return_index=True)
        keep = np.zeros(len(ec), bool); keep[first] = True
        for k in ("edge_point", "obs", "inv_sigma2"):
            p[k] = p[k][keep]
        p["edge_cam"] = ec[keep].astype(np.int32)
    elif name == "two maps":
        p = _two_maps(dict(n_free=40, n_fixed=1, n_points=1500, seed=5601, band=5), dict(n_free=50, n_fixed=2, n_points=1800, seed=5602, band=6))
    elif name == "dense 80":
        p = synth.synth_ba(n_free=80, n_fixed=1, n_points=3000, seed=5200)
    else:
        p = _two_maps(dict(n_free=36, n_fixed=1, n_points=1400, seed=5603, band=4), dict(n_free=33, n_fixed=1, n_points=1300, seed=5604))
    monkeypatch.setenv("EAO_BA_ND", str(segments))
    r = gpu.Optimizer.BundleAdjustment(p, 8, bRobust=robust)
    o = oracle.bundle_adjustment(p, 8, robust)
    assert list(r["iters"]) == [int(o["iters"][0]), 0] and r["iters"][0] >= 3
    _check_trace(r, o, rel=CHI2_REL_PLANES)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")
    r2 = gpu.Optimizer.BundleAdjustment(p, 8, bRobust=robust)          # (the cached plan: bit for bit the same call)
    assert np.array_equal(r["poses"], r2["poses"]) and np.array_equal(r["points"], r2["points"])


@pytest.mark.parametrize("kw", [dict(n_free=80, n_fixed=1, n_points=3000, seed=5200), dict(n_free=60, n_fixed=1, n_points=2400, seed=5560, band=7)])
def test_map_scale_assembly_with_four_wave_pairs(gpu, oracle, kw, monkeypatch):
    """The Schur assembly of the map-scale path runs one wavefront per camera pair and FOUR for pairs with more than 2048 common landmarks (a keyframe's diagonal
    pair on a dense map); EAO_BA_PAIR_LONG lowers the limit so that maps of test size take the four-wave kernel for most of their pairs and the one-wave kernel
    for the rest -- both launches, the XCD deal of each class with its idle slots -- against the oracle."""
    monkeypatch.setenv("EAO_BA_PAIR_LONG", "48")
    p = synth.synth_ba(**kw)
    r = gpu.Optimizer.BundleAdjustment(p, 8, bRobust=False)
    o = oracle.bundle_adjustment(p, 8, False)
    assert list(r["iters"]) == [int(o["iters"][0]), 0] and r["iters"][0] >= 3
    _check_trace(r, o, rel=CHI2_REL_PLANES)
    _check_updates(r["poses"], o["poses"], p["poses"], "poses")
    _check_updates(r["points"], o["points"], p["points"], "points")


@pytest.mark.parametrize("kw", [dict(n_free=110, n_fixed=1, n_points=4400, seed=5610, band=3), dict(n_free=70, n_fixed=2, n_points=2500, seed=5611)])
def test_map_scale_set_up_on_the_host_crew(gpu, oracle, kw, monkeypatch):
    """Round 5: the covisibility structure of the map-scale path (pairs of every camera, their landmark lists) is counted and filled camera by camera, on the
    process-wide host crew for large maps.  EAO_BA_SETUP_THREADS forces the crew onto a map of test size: the device arrays -- hence the result -- must not depend on
    how many workers built them, and a map-scale window INSIDE a batch call (whose set-up already runs on a crew thread) builds its structure on that thread alone."""
    p = synth.synth_ba(**kw)
    monkeypatch.setenv("EAO_BA_SETUP_THREADS", "1")
    one = gpu.Optimizer.BundleAdjustment(p, 6, bRobust=False)
    for nt in ("2", "5", "12"):
        monkeypatch.setenv("EAO_BA_SETUP_THREADS", nt)
        r = gpu.Optimizer.BundleAdjustment(p, 6, bRobust=False)
        assert np.array_equal(r["poses"], one["poses"]) and np.array_equal(r["points"], one["points"]) and list(r["iters"]) == list(one["iters"]), nt
    o = oracle.bundle_adjustment(p, 6, False)
    assert list(one["iters"]) == [int(o["iters"][0]), 0]
    _check_updates(one["poses"], o["poses"], p["poses"], "poses")
    _check_updates(one["points"], o["points"], p["points"], "points")
    # the same map as a LocalBundleAdjustment window beside small windows in one batch call: equal to its own single call
    probs = [synth.synth_ba(seed=6200), p, synth.synth_ba(seed=6201)]
    single = gpu.Optimizer.LocalBundleAdjustment(p)
    batch = gpu.Optimizer.LocalBundleAdjustmentBatch(probs)
    assert _same_result(single, batch[1])


@pytest.mark.parametrize("threads", ["1", "5"])
def test_map_scale_set_up_refuses_bad_edge_lists_the_same_way_on_any_crew(gpu, threads, monkeypatch):
    """Round 6: the validation pass of a map-scale call runs in chunks on a crew session.  What it refuses -- an index out of range, two edges between one camera and one
    point -- and the words it refuses them with must not depend on the crew; an edge list that is NOT grouped by landmark falls back to the serial walks and gives the
    result of its ordered twin (edge order only permutes the sums inside one landmark's and one camera's lists: same bits is not promised, the same LM trace is)."""
    monkeypatch.setenv("EAO_BA_SETUP_THREADS", threads)
    p = synth.synth_ba(n_free=45, n_fixed=2, n_points=1800, seed=5750)
    bad = dict(p); ec = p["edge_cam"].copy(); ec[1234] = len(p["poses"]); bad["edge_cam"] = ec
    with pytest.raises(Exception, match="edge 1234 out of range"):
        gpu.Optimizer.BundleAdjustment(bad, 3, bRobust=False)
    bad = dict(p); ep = p["edge_point"].copy(); ep[0] = -1; bad["edge_point"] = ep
    with pytest.raises(Exception, match="edge 0 out of range"):
        gpu.Optimizer.BundleAdjustment(bad, 3, bRobust=False)
    e = 4321
    dup = dict(p)
    for k in ("edge_cam", "edge_point", "obs", "inv_sigma2"):
        dup[k] = np.ascontiguousarray(np.insert(p[k], e + 1, p[k][e], axis=0))
    with pytest.raises(Exception, match="two edges join camera %d and point %d" % (p["edge_cam"][e], p["edge_point"][e])):
        gpu.Optimizer.BundleAdjustment(dup, 3, bRobust=False)
    q, _ = _shuffle_edges(p, 5750)
    a, b = gpu.Optimizer.BundleAdjustment(p, 4, bRobust=False), gpu.Optimizer.BundleAdjustment(q, 4, bRobust=False)
    assert list(a["iters"]) == list(b["iters"]) and list(a["trace"]["trials"]) == list(b["trace"]["trials"])
    _check_updates(b["poses"], a["poses"], p["poses"], "poses")
    _check_updates(b["points"], a["points"], p["points"], "points")


def test_bundle_adjustment_beyond_2048_keyframes(gpu, monkeypatch):
    """Round 6: the map-scale path takes up to 8192 free keyframes (2048 until then -- a limit nothing in the kernels needed).  No dense oracle at that size: a 2304-keyframe band
    in the nested-dissection order the library chooses against the SAME map in natural keyframe order (the chain of rounds 3-5: another elimination order, another
    rounding, the same system -- both orders are held to the oracle on smaller maps above), the LM schedule and the update bar; and the limit itself, loudly."""
    p = synth.synth_ba(n_free=2304, n_fixed=1, n_points=36000, seed=5760, band=9)
    a = gpu.Optimizer.BundleAdjustment(p, 4, bRobust=False)
    monkeypatch.setenv("EAO_BA_ND", "1")
    b = gpu.Optimizer.BundleAdjustment(p, 4, bRobust=False)
    monkeypatch.delenv("EAO_BA_ND")
    assert list(a["iters"]) == list(b["iters"]) == [4, 0] and list(a["trace"]["trials"]) == list(b["trace"]["trials"])
    assert np.all(np.isfinite(a["poses"])) and np.all(np.isfinite(a["points"]))
    assert a["trace"]["chi2"][-1] < a["trace"]["chi2"][0]
    _check_updates(a["poses"], b["poses"], p["poses"], "poses")
    _check_updates(a["points"], b["points"], p["points"], "points")
    big = dict(p)
    big["poses"] = np.ascontiguousarray(np.tile(p["poses"][:1], (8200, 1, 1))); big["fixed"] = np.zeros(8200, np.uint8)
    with pytest.raises(Exception, match="at most 8192 free keyframes"):
        gpu.Optimizer.BundleAdjustment(big, 1, bRobust=False)


def test_two_map_scale_windows_in_one_batch(tmp_path):
    """Round 5 regression: the host-side panel tables of a map-scale window (which panel launches which work records) were thread-local to the set-up worker; a second
    map-scale window prepared by the SAME worker overwrote them before the first window's launches were enqueued, and the first window silently ran with the second
    one's panels (wrong result, wrong iteration counts).  The worker count is read once per process, so the check runs in a process of its own with ONE set-up worker."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EAO_BA_BATCH_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "dbg_batch_two_maps.py")], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln[:2] in ("A ", "S ", "B ")]
    assert len(lines) == 3 and all(ln.endswith("identical") for ln in lines), out.stdout


def _shuffle_edges(p, seed):
    """the same problem with its edge list in random order (edge_cam / edge_point / obs / inv_sigma2 permuted together)"""
    q = dict(p)
    perm = np.random.default_rng(seed).permutation(len(p["edge_cam"]))
    for k in ("edge_cam", "edge_point", "obs", "inv_sigma2"):
        q[k] = np.ascontiguousarray(p[k][perm])
    return q, perm


@pytest.mark.parametrize("kw", [dict(n_free=6, n_fixed=2, n_points=400, seed=5700), dict(n_free=20, n_fixed=4, n_points=3000, seed=5701),
                                dict(n_free=60, n_fixed=1, n_points=2400, seed=5702, band=7), dict(n_free=45, n_fixed=2, n_points=1800, seed=5703)])
def test_edge_lists_in_any_order(gpu, oracle, kw):
    """Round 5: the host set-up takes short cuts when the edge list comes landmark by landmark (what every generator and both adapters produce: the landmarks' edge
    lists are then the edge list itself, the observer lists a filtered copy).  A caller may hand the edges over in ANY order: the general paths, against the oracle on the
    same shuffled problem -- and, through the permutation, the same outlier table as the ordered problem gives."""
    p = synth.synth_ba(**kw)
    q, perm = _shuffle_edges(p, kw["seed"])
    if kw["n_free"] > 30:
        r, o = gpu.Optimizer.BundleAdjustment(q, 6, bRobust=False), oracle.bundle_adjustment(q, 6, False)
        assert list(r["iters"]) == [int(o["iters"][0]), 0]
    else:
        r, o = gpu.Optimizer.LocalBundleAdjustment(q), oracle.local_ba(q)
        assert list(r["iters"]) == list(o["iters"]) and np.array_equal(r["edge_outlier"], o["edge_outlier"])
        ordered = gpu.Optimizer.LocalBundleAdjustment(p)
        assert np.array_equal(r["edge_outlier"], ordered["edge_outlier"][perm])
    _check_updates(r["poses"], o["poses"], q["poses"], "poses")
    _check_updates(r["points"], o["points"], q["points"], "points")
