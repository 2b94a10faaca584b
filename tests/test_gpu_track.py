"""Row f1, second half: the device-resident tracked frame.  eao_tracker_track_local_map chains ComputeStereoFromRGBD +
AssignFeaturesToGrid -> isInFrustum over the local map -> SearchByProjection(points) -> PoseOptimization on the device behind
the extractor's outputs; it must give what the CPU oracle gives for the same chain (integer tables bit for bit, the pose within the LM bound), and, bit for bit,
what the host-hop calls of the same C-ABI give on the same data."""
import numpy as np
import pytest
import torch

from eao_fusion_amd import synth

pytestmark = pytest.mark.gpu
KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def _scene(seed, n=900, prior_frac=0.0, mono_frac=0.25):
    """A frame (keypoints as the extractor would leave them + a depth image) and a local map seen from a pose prior."""
    rng = np.random.default_rng(seed)
    cur, last, _ = synth.synth_tracking(n=n, seed=seed, mono_frac=0.0, occupied_frac=0.0)
    N = len(cur["kp_x"])
    ok = (cur["kp_x"] >= 1) & (cur["kp_x"] < 638) & (cur["kp_y"] >= 1) & (cur["kp_y"] < 478)
    kx, ky = np.where(ok, cur["kp_x"], 5.5).astype(np.float32), np.where(ok, cur["kp_y"], 7.25).astype(np.float32)
    kps = np.zeros(N, KP)
    kps["x"], kps["y"], kps["angle"], kps["octave"], kps["size"], kps["class_id"] = kx, ky, cur["kp_angle"], cur["kp_octave"], 31, -1
    # depth image: the depth of the map point each keypoint observes where known, noise elsewhere; a fraction of pixels invalid (0)
    depth = rng.uniform(1.5, 6.0, (480, 640)).astype(np.float32)
    Xw = last["Xw"].astype(np.float64)
    T = cur["Tcw"].astype(np.float64)
    zc = (Xw @ T[:3, :3].T + T[:3, 3])[:, 2]
    # (synth_tracking permutes the keypoints: recover which keypoint observes which point through the descriptors' order is not
    #  needed -- any valid depth works for the chain; give every keypoint pixel a plausible depth)
    depth[ky.astype(int), kx.astype(int)] = rng.uniform(1.8, 6.2, N).astype(np.float32)
    inval = rng.random(N) < mono_frac
    depth[ky[inval].astype(int), kx[inval].astype(int)] = 0.0
    M = len(Xw)
    dist = np.linalg.norm(Xw, axis=1).astype(np.float32)
    normal = (-Xw / np.linalg.norm(Xw, axis=1, keepdims=True)).astype(np.float32)     # facing the camera at the origin
    pts = dict(active=(rng.random(M) < 0.95).astype(np.uint8), Xw=last["Xw"], normal=-normal, min_dist_inv=(0.6 * dist).astype(np.float32),
               max_dist_inv=(1.7 * dist).astype(np.float32), max_dist=(dist * np.float32(1.2) ** (last["octave"] - 0.5)).astype(np.float32),
               descriptors=last["descriptors"])
    prior = None
    if prior_frac > 0:
        prior = np.full(N, -1, np.int32)
        ks = rng.choice(N, int(prior_frac * N), replace=False)
        prior[ks] = rng.choice(M, len(ks), replace=False)
    return cur, kps, np.ascontiguousarray(cur["descriptors"]), depth, pts, prior


class _ProductCalls:
    """The host-hop calls of the product's C-ABI (each of them parity-tested against the oracle in its own test file)."""
    def __init__(self, E):
        from eao_fusion_amd import frame as FR
        self.stereo, self.frustum = FR.compute_stereo_from_rgbd, FR.is_in_frustum
        self.undistort, self.bounds = FR.undistort_keypoints, FR.compute_image_bounds
        self.search = lambda frame, mps, th, nnratio: E.ORBmatcher(nnratio, True).SearchByProjectionPoints(frame, mps, th)
        self.pose = E.Optimizer.PoseOptimization


class _OracleCalls:
    """The same four steps on the CPU oracle: oracle/frame_cpu.cpp (src/Frame.cc:638-695, 1016-1037), oracle/match_cpu.cpp
    (src/ORBmatcher.cc:45-129), oracle/lm_cpu.cpp (src/Optimizer.cc:325-673) -- Tracking::TrackLocalMap's data path
    (src/Tracking.cc:2233-2297, 2587-2641) chained on the host."""
    def __init__(self, O):
        fb = O.frame_binding()
        self.stereo, self.frustum = fb.compute_stereo_from_rgbd, fb.is_in_frustum
        self.undistort, self.bounds = fb.undistort_keypoints, fb.compute_image_bounds
        self.search = O.search_by_projection_points
        self.pose = O.pose_optimization


def _chain(calls, cur, kps, desc, depth, pts, prior, th, nnratio, prior_Xw=None, dist=None):
    """[Frame::UndistortKeyPoints + ComputeImageBounds ->] Frame::ComputeStereoFromRGBD -> Tracking::SearchLocalPoints (prior matches on bad points dropped,
    isInFrustum over the rest, SearchByProjection) -> Optimizer::PoseOptimization over every mvpMapPoints entry, step by step."""
    N, M = len(kps), len(pts["Xw"])
    kxd, kyd = np.ascontiguousarray(kps["x"]), np.ascontiguousarray(kps["y"])          # mvKeys
    bnd = np.array([0.0, 640.0, 0.0, 480.0], np.float32)
    kx, ky = kxd, kyd                                                                    # mvKeysUn
    if dist is not None:
        kx, ky = calls.undistort(kxd, kyd, cur["fx"], cur["fy"], cur["cx"], cur["cy"], dist)
        bnd = calls.bounds(640, 480, cur["fx"], cur["fy"], cur["cx"], cur["cy"], dist)
    ur, dz = calls.stereo(kxd, kyd, kx, depth, cur["mbf"])
    T = np.ascontiguousarray(cur["Tcw"], np.float32)
    Ow = (-(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))).astype(np.float32)
    logsf = float(np.log(np.float32(1.2)))
    fr = dict(Tcw=T, Ow=Ow, fx=cur["fx"], fy=cur["fy"], cx=cur["cx"], cy=cur["cy"], mbf=cur["mbf"], min_x=bnd[0], max_x=bnd[1], min_y=bnd[2], max_y=bnd[3],
              log_scale_factor=np.float32(logsf))
    fo = calls.frustum(fr, pts, 0.5)
    active = pts["active"].astype(bool)
    occupied = np.zeros(N, np.uint8)
    skip = (~fo["in_view"].astype(bool)) | (~active)
    kp_mp = np.full(N, -1, np.int32)
    if prior is not None:
        pr = np.array(prior, np.int32)
        pr[(pr >= 0) & ~active[np.maximum(pr, 0)]] = -1        # "if(pMP->isBad()) *vit = NULL", src/Tracking.cc:2596-2599
        occupied[pr != -1] = 1
        skip[pr[pr >= 0]] = True                                # mnLastFrameSeen == mCurrentFrame.mnId, :2615-2616
        kp_mp[:] = pr
    frame = dict(kp_x=kx, kp_y=ky, kp_octave=np.ascontiguousarray(kps["octave"]), kp_angle=np.ascontiguousarray(kps["angle"]), u_right=ur,
                 descriptors=desc, occupied=occupied, min_x=np.float32(bnd[0]), min_y=np.float32(bnd[2]), max_x=np.float32(bnd[1]), max_y=np.float32(bnd[3]),
                 scale_factors=cur["scale_factors"])
    lvl = np.where(skip, 0, fo["pred_level"]).astype(np.int32)
    mps = dict(proj_x=fo["proj_x"], proj_y=fo["proj_y"], proj_xr=fo["proj_xr"], view_cos=fo["view_cos"], level=lvl, descriptors=pts["descriptors"],
               skip=skip.astype(np.uint8))
    nm, match = calls.search(frame, mps, th, nnratio)
    for m in range(M):
        if match[m] >= 0:
            kp_mp[match[m]] = m
    ks = np.nonzero(kp_mp != -1)[0]
    inv_sigma2 = (np.float32(1.0) / (cur["scale_factors"] * cur["scale_factors"])).astype(np.float32)
    # what the caller's visibility counters need: isInFrustum of the points upstream projects (active, not named by a prior match)
    projected = active.copy()
    if prior is not None:
        projected[kp_mp[(kp_mp >= 0) & (occupied == 1)]] = False
    res = dict(n_matches=nm, kp_map_point=kp_mp, u_right=ur, depth=dz, n_edges=len(ks), in_view=fo["in_view"].astype(bool) & active, projected=projected)
    Xe = np.zeros((len(ks), 3), np.float32)
    inmap = kp_mp[ks] >= 0
    Xe[inmap] = pts["Xw"][kp_mp[ks][inmap]]
    if (~inmap).any():
        Xe[~inmap] = np.asarray(prior_Xw, np.float32)[ks[~inmap]]
    prob = dict(Tcw=T, points=Xe, obs=np.stack([kx[ks], ky[ks], ur[ks]], 1).astype(np.float32),
                inv_sigma2=inv_sigma2[kps["octave"][ks]], fx=cur["fx"], fy=cur["fy"], cx=cur["cx"], cy=cur["cy"], bf=cur["mbf"])
    if len(ks) < 3:
        res.update(Tcw=T, n_inliers=0, kp_outlier=np.zeros(N, np.uint8))
        return res
    r = calls.pose(prob)
    outl = np.zeros(N, np.uint8)
    outl[ks] = r["outlier"]
    res.update(Tcw=r["Tcw"], n_inliers=r["n_inliers"], kp_outlier=outl)
    return res


def _host_hop(E, cur, kps, desc, depth, pts, prior, th, nnratio):
    return _chain(_ProductCalls(E), cur, kps, desc, depth, pts, prior, th, nnratio)


def _device_buffers(kps, desc, depth, cap):
    dev = torch.device("cuda")
    N = len(kps)
    d_kps = torch.zeros((cap, 28), dtype=torch.uint8, device=dev)
    d_kps[:N] = torch.from_numpy(kps.view(np.uint8).reshape(N, 28)).to(dev)
    d_desc = torch.zeros((cap, 32), dtype=torch.uint8, device=dev)
    d_desc[:N] = torch.from_numpy(desc).to(dev)
    d_n = torch.tensor([N], dtype=torch.int32, device=dev)
    d_depth = torch.from_numpy(depth).to(dev)
    torch.cuda.synchronize()
    return d_kps, d_desc, d_n, d_depth


def _tracker(cur, cap, cap_mp, bounds=(0.0, 640.0, 0.0, 480.0), dist=None):
    from eao_fusion_amd.tracker import Tracker
    sf = cur["scale_factors"]
    trk = Tracker(cur["fx"], cur["fy"], cur["cx"], cur["cy"], cur["mbf"], tuple(float(b) for b in bounds), sf, (np.float32(1.0) / (sf * sf)).astype(np.float32),
                  float(np.log(np.float32(1.2))), cap, cap_mp)
    if dist is not None:
        trk.set_distortion(dist)
    return trk


def _pose_close(got, want, old):
    """tests/test_gpu_lm.py's bound: 1e-4 of the update (+ two float32 ulps of the value: outputs are float32)."""
    upd = max(np.abs(want.astype(np.float64) - old.astype(np.float64)).max(), 1e-6)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
    return err <= 1e-4 * upd + 2 * np.spacing(np.abs(want).max().astype(np.float32)), err, upd


@pytest.mark.parametrize("case", [dict(seed=7100), dict(seed=7101, prior_frac=0.3), dict(seed=7102, th=3.0, mono_frac=0.6),
                                  dict(seed=7103, n=300, prior_frac=0.1, nnratio=0.9), dict(seed=7104, n=1500, th=5.0),
                                  dict(seed=7106, n=1200, prior_frac=0.2, th=3.0), dict(seed=7107, n=600, mono_frac=1.0)])
def test_chained_device_path_equals_the_oracle_chain(case, oracle):
    """the device chain against the ORACLE (not against other GPU calls): every integer table of the chain
    -- mvuRight / mvDepth, the in-view flags, the greedy assignment, mvpMapPoints, the edge count, mvbOutlier, the inlier count --
    bit for bit, the optimised pose within the LM bound (1e-4 of the update)."""
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    kw = dict(case)
    th, nnratio = kw.pop("th", 1.0), kw.pop("nnratio", 0.8)
    cur, kps, desc, depth, pts, prior = _scene(**kw)
    want = _chain(_OracleCalls(oracle), cur, kps, desc, depth, pts, prior, th, nnratio)
    N, cap = len(kps), 2048
    trk = _tracker(cur, cap, 2048)
    trk.set_local_map(pts)
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, th, nnratio,
                              torch.cuda.current_stream().cuda_stream)
    assert got["n_keypoints"] == N
    assert np.array_equal(got["u_right"], want["u_right"]) and np.array_equal(got["depth"], want["depth"])
    pj = want["projected"]
    assert np.array_equal(got["map_in_view"].astype(bool)[pj], want["in_view"][pj]) and want["in_view"][pj].sum() > 50
    assert got["n_matches"] == want["n_matches"] and want["n_matches"] > 50
    assert np.array_equal(got["kp_map_point"], want["kp_map_point"])
    assert got["n_edges"] == want["n_edges"]
    assert got["n_inliers"] == want["n_inliers"]
    assert np.array_equal(got["kp_outlier"], want["kp_outlier"])
    ok, err, upd = _pose_close(got["Tcw"], want["Tcw"], cur["Tcw"])
    assert ok, "pose: |gpu - oracle| %.3e vs update %.3e" % (err, upd)


def test_prior_matches_bad_and_outside_the_local_map(oracle):
    """a prior match on an inactive (bad) point is dropped -- the keypoint is free again, as upstream's
    SearchLocalPoints sets it to NULL; a prior match that is not in the local map (-2) keeps its keypoint and is an edge built
    from the position handed over; an index beyond the uploaded map is refused before any kernel runs.  Against the oracle chain."""
    import eao_fusion_amd as E
    cur, kps, desc, depth, pts, prior = _scene(7110, n=800, prior_frac=0.2)
    rng = np.random.default_rng(5)
    N, M = len(kps), len(pts["Xw"])
    inactive = np.nonzero(pts["active"] == 0)[0]
    assert len(inactive) >= 5
    free = np.nonzero(prior < 0)[0]
    unused = np.setdiff1d(inactive, prior[prior >= 0])
    prior[free[:3]] = unused[:3]                      # three keypoints matched to points that have gone bad
    outside = free[3:23]                              # twenty keypoints matched to points the local map does not hold
    prior[outside] = -2
    prior_Xw = np.zeros((N, 3), np.float32)
    T = cur["Tcw"].astype(np.float64)
    z = rng.uniform(2.0, 5.0, len(outside))
    Xc = np.stack([(kps["x"][outside] - cur["cx"]) * z / cur["fx"], (kps["y"][outside] - cur["cy"]) * z / cur["fy"], z], 1)
    prior_Xw[outside] = ((Xc - T[:3, 3]) @ T[:3, :3]).astype(np.float32)
    want = _chain(_OracleCalls(oracle), cur, kps, desc, depth, pts, prior, 3.0, 0.8, prior_Xw)
    cap = 2048
    trk = _tracker(cur, cap, 2048)
    trk.set_local_map(pts)
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    st = torch.cuda.current_stream().cuda_stream
    got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, 3.0, 0.8, st, prior_Xw)
    assert np.array_equal(got["kp_map_point"], want["kp_map_point"])
    assert (got["kp_map_point"][outside] == -2).all() and not np.isin(got["kp_map_point"][free[:3]], unused[:3]).any()
    assert got["n_matches"] == want["n_matches"] and got["n_edges"] == want["n_edges"] and got["n_inliers"] == want["n_inliers"]
    assert np.array_equal(got["kp_outlier"], want["kp_outlier"])
    ok, err, upd = _pose_close(got["Tcw"], want["Tcw"], cur["Tcw"])
    assert ok, "pose: |gpu - oracle| %.3e vs update %.3e" % (err, upd)
    # -2 without positions, and a stale index, are refused
    with pytest.raises(Exception, match="prior_kp_Xw"):
        trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, 3.0, 0.8, st)
    stale = prior.copy()
    stale[free[30]] = M
    with pytest.raises(Exception, match="names no point of the local map"):
        trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], stale, 3.0, 0.8, st, prior_Xw)
    small = {k: (v[:M // 2] if isinstance(v, np.ndarray) else v) for k, v in pts.items()}
    trk.set_local_map(small)                          # the map shrank: the old table now points beyond it
    if (prior >= M // 2).any():
        with pytest.raises(Exception, match="names no point of the local map"):
            trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, 3.0, 0.8, st, prior_Xw)


def test_local_map_beyond_4096_points(oracle):
    """upstream's mvpLocalMapPoints has no limit (80 local keyframes give 5-10 k points).  One assignment workgroup
    now carries up to 16 points per thread: 6 000 and 12 000 local map points against the oracle chain."""
    import eao_fusion_amd as E
    for seed, big in ((7120, 6000), (7121, 12000)):
        cur, kps, desc, depth, pts, prior = _scene(seed, n=1000)
        rng = np.random.default_rng(seed)
        M = len(pts["Xw"])
        extra = big - M
        # the extra points: around the scene (some in view, most not), random descriptors
        Xe = rng.uniform([-8, -6, -2], [8, 6, 10], (extra, 3)).astype(np.float32)
        de = np.maximum(np.linalg.norm(Xe, axis=1), 0.1).astype(np.float32)
        more = dict(active=np.ones(extra, np.uint8), Xw=Xe, normal=(Xe / de[:, None]).astype(np.float32), min_dist_inv=(0.6 * de).astype(np.float32),
                    max_dist_inv=(1.7 * de).astype(np.float32), max_dist=(de * np.float32(1.2) ** 3).astype(np.float32),
                    descriptors=rng.integers(0, 256, (extra, 32), dtype=np.uint8))
        order = rng.permutation(big)                   # the real points spread over the whole index range
        allp = {k: np.ascontiguousarray(np.concatenate([pts[k], more[k]])[order]) for k in more}
        want = _chain(_OracleCalls(oracle), cur, kps, desc, depth, allp, None, 3.0, 0.8)
        cap = 2048
        trk = _tracker(cur, cap, 16384)
        trk.set_local_map(allp)
        d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
        got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], None, 3.0, 0.8,
                                  torch.cuda.current_stream().cuda_stream)
        assert got["n_matches"] == want["n_matches"] and want["n_matches"] > 300
        assert np.array_equal(got["kp_map_point"], want["kp_map_point"]) and np.array_equal(got["kp_outlier"], want["kp_outlier"])
        assert np.array_equal(got["map_in_view"].astype(bool), want["in_view"])
        ok, err, upd = _pose_close(got["Tcw"], want["Tcw"], cur["Tcw"])
        assert ok, "pose: |gpu - oracle| %.3e vs update %.3e" % (err, upd)


@pytest.mark.parametrize("case", [dict(seed=7100), dict(seed=7101, prior_frac=0.3), dict(seed=7102, th=3.0, mono_frac=0.6),
                                  dict(seed=7103, n=300, prior_frac=0.1, nnratio=0.9), dict(seed=7104, n=1500, th=5.0)])
def test_chained_device_path_equals_host_hops(case):
    import eao_fusion_amd as E
    from eao_fusion_amd.tracker import Tracker
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    kw = dict(case)
    th, nnratio = kw.pop("th", 1.0), kw.pop("nnratio", 0.8)
    cur, kps, desc, depth, pts, prior = _scene(**kw)
    want = _host_hop(E, cur, kps, desc, depth, pts, prior, th, nnratio)
    N = len(kps)
    cap = 2048
    sf = cur["scale_factors"]
    trk = Tracker(cur["fx"], cur["fy"], cur["cx"], cur["cy"], cur["mbf"], (0.0, 640.0, 0.0, 480.0), sf, (np.float32(1.0) / (sf * sf)).astype(np.float32),
                  float(np.log(np.float32(1.2))), cap, 2048)
    trk.set_local_map(pts)
    dev = torch.device("cuda")
    d_kps = torch.zeros((cap, 28), dtype=torch.uint8, device=dev)
    d_kps[:N] = torch.from_numpy(kps.view(np.uint8).reshape(N, 28)).to(dev)
    d_desc = torch.zeros((cap, 32), dtype=torch.uint8, device=dev)
    d_desc[:N] = torch.from_numpy(desc).to(dev)
    d_n = torch.tensor([N], dtype=torch.int32, device=dev)
    d_depth = torch.from_numpy(depth).to(dev)
    torch.cuda.synchronize()
    for rep in range(2):       # twice: the handle carries no state from one frame to the next
        got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, th, nnratio,
                                  torch.cuda.current_stream().cuda_stream)
        assert got["n_keypoints"] == N
        assert np.array_equal(got["u_right"], want["u_right"]) and np.array_equal(got["depth"], want["depth"])
        assert got["n_matches"] == want["n_matches"] and want["n_matches"] > 50
        assert np.array_equal(got["kp_map_point"], want["kp_map_point"])
        assert got["n_edges"] == want["n_edges"]
        assert got["n_inliers"] == want["n_inliers"]
        assert np.array_equal(got["kp_outlier"], want["kp_outlier"])
        assert np.array_equal(got["Tcw"], want["Tcw"]), "pose differs from the host-hop path"


def test_tracker_without_correspondences_keeps_the_prior():
    import eao_fusion_amd as E
    from eao_fusion_amd.tracker import Tracker
    cur, kps, desc, depth, pts, prior = _scene(7105, n=200)
    pts = dict(pts, active=np.zeros(len(pts["Xw"]), np.uint8))
    sf = cur["scale_factors"]
    trk = Tracker(cur["fx"], cur["fy"], cur["cx"], cur["cy"], cur["mbf"], (0.0, 640.0, 0.0, 480.0), sf, (np.float32(1.0) / (sf * sf)).astype(np.float32),
                  float(np.log(np.float32(1.2))), 1024, 1024)
    trk.set_local_map(pts)
    dev = torch.device("cuda")
    N = len(kps)
    d_kps = torch.zeros((1024, 28), dtype=torch.uint8, device=dev)
    d_kps[:N] = torch.from_numpy(kps.view(np.uint8).reshape(N, 28)).to(dev)
    d_desc = torch.zeros((1024, 32), dtype=torch.uint8, device=dev)
    d_n = torch.tensor([N], dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), None, 0, 640, 480, cur["Tcw"], None, 1.0, 0.8, 0)
    assert got["n_matches"] == 0 and got["n_edges"] == 0 and got["n_inliers"] == 0
    assert np.array_equal(got["Tcw"], cur["Tcw"]) and (got["u_right"] == -1).all() and not got["kp_outlier"].any()


def test_crowded_grid_cell_takes_the_sorting_network(oracle):
    """The frame set-up orders the (cell, keypoint) keys by a counting sort when no grid cell holds more than 32 keypoints, by the bitonic
    network otherwise: 70 keypoints moved into ONE cell (and 40 more onto one of its borders) must give the oracle chain's tables all the same."""
    cur, kps, desc, depth, pts, prior = _scene(7130, n=700)
    rng = np.random.default_rng(7130)
    N = len(kps)
    crowd = rng.choice(N, 110, replace=False)
    kps["x"][crowd[:70]] = (205.0 + rng.uniform(0.5, 9.0, 70)).astype(np.float32)      # cell (20 or 21, 20) of the 64 x 48 grid: 10 px cells
    kps["y"][crowd[:70]] = (203.0 + rng.uniform(0.5, 6.5, 70)).astype(np.float32)
    kps["x"][crowd[70:]] = np.float32(215.0)                                              # exactly on a cell border (round half away from zero)
    kps["y"][crowd[70:]] = (150.0 + rng.uniform(0, 100, 40)).astype(np.float32)
    depth[kps["y"].astype(int), kps["x"].astype(int)] = rng.uniform(1.8, 6.2, N).astype(np.float32)
    cur = dict(cur); cur["kp_x"], cur["kp_y"] = np.ascontiguousarray(kps["x"]), np.ascontiguousarray(kps["y"])
    want = _chain(_OracleCalls(oracle), cur, kps, desc, depth, pts, prior, 5.0, 0.8)
    cap = 2048
    trk = _tracker(cur, cap, 2048)
    trk.set_local_map(pts)
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, 5.0, 0.8,
                              torch.cuda.current_stream().cuda_stream)
    assert got["n_matches"] == want["n_matches"] and want["n_matches"] > 50
    assert np.array_equal(got["kp_map_point"], want["kp_map_point"]) and np.array_equal(got["kp_outlier"], want["kp_outlier"])
    assert np.array_equal(got["u_right"], want["u_right"]) and got["n_inliers"] == want["n_inliers"]


def test_fallback_paths_of_the_chain():
    """The chain's switchable paths (stream synchronisation instead of the polled done word, the sorting network instead of the counting sort,
    every lister a competitor in the assignment rounds) against the oracle chain: a short randomised sweep in a process of its own, because the
    switches are read once per process."""
    import os, subprocess, sys
    env = dict(os.environ, EAO_TRACK_POLL="0", EAO_TRACK_COUNTING_SORT="0", EAO_TRACK_ALL_LISTERS="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "sweep_track.py"), "17", "16"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "tracker sweep: 16 frames, 0 mismatches" in out.stdout, out.stdout[-2000:]


def _chain_motion(calls, search_frames, cur, kps, desc, depth, last, th, mono, discard=True, dist=None):
    """Tracking::TrackWithMotionModel's data path step by step (src/Tracking.cc:1717-2231): Frame::ComputeStereoFromRGBD, SearchByProjection(Cur, Last, th, bMono)
    (src/ORBmatcher.cc:1328-1472), PoseOptimization over the matches from the predicted pose, the outlier discard (:2188-2207)."""
    N = len(kps)
    kxd, kyd = np.ascontiguousarray(kps["x"]), np.ascontiguousarray(kps["y"])
    kx, ky, bnd = kxd, kyd, np.array([0.0, 640.0, 0.0, 480.0], np.float32)
    if dist is not None:      # Frame::UndistortKeyPoints / ComputeImageBounds in front of everything that reads mvKeysUn
        kx, ky = calls.undistort(kxd, kyd, cur["fx"], cur["fy"], cur["cx"], cur["cy"], dist)
        bnd = calls.bounds(640, 480, cur["fx"], cur["fy"], cur["cx"], cur["cy"], dist)
    ur, dz = calls.stereo(kxd, kyd, kx, depth, cur["mbf"])
    T = np.ascontiguousarray(cur["Tcw"], np.float32)
    frame = dict(kp_x=kx, kp_y=ky, kp_octave=np.ascontiguousarray(kps["octave"]), kp_angle=np.ascontiguousarray(kps["angle"]), u_right=ur, descriptors=desc,
                 occupied=None, min_x=np.float32(bnd[0]), min_y=np.float32(bnd[2]), max_x=np.float32(bnd[1]), max_y=np.float32(bnd[3]), scale_factors=cur["scale_factors"],
                 Tcw=T, fx=cur["fx"], fy=cur["fy"], cx=cur["cx"], cy=cur["cy"], mbf=cur["mbf"], mb=cur["mb"])
    nm, cur_match = search_frames(frame, last, th, mono)
    ks = np.nonzero(cur_match >= 0)[0]
    inv_sigma2 = (np.float32(1.0) / (cur["scale_factors"] * cur["scale_factors"])).astype(np.float32)
    res = dict(n_matches=nm, kp_map_point=cur_match.copy(), u_right=ur, depth=dz, n_edges=len(ks))
    if len(ks) < 3:
        res.update(Tcw=T, n_inliers=0, kp_outlier=np.zeros(N, np.uint8))
        return res
    prob = dict(Tcw=T, points=np.ascontiguousarray(last["Xw"][cur_match[ks]], np.float32), obs=np.stack([kx[ks], ky[ks], ur[ks]], 1).astype(np.float32),
                inv_sigma2=inv_sigma2[kps["octave"][ks]], fx=cur["fx"], fy=cur["fy"], cx=cur["cx"], cy=cur["cy"], bf=cur["mbf"])
    r = calls.pose(prob)
    outl = np.zeros(N, np.uint8)
    outl[ks] = r["outlier"]
    if discard:
        res["kp_map_point"][outl != 0] = -1
        outl[:] = 0
    res.update(Tcw=r["Tcw"], n_inliers=r["n_inliers"], kp_outlier=outl)
    return res


@pytest.mark.parametrize("case", [dict(seed=7300), dict(seed=7301, th=7.0, mono_frac=0.0), dict(seed=7302, n=300, th=15.0, mono_frac=1.0, mono=True),
                                  dict(seed=7303, n=1500, th=30.0), dict(seed=7304, n=900, moved=0.4), dict(seed=7305, n=900, moved=-0.4, discard=False),
                                  dict(seed=7306, n=60, th=15.0), dict(seed=7307, n=2000, th=7.0, check=False)])
def test_motion_model_stage_equals_the_oracle_chain(case, oracle):
    """Round 4 (VERDICT r3 missing #3): eao_tracker_track_with_motion_model -- frame set-up, SearchByProjection(Cur, Last) with the rotation histogram of this fork
    (factor HISTO_LENGTH / 360), PoseOptimization from the predicted pose, outlier discard -- against the same steps on the CPU oracle: every integer table bit
    for bit (mvuRight / mvDepth, the greedy assignment after the rotation filter as LAST-FRAME indices, the search's return value, the edge count,
    mvbOutlier, the matches left), the pose within the LM bound.  Forward / backward motion beyond the baseline switches the level windows (:1343-1347)."""
    import eao_fusion_amd as E  # noqa: F401
    kw = dict(seed=case["seed"], n=case.get("n", 900), mono_frac=case.get("mono_frac", 0.25))
    cur, kps, desc, depth, pts, _ = _scene(**kw)
    _, last, _ = synth.synth_tracking(n=kw["n"], seed=kw["seed"], mono_frac=0.0, occupied_frac=0.0)
    if "moved" in case:      # the camera moved along its axis by more than the baseline: forward / backward level windows
        cur = dict(cur); T = cur["Tcw"].copy(); T[2, 3] = -case["moved"]; cur["Tcw"] = T
    th, mono, check, discard = case.get("th", 15.0), case.get("mono", False), case.get("check", True), case.get("discard", True)
    if mono:
        depth = np.zeros_like(depth)
    oc = _OracleCalls(oracle)
    want = _chain_motion(oc, lambda f, l, t, m: oracle.search_by_projection_frames(f, l, t, m, check), cur, kps, desc, depth, last, th, mono, discard)
    cap = 2048 if len(kps) <= 2048 else 4096
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    trk = _tracker(cur, cap, 2048)
    got = trk.track_with_motion_model(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], last, th, mono, check, discard,
                                      torch.cuda.current_stream().cuda_stream)
    assert got["n_keypoints"] == len(kps)
    assert np.array_equal(got["u_right"], want["u_right"]) and np.array_equal(got["depth"], want["depth"])
    assert got["n_matches"] == want["n_matches"] and got["n_edges"] == want["n_edges"]
    assert np.array_equal(got["kp_map_point"], want["kp_map_point"]), "matches (last-frame indices)"
    assert np.array_equal(got["kp_outlier"], want["kp_outlier"]) and got["n_inliers"] == want["n_inliers"]
    ok, err, upd = _pose_close(got["Tcw"], want["Tcw"], cur["Tcw"])
    assert ok, "pose |gpu - oracle| %.3e of update %.3e" % (err, upd)
    if want["n_matches"] >= 20:
        assert (got["kp_map_point"] >= 0).sum() > 0
    # the stage leaves the uploaded local map alone: a TrackLocalMap call behind it gives what it gives without it
    if "moved" in case:
        return      # (this far along the axis the local map's predicted levels leave the pyramid: TrackLocalMap refuses, as it should)
    trk.set_local_map(pts)
    a = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], None, 1.0, 0.8, torch.cuda.current_stream().cuda_stream)
    trk.track_with_motion_model(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], last, th, mono, check, discard,
                                torch.cuda.current_stream().cuda_stream)
    b = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], None, 1.0, 0.8, torch.cuda.current_stream().cuda_stream)
    assert all(np.array_equal(a[k], b[k]) if isinstance(a[k], np.ndarray) else a[k] == b[k] for k in a)


def test_motion_model_stage_rejects_bad_input():
    cur, kps, desc, depth, pts, _ = _scene(seed=7310, n=200)
    _, last, _ = synth.synth_tracking(n=200, seed=7310, mono_frac=0.0, occupied_frac=0.0)
    import eao_fusion_amd as E
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, 1024)
    trk = _tracker(cur, 1024, 64)
    args = (d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480)
    bad = cur["Tcw"].copy(); bad[1, 2] = np.nan
    with pytest.raises(E.EaoError):
        trk.track_with_motion_model(*args, bad, last, 15.0)
    l2 = dict(last); l2["octave"] = last["octave"].copy(); l2["octave"][np.nonzero(last["valid"])[0][0]] = 9
    with pytest.raises(E.EaoError):
        trk.track_with_motion_model(*args, cur["Tcw"], l2, 15.0)
    big = {k: (np.concatenate([v] * 8) if isinstance(v, np.ndarray) and v.ndim >= 1 and len(v) == 200 else v) for k, v in last.items()}
    with pytest.raises(E.EaoError):
        trk.track_with_motion_model(*args, cur["Tcw"], big, 15.0)      # 1600 last-frame keypoints > max_keypoints 1024
    assert E.load().eao_abi_version() == 6


def _bow_case(seed, n, n_nodes, flip=0.05, clutter=0.15, mono=False):
    """A reference keyframe (K1 of synth_search_scene: map points, descriptors, angles, feature vector) and a current frame (K2 as the extractor would leave
    it + a depth image that gives every keypoint the depth of the point it observes, 0 where the scene says monocular)."""
    sc = synth.synth_search_scene(n=n, seed=seed, flip=flip, clutter=clutter, n_nodes=n_nodes)
    K1, K2 = sc["K1"], sc["K2"]
    N = len(K2["kp_x"])
    kps = np.zeros(N, KP)
    kps["x"], kps["y"] = np.clip(K2["kp_x"], 1, 638), np.clip(K2["kp_y"], 1, 478)
    kps["angle"], kps["octave"], kps["size"], kps["class_id"] = K2["kp_angle"], K2["kp_octave"], 31, -1
    rng = np.random.default_rng(seed + 5)
    depth = rng.uniform(1.5, 6.0, (480, 640)).astype(np.float32)
    P = sc["points"]
    T2 = sc["T2w"].astype(np.float64)
    mp2 = sc["mp2"]
    z = (P["Xw"][np.maximum(mp2, 0)].astype(np.float64) @ T2[:3, :3].T + T2[:3, 3])[:, 2]
    zz = np.where(mp2 >= 0, z, rng.uniform(1.8, 6.2, N)).astype(np.float32)
    zz[K2["u_right"] < 0] = 0.0
    if mono:
        depth[:] = 0.0
        zz[:] = 0.0
    depth[kps["y"].astype(int), kps["x"].astype(int)] = zz
    mp1 = sc["mp1"]
    kf = dict(valid=(mp1 >= 0).astype(np.uint8), Xw=np.ascontiguousarray(P["Xw"][np.maximum(mp1, 0)], np.float32), descriptors=K1["descriptors"],
              angle=K1["kp_angle"], fv=sc["fv1"])
    fx, fy, cx, cy = sc["K"]
    sf = K2["scale_factors"]
    T_last = sc["T2w"].astype(np.float32).copy()
    T_last[:3, 3] += np.float32([0.02, -0.01, 0.015])      # the last frame's pose: near the truth, as upstream's SetPose(mLastFrame.mTcw)
    cam = dict(fx=np.float32(fx), fy=np.float32(fy), cx=np.float32(cx), cy=np.float32(cy), mbf=np.float32(sc["bf"]), mb=np.float32(sc["bf"] / fx), scale_factors=sf, Tcw=T_last)
    return sc, cam, kps, np.ascontiguousarray(K2["descriptors"]), depth, kf


def _chain_bow(oracle, cam, kps, desc, depth, kf, fv_cur, nnratio, check, discard):
    """Tracking::TrackReferenceKeyFrame's data path step by step (src/Tracking.cc:1568-1631) on the CPU oracle: Frame::ComputeStereoFromRGBD, SearchByBoW(pKF, F)
    (src/ORBmatcher.cc:159-288), PoseOptimization over the matches from the last frame's pose, the outlier discard."""
    oc = _OracleCalls(oracle)
    N = len(kps)
    kx, ky = np.ascontiguousarray(kps["x"]), np.ascontiguousarray(kps["y"])
    ur, dz = oc.stereo(kx, ky, kx, depth, cam["mbf"])
    s2 = dict(descriptors=desc, angle=np.ascontiguousarray(kps["angle"]), valid=None, fv=fv_cur)
    nm, m12 = oracle.search_binding().search_by_bow(0, kf, s2, nnratio, check)
    cur_match = np.full(N, -1, np.int32)
    ks1 = np.nonzero(m12 >= 0)[0]
    cur_match[m12[ks1]] = ks1
    ks = np.nonzero(cur_match >= 0)[0]
    inv_sigma2 = (np.float32(1.0) / (cam["scale_factors"] * cam["scale_factors"])).astype(np.float32)
    T = np.ascontiguousarray(cam["Tcw"], np.float32)
    res = dict(n_matches=nm, kp_map_point=cur_match.copy(), u_right=ur, depth=dz, n_edges=len(ks))
    if len(ks) < 3:
        res.update(Tcw=T, n_inliers=0, kp_outlier=np.zeros(N, np.uint8))
        return res
    prob = dict(Tcw=T, points=np.ascontiguousarray(kf["Xw"][cur_match[ks]], np.float32), obs=np.stack([kx[ks], ky[ks], ur[ks]], 1).astype(np.float32),
                inv_sigma2=inv_sigma2[kps["octave"][ks]], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], bf=cam["mbf"])
    r = oc.pose(prob)
    outl = np.zeros(N, np.uint8)
    outl[ks] = r["outlier"]
    if discard:
        res["kp_map_point"][outl != 0] = -1
        outl[:] = 0
    res.update(Tcw=r["Tcw"], n_inliers=r["n_inliers"], kp_outlier=outl)
    return res


@pytest.mark.parametrize("case", [dict(seed=7400), dict(seed=7401, n=1200, n_nodes=100, flip=0.09), dict(seed=7402, n=300, n_nodes=6, clutter=0.5, ratio=0.9, check=False),
                                  dict(seed=7403, n=900, n_nodes=1), dict(seed=7404, n=700, n_nodes=40, mono=True), dict(seed=7405, n=1500, n_nodes=60, discard=False),
                                  dict(seed=7406, n=40, n_nodes=8)])
def test_reference_keyframe_stage_equals_the_oracle_chain(case, oracle):
    """Round 4 (VERDICT r3 missing #3, second half): eao_tracker_track_reference_keyframe -- frame set-up, SearchByBoW(KeyFrame, Frame) node by node with its
    rotation histogram (factor 1 / HISTO_LENGTH), PoseOptimization from the last frame's pose, outlier discard -- against the same steps on the CPU oracle: every
    integer table bit for bit, the pose within the LM bound.  One node for all keypoints (n_nodes = 1) makes every keyframe feature compete for every keypoint."""
    import eao_fusion_amd as E  # noqa: F401
    sc, cam, kps, desc, depth, kf = _bow_case(case["seed"], case.get("n", 800), case.get("n_nodes", 60), case.get("flip", 0.05), case.get("clutter", 0.15), case.get("mono", False))
    ratio, check, discard = case.get("ratio", 0.7), case.get("check", True), case.get("discard", True)
    want = _chain_bow(oracle, cam, kps, desc, depth, kf, sc["fv2"], ratio, check, discard)
    cap = 2048
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    trk = _tracker(cam, cap, 2048)
    got = trk.track_reference_keyframe(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cam["Tcw"], kf, sc["fv2"], ratio, check, discard,
                                       torch.cuda.current_stream().cuda_stream)
    assert got["n_keypoints"] == len(kps)
    assert np.array_equal(got["u_right"], want["u_right"]) and np.array_equal(got["depth"], want["depth"])
    assert got["n_matches"] == want["n_matches"] and got["n_edges"] == want["n_edges"]
    assert np.array_equal(got["kp_map_point"], want["kp_map_point"]), "matches (keyframe keypoint indices)"
    assert np.array_equal(got["kp_outlier"], want["kp_outlier"]) and got["n_inliers"] == want["n_inliers"]
    ok, err, upd = _pose_close(got["Tcw"], want["Tcw"], cam["Tcw"])
    assert ok, "pose |gpu - oracle| %.3e of update %.3e" % (err, upd)
    if case.get("n", 800) >= 300:
        assert want["n_matches"] >= 15 and got["n_inliers"] >= 10


def test_reference_keyframe_stage_rejects_bad_input():
    sc, cam, kps, desc, depth, kf = _bow_case(7410, 200, 10)
    cap = 2048
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    trk = _tracker(cam, cap, 2048)
    st = torch.cuda.current_stream().cuda_stream
    args = (d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480)

    def bad(Tl, k, fv, msg):
        with pytest.raises(RuntimeError, match=msg):
            trk.track_reference_keyframe(*args, Tl, k, fv, 0.7, True, True, st)

    T = cam["Tcw"].copy(); T[1, 2] = np.nan
    bad(T, kf, sc["fv2"], "NaN")
    fv = dict(sc["fv2"]); fv["node_id"] = fv["node_id"][::-1].copy()
    bad(cam["Tcw"], kf, fv, "ascend")
    fv = dict(sc["fv2"]); idx = fv["index"].copy(); idx[0] = 5000; fv["index"] = idx
    bad(cam["Tcw"], kf, fv, "beyond max_keypoints")
    fv = dict(sc["fv2"]); idx = fv["index"].copy(); idx[1] = idx[0]; fv["index"] = idx
    bad(cam["Tcw"], kf, fv, "twice")
    fv = dict(sc["fv2"]); idx = fv["index"].copy(); idx[idx == idx.max()] = len(kps) + 3; fv["index"] = idx      # in capacity, beyond what the extractor left
    bad(cam["Tcw"], kf, fv, "beyond the")
    k2 = dict(kf); f1 = dict(kf["fv"]); idx = f1["index"].copy(); idx[0] = len(kf["valid"]) + 1; f1["index"] = idx; k2["fv"] = f1
    bad(cam["Tcw"], k2, sc["fv2"], "keyframe feature-vector index")
    # and a good call still works afterwards
    got = trk.track_reference_keyframe(*args, cam["Tcw"], kf, sc["fv2"], 0.7, True, True, st)
    assert got["n_matches"] > 10


def test_reference_keyframe_stage_degenerate_inputs(oracle):
    """No common vocabulary node, empty feature vectors, a keyframe without keypoints: SearchByBoW returns 0, PoseOptimization has fewer than three
    correspondences and leaves the pose alone (src/Optimizer.cc:453-454) -- and the handle works normally afterwards."""
    sc, cam, kps, desc, depth, kf = _bow_case(7420, 300, 12)
    cap = 2048
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    trk = _tracker(cam, cap, 2048)
    st = torch.cuda.current_stream().cuda_stream
    args = (d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480)
    empty = dict(node_id=np.zeros(0, np.uint32), node_start=np.zeros(1, np.int32), index=np.zeros(0, np.uint32))
    far = dict(sc["fv2"]); far["node_id"] = (far["node_id"] + 100000).astype(np.uint32)
    kf0 = dict(valid=np.zeros(0, np.uint8), Xw=np.zeros((0, 3), np.float32), descriptors=np.zeros((0, 32), np.uint8), angle=np.zeros(0, np.float32), fv=empty)
    for k, fv in ((kf, far), (kf, empty), (dict(kf, fv=empty), sc["fv2"]), (kf0, sc["fv2"])):
        got = trk.track_reference_keyframe(*args, cam["Tcw"], k, fv, 0.7, True, True, st)
        assert got["n_keypoints"] == len(kps) and got["n_matches"] == 0 and got["n_edges"] == 0 and got["n_inliers"] == 0
        assert (got["kp_map_point"] == -1).all() and not got["kp_outlier"].any()
        assert np.array_equal(got["Tcw"], np.asarray(cam["Tcw"], np.float32))
    want = _chain_bow(oracle, cam, kps, desc, depth, kf, sc["fv2"], 0.7, True, True)
    got = trk.track_reference_keyframe(*args, cam["Tcw"], kf, sc["fv2"], 0.7, True, True, st)
    assert got["n_matches"] == want["n_matches"] and np.array_equal(got["kp_map_point"], want["kp_map_point"])


def test_motion_model_stage_without_last_frame_points():
    """A last frame that holds no usable map point (or no keypoint at all): the search returns 0, the pose stays the prediction."""
    cur, kps, desc, depth, pts, _ = _scene(seed=7330, n=300)
    _, last, _ = synth.synth_tracking(n=300, seed=7330, mono_frac=0.0, occupied_frac=0.0)
    cap = 2048
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    trk = _tracker(cur, cap, 2048)
    st = torch.cuda.current_stream().cuda_stream
    args = (d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480)
    none = dict(last, valid=np.zeros_like(last["valid"]))
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and v.ndim >= 1 and len(v) == len(last["valid"]) else v) for k, v in last.items()}
    for l in (none, empty):
        got = trk.track_with_motion_model(*args, cur["Tcw"], l, 15.0, False, True, True, st)
        assert got["n_keypoints"] == len(kps) and got["n_matches"] == 0 and got["n_edges"] == 0 and (got["kp_map_point"] == -1).all()
        assert np.array_equal(got["Tcw"], np.asarray(cur["Tcw"], np.float32))
    got = trk.track_with_motion_model(*args, cur["Tcw"], last, 15.0, False, True, True, st)
    assert got["n_matches"] > 50


def _plane_edges(seed, m):
    """m associated planes of a frame as synth_pose makes them (world / observed coefficients, mbSeen), the last one an outlier when m >= 3"""
    pp = synth.synth_pose(n=50, seed=seed, n_planes=m)
    return dict(plane_world=pp["plane_world"], plane_obs=pp["plane_obs"], plane_seen=pp["plane_seen"])


@pytest.mark.parametrize("stage,m", [("local_map", 4), ("local_map", 1), ("motion_model", 6), ("motion_model", 32)])
def test_plane_edges_ride_the_chained_pose_optimisation(stage, m, oracle):
    """in this fork AssociatePlanesByBoundary runs BEFORE Optimizer::PoseOptimization (src/Tracking.cc:1587, before :2181), so a frame's associated
    planes are edges of the optimisation.  eao_tracker_set_options hands them to the chain: the pose, mvbOutlier, the inlier count and mvbPlaneOutlier are those of
    the oracle chain whose PoseOptimization carries the same plane edges -- and differ from the chain without them."""
    import eao_fusion_amd as E  # noqa: F401
    cur, kps, desc, depth, pts, prior = _scene(7400 + m, n=700, prior_frac=0.1)
    planes = _plane_edges(7450 + m, m)
    oc = _OracleCalls(oracle)
    base_pose = oc.pose
    seen = {}

    def pose_with_planes(prob):
        q = dict(prob); q.update(planes)
        r = oracle.pose_optimization(q)
        seen["plane_outlier"] = r["plane_outlier"]
        return r
    cap = 2048
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    trk = _tracker(cur, cap, 2048)
    st = torch.cuda.current_stream().cuda_stream
    if stage == "local_map":
        trk.set_local_map(pts)
        run = lambda: trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, 3.0, 0.8, st)      # noqa: E731
        oc.pose = pose_with_planes
        want = _chain(oc, cur, kps, desc, depth, pts, prior, 3.0, 0.8)
    else:
        _, last, _ = synth.synth_tracking(n=700, seed=7400 + m, mono_frac=0.0, occupied_frac=0.0)
        run = lambda: trk.track_with_motion_model(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], last, 15.0, False, True, False, st)      # noqa: E731
        oc.pose = pose_with_planes
        want = _chain_motion(oc, lambda f, l, t, mo: oracle.search_by_projection_frames(f, l, t, mo, True), cur, kps, desc, depth, last, 15.0, False, False)
    plain = run()
    trk.set_options(planes=planes)
    got = run()
    assert np.array_equal(got["kp_map_point"], want["kp_map_point"]) and got["n_edges"] == want["n_edges"] and got["n_edges"] > 50
    assert got["n_inliers"] == want["n_inliers"] and np.array_equal(got["kp_outlier"], want["kp_outlier"])
    assert np.array_equal(trk.plane_outlier, seen["plane_outlier"])
    ok, err, upd = _pose_close(got["Tcw"], want["Tcw"], cur["Tcw"])
    assert ok, "pose with %d plane edges: |gpu - oracle| %.3e of update %.3e" % (m, err, upd)
    assert not np.array_equal(got["Tcw"], plain["Tcw"]), "the plane edges changed nothing"
    again = run()      # the options are one-shot: the next call is the plain chain again
    assert np.array_equal(again["Tcw"], plain["Tcw"]) and np.array_equal(again["kp_outlier"], plain["kp_outlier"])
    oc.pose = base_pose


def test_min_matches_skips_the_pose_optimisation(oracle):
    """upstream tests the search's return value BEFORE it optimises (TrackWithMotionModel: < 20 -> search again with 2 th, src/Tracking.cc:1756-1763;
    TrackReferenceKeyFrame: < 15 -> return false, :1580-1581).  With min_matches set and a search below it the chain leaves the pose alone and reports the search's
    tables only; at or above it nothing changes."""
    import eao_fusion_amd as E  # noqa: F401
    cur, kps, desc, depth, pts, _ = _scene(7500, n=60)
    _, last, _ = synth.synth_tracking(n=60, seed=7500, mono_frac=0.0, occupied_frac=0.0)
    cap = 2048
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    trk = _tracker(cur, cap, 2048)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda th: trk.track_with_motion_model(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], last, th, False, True, False, st)      # noqa: E731
    full = run(15.0)
    assert 3 <= full["n_matches"] and full["n_edges"] == full["n_matches"]
    trk.set_options(min_matches=full["n_matches"] + 1)
    skipped = run(15.0)
    assert skipped["n_matches"] == full["n_matches"] and np.array_equal(skipped["kp_map_point"], full["kp_map_point"])      # the search's tables
    assert skipped["n_edges"] == 0 and skipped["n_inliers"] == 0 and not skipped["kp_outlier"].any() and np.array_equal(skipped["Tcw"], cur["Tcw"])
    trk.set_options(min_matches=full["n_matches"])
    same = run(15.0)
    assert all(np.array_equal(same[k], full[k]) if isinstance(full[k], np.ndarray) else same[k] == full[k] for k in full)


def _distort_keypoints(cur, kps, D):
    """the scene's keypoints are where the map points project: move them to where a lens with coefficients D would image them (the forward model of
    cv::undistortPoints), so that UNDISTORTING them lands near the projections again"""
    k1, k2, p1, p2, k3 = D
    xn = (kps["x"].astype(np.float64) - cur["cx"]) / cur["fx"]; yn = (kps["y"].astype(np.float64) - cur["cy"]) / cur["fy"]
    r2 = xn * xn + yn * yn
    rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn); yd = yn * rad + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    out = kps.copy()
    out["x"] = np.clip(xd * cur["fx"] + cur["cx"], 0.0, 639.0).astype(np.float32); out["y"] = np.clip(yd * cur["fy"] + cur["cy"], 0.0, 479.0).astype(np.float32)
    return out


@pytest.mark.parametrize("case", [dict(seed=7400, cam="TUM1"), dict(seed=7401, cam="TUM2", prior_frac=0.2, th=3.0), dict(seed=7402, cam="TUM1", n=1400, nc=4)])
def test_chain_with_a_distorted_camera(case, oracle):
    """Round 5 (VERDICT r4 missing #6): Frame::UndistortKeyPoints on the device chain.  With the distortion coefficients of the reference's TUM1 / TUM2 camera files
    the frame set-up undistorts the keypoints first; mvuRight is built from the undistorted column and the depth at the DISTORTED pixel, the grid, the search and
    the pose edges read mvKeysUn, the frustum test the undistorted image bounds -- every table as the oracle chain (undistortion included) gives it, bit for bit, the
    pose within the LM bound; and the distortion does change the result (the same frame through a distortion-free tracker differs)."""
    from golden_cases import TUM_CAMERAS
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    D = np.asarray(TUM_CAMERAS[case["cam"]][1][:case.get("nc", 5)], np.float32)
    cur, kps, desc, depth, pts, prior = _scene(case["seed"], n=case.get("n", 900), prior_frac=case.get("prior_frac", 0.0))
    kps = _distort_keypoints(cur, kps, list(D.astype(np.float64)) + [0.0] * (5 - len(D)))
    N = len(kps)
    rng = np.random.default_rng(case["seed"])
    depth[kps["y"].astype(int), kps["x"].astype(int)] = rng.uniform(1.8, 6.2, N).astype(np.float32)      # (a depth at every distorted pixel)
    th, nnratio = case.get("th", 1.0), 0.8
    oc = _OracleCalls(oracle)
    want = _chain(oc, cur, kps, desc, depth, pts, prior, th, nnratio, dist=D)
    bnd = oc.bounds(640, 480, cur["fx"], cur["fy"], cur["cx"], cur["cy"], D)
    cap = 2048
    trk = _tracker(cur, cap, 2048, bnd, D)
    trk.set_local_map(pts)
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, cap)
    st = torch.cuda.current_stream().cuda_stream
    got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, th, nnratio, st)
    assert np.array_equal(got["u_right"], want["u_right"]) and np.array_equal(got["depth"], want["depth"])
    pj = want["projected"]
    assert np.array_equal(got["map_in_view"].astype(bool)[pj], want["in_view"][pj])
    assert got["n_matches"] == want["n_matches"] and want["n_matches"] > 50
    assert np.array_equal(got["kp_map_point"], want["kp_map_point"]) and got["n_edges"] == want["n_edges"]
    assert got["n_inliers"] == want["n_inliers"] and np.array_equal(got["kp_outlier"], want["kp_outlier"])
    ok, err, upd = _pose_close(got["Tcw"], want["Tcw"], cur["Tcw"])
    assert ok, "pose: |gpu - oracle| %.3e vs update %.3e" % (err, upd)
    # ... and the host-hop calls of the product give the same tables
    hop = _chain(_ProductCalls(E), cur, kps, desc, depth, pts, prior, th, nnratio, dist=D)
    assert np.array_equal(hop["kp_map_point"], got["kp_map_point"]) and np.array_equal(hop["u_right"], got["u_right"])
    plain = _tracker(cur, cap, 2048)
    plain.set_local_map(pts)
    other = plain.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, th, nnratio, st)
    assert not np.array_equal(other["u_right"], got["u_right"])
    # a handle switched back to "no distortion" is the plain tracker again
    trk.set_distortion(())
    again = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, th, nnratio, st)
    assert np.array_equal(again["u_right"], other["u_right"])


@pytest.mark.parametrize("case", [dict(seed=7410, cam="TUM1"), dict(seed=7411, cam="TUM2", th=7.0)])
def test_motion_model_stage_with_a_distorted_camera(case, oracle):
    """The motion-model stage on the same frame set-up: SearchByProjection(Cur, Last) and the pose edges read the undistorted keypoints, the image bounds of the search
    windows are the undistorted ones -- against the oracle chain with Frame::UndistortKeyPoints in front."""
    from golden_cases import TUM_CAMERAS
    D = np.asarray(TUM_CAMERAS[case["cam"]][1], np.float32)
    cur, kps, desc, depth, pts, _ = _scene(case["seed"], n=900)
    _, last, _ = synth.synth_tracking(n=900, seed=case["seed"], mono_frac=0.0, occupied_frac=0.0)
    kps = _distort_keypoints(cur, kps, list(D.astype(np.float64)))
    rng = np.random.default_rng(case["seed"])
    depth[kps["y"].astype(int), kps["x"].astype(int)] = rng.uniform(1.8, 6.2, len(kps)).astype(np.float32)
    th = case.get("th", 15.0)
    oc = _OracleCalls(oracle)
    want = _chain_motion(oc, lambda f, l, t, m: oracle.search_by_projection_frames(f, l, t, m, True), cur, kps, desc, depth, last, th, False, True, dist=D)
    bnd = oc.bounds(640, 480, cur["fx"], cur["fy"], cur["cx"], cur["cy"], D)
    d_kps, d_desc, d_n, d_depth = _device_buffers(kps, desc, depth, 2048)
    trk = _tracker(cur, 2048, 2048, bnd, D)
    got = trk.track_with_motion_model(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], last, th, False, True, True,
                                      torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(got["u_right"], want["u_right"]) and np.array_equal(got["depth"], want["depth"])
    assert got["n_matches"] == want["n_matches"] and want["n_matches"] > 100 and got["n_edges"] == want["n_edges"]
    assert np.array_equal(got["kp_map_point"], want["kp_map_point"]) and got["n_inliers"] == want["n_inliers"]
    ok, err, upd = _pose_close(got["Tcw"], want["Tcw"], cur["Tcw"])
    assert ok, "pose |gpu - oracle| %.3e of update %.3e" % (err, upd)
