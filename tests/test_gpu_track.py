"""Row f1, second half: the device-resident tracked frame.  eao_tracker_track_local_map chains ComputeStereoFromRGBD +
AssignFeaturesToGrid -> isInFrustum over the local map -> SearchByProjection(points) -> PoseOptimization on the device behind
the extractor's outputs; it must give, bit for bit, what the host-hop calls of the same C-ABI give on the same data."""
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


def _host_hop(E, cur, kps, desc, depth, pts, prior, th, nnratio):
    from eao_fusion_amd import frame as FR
    N, M = len(kps), len(pts["Xw"])
    kx, ky = np.ascontiguousarray(kps["x"]), np.ascontiguousarray(kps["y"])
    ur, dz = FR.compute_stereo_from_rgbd(kx, ky, kx, depth, cur["mbf"])
    T = np.ascontiguousarray(cur["Tcw"], np.float32)
    Ow = (-(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))).astype(np.float32)
    logsf = float(np.log(np.float32(1.2)))
    fr = dict(Tcw=T, Ow=Ow, fx=cur["fx"], fy=cur["fy"], cx=cur["cx"], cy=cur["cy"], mbf=cur["mbf"], min_x=0.0, max_x=640.0, min_y=0.0, max_y=480.0,
              log_scale_factor=np.float32(logsf))
    fo = FR.is_in_frustum(fr, pts, 0.5)
    occupied = np.zeros(N, np.uint8)
    skip = (~fo["in_view"].astype(bool)) | (~pts["active"].astype(bool))
    kp_mp = np.full(N, -1, np.int32)
    if prior is not None:
        occupied[prior >= 0] = 1
        skip[prior[prior >= 0]] = True
        kp_mp[:] = prior
    frame = dict(kp_x=kx, kp_y=ky, kp_octave=np.ascontiguousarray(kps["octave"]), kp_angle=np.ascontiguousarray(kps["angle"]), u_right=ur,
                 descriptors=desc, occupied=occupied, min_x=np.float32(0), min_y=np.float32(0), max_x=np.float32(640), max_y=np.float32(480),
                 scale_factors=cur["scale_factors"])
    lvl = np.where(skip, 0, fo["pred_level"]).astype(np.int32)
    mps = dict(proj_x=fo["proj_x"], proj_y=fo["proj_y"], proj_xr=fo["proj_xr"], view_cos=fo["view_cos"], level=lvl, descriptors=pts["descriptors"],
               skip=skip.astype(np.uint8))
    nm, match = E.ORBmatcher(nnratio, True).SearchByProjectionPoints(frame, mps, th)
    for m in range(M):
        if match[m] >= 0:
            kp_mp[match[m]] = m
    ks = np.nonzero(kp_mp >= 0)[0]
    inv_sigma2 = (np.float32(1.0) / (cur["scale_factors"] * cur["scale_factors"])).astype(np.float32)
    res = dict(n_matches=nm, kp_map_point=kp_mp, u_right=ur, depth=dz, n_edges=len(ks))
    prob = dict(Tcw=T, points=pts["Xw"][kp_mp[ks]], obs=np.stack([kx[ks], ky[ks], ur[ks]], 1).astype(np.float32),
                inv_sigma2=inv_sigma2[kps["octave"][ks]], fx=cur["fx"], fy=cur["fy"], cx=cur["cx"], cy=cur["cy"], bf=cur["mbf"])
    r = E.Optimizer.PoseOptimization(prob)
    outl = np.zeros(N, np.uint8)
    outl[ks] = r["outlier"]
    res.update(Tcw=r["Tcw"], n_inliers=r["n_inliers"], kp_outlier=outl)
    return res


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
