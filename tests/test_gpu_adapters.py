"""The header-only C++ adapters (include/eaofusion/) compiled with g++ against small stand-ins of the reference's
Frame / KeyFrame / MapPoint / Map, driven the way Tracking.cc / LocalMapping.cc drive the reference classes.  Results
must equal what the (already parity-tested) C-ABI returns through the Python mirror."""
import os
import struct
import subprocess

import numpy as np
import pytest

from eao_fusion_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_adapters_end_to_end(tmp_path):
    import eao_fusion_amd as E
    exe = str(tmp_path / "adapter_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_test.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "eao_fusion_amd"), "-leaofusion_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "eao_fusion_amd"), "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    img = synth.synth_frame(1234)
    pp = synth.synth_pose(n=400, seed=4100, n_planes=4)
    bp = synth.synth_ba(n_free=6, n_fixed=3, n_points=400, seed=3100, mono_frac=0.2)
    prob = str(tmp_path / "problem.bin")
    with open(prob, "wb") as f:
        f.write(struct.pack("<ii", *img.shape)); f.write(img.tobytes())
        f.write(struct.pack("<i", len(pp["points"])))
        for k in ("Tcw", "points", "obs", "inv_sigma2"):
            f.write(np.ascontiguousarray(pp[k], np.float32).tobytes())
        f.write(np.array([pp[k] for k in ("fx", "fy", "cx", "cy", "bf")], np.float32).tobytes())
        f.write(struct.pack("<i", len(pp["plane_world"])))
        f.write(pp["plane_world"].tobytes()); f.write(pp["plane_obs"].tobytes()); f.write(pp["plane_seen"].tobytes())
        f.write(struct.pack("<iii", len(bp["poses"]), len(bp["points"]), len(bp["edge_cam"])))
        f.write(bp["poses"].tobytes()); f.write(bp["fixed"].tobytes()); f.write(bp["points"].tobytes())
        f.write(bp["edge_cam"].tobytes()); f.write(bp["edge_point"].tobytes()); f.write(bp["obs"].tobytes()); f.write(bp["inv_sigma2"].tobytes())
        f.write(np.array([bp[k] for k in ("fx", "fy", "cx", "cy", "bf")], np.float32).tobytes())
        sl, sr = synth.synth_stereo_pair(9000)
        f.write(struct.pack("<ii", sl.shape[1], sl.shape[0])); f.write(sl.tobytes()); f.write(sr.tobytes())
    res = str(tmp_path / "result.bin")
    out = subprocess.run([exe, prob, res], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    buf = open(res, "rb").read()
    off = 0

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=n, offset=off)
        off += a.nbytes
        return a
    # ---- ORBextractor
    nk = int(take(np.int32, 1)[0])
    kps = take(E.KP_DTYPE, nk)
    desc = take(np.uint8, nk * 32).reshape(nk, 32)
    pyr = take(np.int32, 5)
    eager_same = int(take(np.int32, 1)[0])
    dd = int(take(np.int32, 1)[0])
    untouched = int(take(np.int32, 1)[0])
    ext = E.ORBextractor(1000, 1.2, 8, 20, 7)
    k2, d2 = ext(img)
    assert np.array_equal(kps, k2) and np.array_equal(desc, d2)
    assert list(pyr) == [480, 640, 179, 1, 0]         # mvImagePyramid[l] is the w x h view (border lives around it); empty until asked for; 0 wrong border pixels
    assert eager_same == 1                            # keepPyramid = true: filled by operator() itself
    assert dd == int(np.unpackbits(desc[0] ^ desc[1]).sum())
    assert untouched == 3
    # ---- ORBmatcher::SearchByProjection over the extractor's own keypoints: (almost) every point re-finds itself
    nm, self_hits = (int(x) for x in take(np.int32, 2))
    assert nm > 0.9 * nk and self_hits > 0.9 * nm
    # ---- PoseOptimization
    inl = int(take(np.int32, 1)[0])
    pfl = take(np.uint8, len(pp["plane_world"]) + 1)
    Tcw = take(np.float32, 16).reshape(4, 4)
    outl = take(np.uint8, len(pp["points"]))
    r = E.Optimizer.PoseOptimization(pp)
    assert inl == r["n_inliers"] and np.array_equal(outl, r["outlier"]) and np.array_equal(Tcw, r["Tcw"])
    assert pfl[0] == 1 and np.array_equal(pfl[1:], r["plane_outlier"])      # the empty plane slot keeps its flag
    # ---- LocalBundleAdjustment (edge insertion order differs from the flat problem => rounding-level differences only)
    nc, npnt = len(bp["poses"]), len(bp["points"])
    poses = take(np.float32, nc * 16).reshape(nc, 4, 4)
    pts = take(np.float32, npnt * 3).reshape(npnt, 3)
    erased, normals, same = (int(x) for x in take(np.int32, 3))
    rb = E.Optimizer.LocalBundleAdjustment(bp)
    assert np.allclose(poses, rb["poses"], rtol=0, atol=2e-6)
    # the reference only gathers map points matched in a LOCAL keyframe (src/Optimizer.cc:693-719): points seen by fixed
    # cameras alone are not part of the window the adapter builds, and keep their value
    free_seen = np.zeros(npnt, bool)
    free_seen[bp["edge_point"][~bp["fixed"].astype(bool)[bp["edge_cam"]]]] = True
    assert free_seen.sum() > 0.9 * npnt
    assert np.allclose(pts[free_seen], rb["points"][free_seen], rtol=0, atol=2e-5)
    assert np.array_equal(pts[~free_seen], bp["points"][~free_seen])
    assert normals == int(free_seen.sum()) and same == 1
    assert erased > 0
    f = bp["fixed"].astype(bool)
    assert np.array_equal(poses[f], bp["poses"][f])    # fixed keyframes are never written back
    # ---- Frame::ComputeStereoMatches through the two ORBextractor adapters
    ns = int(take(np.int32, 1)[0])
    ur_cpp, dp_cpp = take(np.float32, ns), take(np.float32, ns)
    el, er = E.ORBextractor(1000, 1.2, 8, 20, 7), E.ORBextractor(1000, 1.2, 8, 20, 7)
    skl, sdl = el(sl)
    skr, sdr = er(sr)
    ur, dp = E.compute_stereo_matches(el, er, skl, sdl, skr, sdr, np.float32(40.0) / np.float32(535.4), np.float32(40.0))
    assert ns == len(skl) and np.array_equal(ur_cpp, ur) and np.array_equal(dp_cpp, dp) and (ur >= 0).sum() > 100
    # ---- Optimizer::BundleAdjustment (keyframes + map points): only keyframe 0 is fixed, one optimize(10), no Huber kernels
    gposes = take(np.float32, nc * 16).reshape(nc, 4, 4)
    gpts = take(np.float32, npnt * 3).reshape(npnt, 3)
    normals_g = int(take(np.int32, 1)[0])
    parked_poses = take(np.float32, nc * 16).reshape(nc, 4, 4)
    parked = int(take(np.int32, 1)[0])
    plobs = take(np.float32, nc * 4).reshape(nc, 4)
    plposes = take(np.float32, nc * 16).reshape(nc, 4, 4)
    plworld = take(np.float32, 4)
    gp = dict(bp)
    gp["fixed"] = (np.arange(nc) == 0).astype(np.uint8)
    rg = E.Optimizer.BundleAdjustment(gp, 10, bRobust=False)
    assert rg["iters"][0] >= 3
    assert np.allclose(gposes, rg["poses"], rtol=0, atol=2e-6) and np.allclose(gpts, rg["points"], rtol=0, atol=2e-5)
    assert np.array_equal(gposes[0], bp["poses"][0]) and normals_g == npnt
    assert np.array_equal(parked_poses, gposes) and parked == 1
    # ... and with a live MapPlane seen by every keyframe (src/Optimizer.cc:203-252): the template's flattening = this one
    gq = dict(gp)
    gq.update(planes=np.array([[0.12, -0.2, 0.97, 3.4]], np.float32), pedge_plane=np.zeros(nc, np.int32), pedge_cam=np.arange(nc, dtype=np.int32), pedge_obs=plobs)
    rq = E.Optimizer.BundleAdjustment(gq, 10, bRobust=True)
    assert np.allclose(plposes, rq["poses"], rtol=0, atol=2e-6) and np.allclose(plworld, rq["planes"][0], rtol=0, atol=2e-6)
    assert np.abs(rq["planes"][0] - _normalized_plane(gq["planes"][0])).max() > 1e-3 and np.abs(rq["poses"] - rg["poses"]).max() > 1e-5


def _normalized_plane(c):
    c = c.astype(np.float64) / np.linalg.norm(c[:3].astype(np.float64))
    return (-c if c[3] < 0 else c).astype(np.float32)


def test_cpp_search_adapters(tmp_path):
    """The remaining ORBmatcher templates (BoW x2, triangulation, initialisation, projection loop / relocalisation, Sim3,
    Fuse x2) over mock KeyFrame / Frame / MapPoint classes must reproduce what the C-ABI returns for the same data."""
    import torch  # noqa: F401
    import eao_fusion_amd as E
    from eao_fusion_amd import search
    exe = str(tmp_path / "search_adapter_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "search_adapter_test.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "eao_fusion_amd"), "-leaofusion_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "eao_fusion_amd"), "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    sc = synth.synth_search_scene(n=500, seed=8200)
    K1, K2, P = sc["K1"], sc["K2"], sc["points"]
    mp1, mp2 = sc["mp1"], sc["mp2"]
    scene = str(tmp_path / "scene.bin")
    with open(scene, "wb") as f:
        for K, mp, fv in ((K1, mp1, sc["fv1"]), (K2, mp2, sc["fv2"])):
            f.write(struct.pack("<i", len(K["kp_x"])))
            for k in ("kp_x", "kp_y", "kp_angle", "u_right"):
                f.write(np.ascontiguousarray(K[k], np.float32).tobytes())
            f.write(np.ascontiguousarray(K["kp_octave"], np.int32).tobytes()); f.write(np.ascontiguousarray(mp, np.int32).tobytes())
            f.write(np.ascontiguousarray(K["descriptors"], np.uint8).tobytes())
            f.write(struct.pack("<i", len(fv["node_id"])))
            f.write(fv["node_id"].astype(np.uint32).tobytes()); f.write(fv["node_start"].astype(np.int32).tobytes()); f.write(fv["index"].astype(np.uint32).tobytes())
        f.write(struct.pack("<i", len(P["active"])))
        f.write(P["active"].astype(np.uint8).tobytes())
        for k in ("Xw", "normal", "min_dist", "max_dist"):
            f.write(np.ascontiguousarray(P[k], np.float32).tobytes())
        f.write(np.ascontiguousarray(P["descriptors"], np.uint8).tobytes())
        f.write(sc["T1w"].tobytes()); f.write(sc["T2w"].tobytes()); f.write(np.array(sc["K"], np.float32).tobytes())
        f.write(struct.pack("<f", float(sc["bf"])))
        f.write(sc["F12"].tobytes()); f.write(sc["Scw"].tobytes()); f.write(sc["R12"].tobytes()); f.write(sc["t12"].tobytes())
        f.write(K1["scale_factors"].tobytes()); f.write(K1["level_sigma2"].tobytes()); f.write(K1["inv_level_sigma2"].tobytes())
        f.write(struct.pack("<f", float(K1["log_scale_factor"])))
    res = str(tmp_path / "search_result.bin")
    out = subprocess.run([exe, scene, res], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    buf = open(res, "rb").read()
    off = 0

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=n, offset=off)
        off += a.nbytes
        return a

    def take_table():
        n = int(take(np.int32, 1)[0])
        return take(np.int32, n)

    g = search.product()
    act = P["active"].astype(bool)
    v1 = ((mp1 >= 0) & act[np.maximum(mp1, 0)]).astype(np.uint8)
    v2 = ((mp2 >= 0) & act[np.maximum(mp2, 0)]).astype(np.uint8)
    s1 = dict(descriptors=K1["descriptors"], angle=K1["kp_angle"], valid=v1, fv=sc["fv1"])
    s2 = dict(descriptors=K2["descriptors"], angle=K2["kp_angle"], valid=v2, fv=sc["fv2"])
    # 1. SearchByBoW(KF, KF)
    n = int(take(np.int32, 1)[0]); tab = take_table()
    en, m12 = g.search_by_bow(1, s1, s2, 0.75, True)
    assert n == en and np.array_equal(tab, np.where(m12 >= 0, mp2[np.maximum(m12, 0)], -1)) and n > 10
    # 2. SearchByBoW(KF, Frame)
    n = int(take(np.int32, 1)[0]); tab = take_table()
    en, m12 = g.search_by_bow(0, s1, s2, 0.75, True)
    exp = np.full(len(K2["kp_x"]), -1, np.int32)
    exp[m12[m12 >= 0]] = mp1[m12 >= 0]
    assert n == en and np.array_equal(tab, exp)
    # 3. SearchForTriangulation
    k1, k2 = dict(K1), dict(K2)
    k1["occupied"] = ((mp1 >= 0) & (np.arange(len(mp1)) % 2 == 0)).astype(np.uint8)
    k2["occupied"] = ((mp2 >= 0) & (np.arange(len(mp2)) % 3 == 0)).astype(np.uint8)
    n = int(take(np.int32, 1)[0]); tab = take(np.int32, len(mp1))
    en, m12 = g.search_for_triangulation(k1, sc["fv1"], k2, sc["fv2"], sc["F12"], sc["ex"], sc["ey"], 0, True)
    assert n == en and np.array_equal(tab, m12)
    # 4. SearchForInitialization
    pm = np.stack([K1["kp_x"], K1["kp_y"]], 1)
    n = int(take(np.int32, 1)[0]); tab = take(np.int32, len(mp1)); pm_cpp = take(np.float32, 2 * len(mp1)).reshape(-1, 2)
    en, m12, pm_out = g.search_for_initialization(K1, K2, pm, 100, 0.9, True)
    assert n == en and np.array_equal(tab, m12) and np.array_equal(pm_cpp, pm_out)
    # 5. SearchByProjection(KF, Scw, ...)
    kf = dict(K2)
    kf["occupied"] = (np.arange(len(mp2)) % 13 == 0).astype(np.uint8)
    pts = dict(P)
    a5 = P["active"].copy(); a5[0] = 0
    pts["active"] = a5
    n = int(take(np.int32, 1)[0]); tab = take_table()
    en, km = g.search_by_projection_sim3(kf, sc["Scw"], sc["K"], pts, 10)
    assert n == en and np.array_equal(tab, km) and n > 10
    # 6. SearchByProjection(Frame, KF, found, th, ORBdist)
    idx1 = np.maximum(mp1, 0)
    pk = {k: np.ascontiguousarray(P[k][idx1]) for k in ("Xw", "normal", "min_dist_inv", "max_dist_inv", "max_dist", "descriptors")}
    pk["active"] = ((mp1 >= 0) & act[idx1] & (idx1 % 17 != 0)).astype(np.uint8)
    n = int(take(np.int32, 1)[0]); tab = take_table()
    en, cm = g.search_by_projection_kf(K2, sc["T2w"], sc["K"], pk, K1["kp_angle"], 15.0, 100, True)
    assert n == en and np.array_equal(tab, np.where(cm >= 0, mp1[np.maximum(cm, 0)], -1))
    # 7. SearchBySim3
    idx2 = np.maximum(mp2, 0)
    p1 = dict(pk); p1["active"] = v1
    p2 = {k: np.ascontiguousarray(P[k][idx2]) for k in ("Xw", "normal", "min_dist_inv", "max_dist_inv", "max_dist", "descriptors")}
    p2["active"] = v2
    n = int(take(np.int32, 1)[0]); tab = take_table()
    en, m12 = g.search_by_sim3(K1, sc["T1w"], p1, K2, sc["T2w"], p2, sc["K"], 1.0, sc["R12"], sc["t12"], 7.5)
    assert n == en and np.array_equal(tab, np.where(m12 >= 0, mp2[np.maximum(m12, 0)], -1)) and n > 10
    # 8. Fuse(KF, Scw, ...) into an empty keyframe: the first point to reach a keypoint is added, later ones replace
    n = int(take(np.int32, 1)[0]); tab = take(np.int32, len(act))
    en, best = g.fuse_search(K2, 1, sc["Scw"], sc["K"], 0.0, P, 3.0)
    exp = np.full(len(act), -1, np.int32)
    seen = set()
    for i, b in enumerate(best):
        if b >= 0 and int(b) not in seen:
            exp[i] = b
            seen.add(int(b))
    assert n == en and np.array_equal(tab, exp) and n > 10
    # 9. Fuse(KF, points, th): every hit on a keypoint that already has a point replaces, the others add
    T = sc["T2w"].astype(np.float64)
    pose = np.concatenate([T[:3, :3].ravel(), T[:3, 3], -T[:3, :3].T @ T[:3, 3]]).astype(np.float32)
    n, replaced, added = (int(x) for x in take(np.int32, 3))
    en, best = g.fuse_search(K2, 0, pose, sc["K"], sc["bf"], P, 3.0)
    slot = mp2 >= 0
    slot = slot.copy()
    e_add = 0
    for b in best:
        if b >= 0 and not slot[b]:
            slot[b] = True
            e_add += 1
    assert n == en and added == e_add and replaced == en - e_add and n > 10
    # 10. ComputeDistinctiveDescriptors: each point observed by (at most) one keypoint of K1 and one of K2
    tab = take(np.int32, len(act))
    sets = [[] for _ in range(len(act))]
    for k, m in enumerate(mp1):
        if m >= 0:
            sets[m].append(K1["descriptors"][k])
    for k, m in enumerate(mp2):
        if m >= 0:
            sets[m].append(K2["descriptors"][k])
    sets[0] = []
    exp = E.distinctive_descriptors([np.array(x, np.uint8).reshape(-1, 32) for x in sets])
    assert np.array_equal(tab, exp) and exp[0] == -1


def test_cpp_frame_adapters(tmp_path):
    """include/eaofusion/Frame.h (IsInFrustum over a local map with upstream's skip rules, AssignFeaturesToGrid into
    mGrid[64][48], ComputeStereoFromRGBD) over mock Frame / MapPoint classes: the C++ test restates the reference loops
    (src/Frame.cc:597-614, 638-695, 1016-1037) and exits non-zero on any difference."""
    exe = str(tmp_path / "frame_adapter_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "frame_adapter_test.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "eao_fusion_amd"), "-leaofusion_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "eao_fusion_amd"), "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "in view" in out.stderr and "inside the grid" in out.stderr


def test_cpp_device_tracker_adapter(tmp_path):
    """include/eaofusion/DeviceTracker.h (row f1, second half) over mock Frame / MapPoint classes: the adapter's TrackLocalMap
    leaves the frame exactly as a direct eao_tracker_track_local_map call on hand-built arrays does.  Built with hipcc: the test
    owns the device buffers an extractor would hand over."""
    exe = str(tmp_path / "tracker_adapter_test")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "tracker_adapter_test.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "eao_fusion_amd"), "-leaofusion_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "eao_fusion_amd"), "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 disagreements" in out.stderr
