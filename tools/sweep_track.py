"""Randomised parity sweep of the device-resident tracked frame (eao_tracker_track_local_map) against the CPU oracle's chain
(ComputeStereoFromRGBD -> isInFrustum -> SearchByProjection(points) -> PoseOptimization): frame sizes, local-map sizes up to 16 384
points, search radii, ratio thresholds, prior matches (on active, bad and foreign points), monocular fractions, and (round 5) a third of the frames through a camera
with random lens distortion (Frame::UndistortKeyPoints on the chain).  Integer tables bit for
bit, the pose within 1e-4 of the update.  Not part of the test suite: run by hand on a GPU box.
    python tools/sweep_track.py [seed] [cases]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import eao_fusion_amd as E
from oracle import oracle as O
import test_gpu_track as T
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(N):
    kw = dict(seed=int(rng.integers(0, 1 << 30)), n=int(rng.choice([rng.integers(60, 300), rng.integers(300, 1100), rng.integers(1100, 1800)])),
              prior_frac=float(rng.choice([0.0, 0.0, 0.1, 0.4])), mono_frac=float(rng.choice([0.0, 0.25, 1.0])))
    th, nnratio = float(rng.choice([1.0, 3.0, 5.0, 8.0])), float(rng.choice([0.6, 0.8, 0.9]))
    cur, kps, desc, depth, pts, prior = T._scene(**kw)
    M = len(pts["Xw"])
    extra = int(rng.choice([0, 0, 3000, 9000, 15000]))
    extra = min(extra, 16384 - M)
    if extra > 0:      # a larger local map around the scene, the real points spread over the whole index range
        Xe = rng.uniform([-8, -6, -2], [8, 6, 10], (extra, 3)).astype(np.float32)
        de = np.maximum(np.linalg.norm(Xe, axis=1), 0.1).astype(np.float32)
        more = dict(active=(rng.random(extra) < 0.9).astype(np.uint8), Xw=Xe, normal=(Xe / de[:, None]).astype(np.float32), min_dist_inv=(0.6 * de).astype(np.float32),
                    max_dist_inv=(1.7 * de).astype(np.float32), max_dist=(de * np.float32(1.2) ** 3).astype(np.float32),
                    descriptors=rng.integers(0, 256, (extra, 32), dtype=np.uint8))
        order = rng.permutation(M + extra)
        inv = np.empty_like(order); inv[order] = np.arange(M + extra)
        pts = {k: np.ascontiguousarray(np.concatenate([pts[k], more[k]])[order]) for k in more}
        if prior is not None: prior = np.where(prior >= 0, inv[np.maximum(prior, 0)], prior).astype(np.int32)
    prior_Xw = None
    if prior is not None and rng.random() < 0.5:      # some prior matches on points the local map does not hold
        free = np.nonzero(prior < 0)[0]
        out = free[:int(rng.integers(1, 30))]
        prior[out] = -2
        prior_Xw = np.zeros((len(kps), 3), np.float32)
        Tm = cur["Tcw"].astype(np.float64)
        z = rng.uniform(2.0, 5.0, len(out))
        Xc = np.stack([(kps["x"][out] - cur["cx"]) * z / cur["fx"], (kps["y"][out] - cur["cy"]) * z / cur["fy"], z], 1)
        prior_Xw[out] = ((Xc - Tm[:3, 3]) @ Tm[:3, :3]).astype(np.float32)
    dist, bnd = None, (0.0, 640.0, 0.0, 480.0)
    if rng.random() < 0.34 and prior_Xw is None:      # a distorted camera: k1, k2, p1, p2 [, k3] around the reference's TUM1 / TUM2 values
        dist = np.array([rng.uniform(0.05, 0.3), rng.uniform(-1.0, -0.1), rng.uniform(-0.006, 0.006), rng.uniform(-0.003, 0.003), rng.uniform(0.5, 1.2)], np.float32)
        if rng.random() < 0.4: dist = dist[:4]
        bnd = T._OracleCalls(O).bounds(640, 480, cur["fx"], cur["fy"], cur["cx"], cur["cy"], dist)
        if not (bnd[1] - bnd[0] > 300 and bnd[3] - bnd[2] > 200):      # coefficients whose model folds over inside the image (no lens): a tamer draw
            dist = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32) * np.float32(rng.uniform(0.3, 1.0))
            bnd = T._OracleCalls(O).bounds(640, 480, cur["fx"], cur["fy"], cur["cx"], cur["cy"], dist)
        kps = T._distort_keypoints(cur, kps, list(dist.astype(np.float64)) + [0.0] * (5 - len(dist)))
        depth[kps["y"].astype(int), kps["x"].astype(int)] = rng.uniform(1.8, 6.2, len(kps)).astype(np.float32)
    try:
        want = T._chain(T._OracleCalls(O), cur, kps, desc, depth, pts, prior, th, nnratio, prior_Xw, dist=dist)
        cap = 2048
        trk = T._tracker(cur, cap, 16384, bnd, dist)
        trk.set_local_map(pts)
        d_kps, d_desc, d_n, d_depth = T._device_buffers(kps, desc, depth, cap)
        got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, th, nnratio,
                                  torch.cuda.current_stream().cuda_stream, prior_Xw)
        pj = want["projected"]
        ok = (np.array_equal(got["u_right"], want["u_right"]) and np.array_equal(got["depth"], want["depth"])
              and np.array_equal(got["map_in_view"].astype(bool)[pj], want["in_view"][pj]) and got["n_matches"] == want["n_matches"]
              and np.array_equal(got["kp_map_point"], want["kp_map_point"]) and got["n_edges"] == want["n_edges"] and got["n_inliers"] == want["n_inliers"]
              and np.array_equal(got["kp_outlier"], want["kp_outlier"]) and T._pose_close(got["Tcw"], want["Tcw"], cur["Tcw"])[0])
        what = "matches %d / %d, edges %d / %d, inliers %d / %d" % (got["n_matches"], want["n_matches"], got["n_edges"], want["n_edges"], got["n_inliers"], want["n_inliers"])
        if not ok:
            pc = T._pose_close(got["Tcw"], want["Tcw"], cur["Tcw"])
            what += "; differing: %s; pose |d| %.3e vs update %.3e" % ([k for k, a, b in (("u_right", got["u_right"], want["u_right"]), ("depth", got["depth"], want["depth"]),
                ("in_view", got["map_in_view"].astype(bool)[pj], want["in_view"][pj]), ("kp_map_point", got["kp_map_point"], want["kp_map_point"]),
                ("kp_outlier", got["kp_outlier"], want["kp_outlier"])) if not np.array_equal(a, b)], pc[1], pc[2])
        if not ok and not np.array_equal(got["kp_map_point"], want["kp_map_point"]):
            ks = np.nonzero(got["kp_map_point"] != want["kp_map_point"])[0]
            what += "; mvpMapPoints differs at keypoints %s: got %s, oracle %s" % (ks[:6], got["kp_map_point"][ks[:6]], want["kp_map_point"][ks[:6]])
    except Exception as e:
        ok, what = False, "%s: %s" % (type(e).__name__, e)
    if not ok:
        bad += 1
        print("MISMATCH %s th %.1f nnratio %.1f map %d (+%d) outside %s: %s" % (kw, th, nnratio, M, extra, prior_Xw is not None, what), flush=True)
print("tracker sweep: %d frames, %d mismatches" % (N, bad))
