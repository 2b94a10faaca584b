"""Randomised parity sweep of the two first stages of a tracked frame on the device (round 4) against the CPU oracle's chains: eao_tracker_track_with_motion_model
(ComputeStereoFromRGBD -> SearchByProjection(Cur, Last) with the rotation histogram -> PoseOptimization -> outlier discard) and eao_tracker_track_reference_keyframe
(... -> SearchByBoW(KF, Frame) -> ...): frame sizes, radii / ratios, monocular and stereo frames, forward / backward motion, vocabulary sizes down to one node, orientation
check on / off, discard on / off.  Integer tables bit for bit, the pose within 1e-4 of the update.  Not part of the test suite: run by hand on a GPU box.
    python tools/sweep_track_stages.py [seed] [cases]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import eao_fusion_amd as E  # noqa: F401
from eao_fusion_amd import synth
from oracle import oracle as O
import test_gpu_track as T
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
O.build(); O.lib()
bad = [0, 0]
st = lambda: torch.cuda.current_stream().cuda_stream
def same(got, want, T0):
    return (np.array_equal(got["u_right"], want["u_right"]) and np.array_equal(got["depth"], want["depth"]) and got["n_matches"] == want["n_matches"]
            and got["n_edges"] == want["n_edges"] and np.array_equal(got["kp_map_point"], want["kp_map_point"]) and np.array_equal(got["kp_outlier"], want["kp_outlier"])
            and got["n_inliers"] == want["n_inliers"] and T._pose_close(got["Tcw"], want["Tcw"], T0)[0])
for it in range(N):
    # ---- motion model
    kw = dict(seed=int(rng.integers(0, 1 << 30)), n=int(rng.choice([rng.integers(40, 300), rng.integers(300, 1100), rng.integers(1100, 1700)])), mono_frac=float(rng.choice([0.0, 0.25, 1.0])))
    th, mono, check, discard = float(rng.choice([7.0, 15.0, 30.0])), bool(rng.random() < 0.2), bool(rng.random() < 0.8), bool(rng.random() < 0.8)
    try:
        cur, kps, desc, depth, pts, _ = T._scene(**kw)
        _, last, _ = synth.synth_tracking(n=kw["n"], seed=kw["seed"], mono_frac=0.0, occupied_frac=0.0)
        if rng.random() < 0.3:
            cur = dict(cur); Tm = cur["Tcw"].copy(); Tm[2, 3] = float(rng.choice([-0.4, 0.4])); cur["Tcw"] = Tm
        if mono: depth = np.zeros_like(depth)
        want = T._chain_motion(T._OracleCalls(O), lambda f, l, t, m: O.search_by_projection_frames(f, l, t, m, check), cur, kps, desc, depth, last, th, mono, discard)
        d_kps, d_desc, d_n, d_depth = T._device_buffers(kps, desc, depth, 2048)
        trk = T._tracker(cur, 2048, 2048)
        got = trk.track_with_motion_model(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], last, th, mono, check, discard, st())
        ok, what = same(got, want, cur["Tcw"]), "matches %d / %d, kept %d / %d" % (got["n_matches"], want["n_matches"], got["n_inliers"], want["n_inliers"])
    except Exception as e:  # noqa: BLE001
        ok, what = False, "%s: %s" % (type(e).__name__, e)
    if not ok:
        bad[0] += 1
        print("MISMATCH motion model %s th %.0f mono %s check %s discard %s: %s" % (kw, th, mono, check, discard, what), flush=True)
    # ---- reference keyframe
    kw = dict(seed=int(rng.integers(0, 1 << 30)), n=int(rng.choice([rng.integers(30, 300), rng.integers(300, 1000), rng.integers(1000, 1500)])),
              n_nodes=int(rng.choice([1, 5, 40, 100, 400])), flip=float(rng.choice([0.03, 0.06, 0.1])), clutter=float(rng.choice([0.0, 0.15, 0.5])), mono=bool(rng.random() < 0.2))
    ratio, check, discard = float(rng.choice([0.6, 0.7, 0.9])), bool(rng.random() < 0.8), bool(rng.random() < 0.8)
    try:
        sc, cam, kps, desc, depth, kf = T._bow_case(kw["seed"], kw["n"], kw["n_nodes"], kw["flip"], kw["clutter"], kw["mono"])
        want = T._chain_bow(O, cam, kps, desc, depth, kf, sc["fv2"], ratio, check, discard)
        cap = 2048 if max(len(kps), len(kf["valid"])) <= 2048 else 4096
        d_kps, d_desc, d_n, d_depth = T._device_buffers(kps, desc, depth, cap)
        trk = T._tracker(cam, cap, 2048)
        got = trk.track_reference_keyframe(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cam["Tcw"], kf, sc["fv2"], ratio, check, discard, st())
        ok, what = same(got, want, cam["Tcw"]), "matches %d / %d, kept %d / %d" % (got["n_matches"], want["n_matches"], got["n_inliers"], want["n_inliers"])
    except Exception as e:  # noqa: BLE001
        ok, what = False, "%s: %s" % (type(e).__name__, e)
    if not ok:
        bad[1] += 1
        print("MISMATCH reference keyframe %s ratio %.1f check %s discard %s: %s" % (kw, ratio, check, discard, what), flush=True)
print("tracker stage sweep: %d motion-model frames, %d mismatches; %d reference-keyframe frames, %d mismatches" % (N, bad[0], N, bad[1]))
