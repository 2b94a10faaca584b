"""Stress test of the tracker's result hand-over (VERDICT r3 weak #2 / ADVICE r3 medium): N calls of eao_tracker_track_local_map in which
CONSECUTIVE CALLS ON A HANDLE DIFFER (another prior table / window factor / ratio), so a result block whose sections belong to two different
calls -- what a reordered hand-over would produce -- cannot pass for a correct one.  Every call is compared bit for bit with the first result
its (scene, variant) produced; the first result of every (scene, variant) is compared with the oracle chain.  The mode is the library's
EAO_TRACK_POLL (1: the host polls the done word, 0: hipStreamSynchronize), read from the environment as the library does.

    EAO_TRACK_POLL=1 python tools/stress_track_poll.py [calls] [scenes] [seed]"""
import os
import sys
import time

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
import eao_fusion_amd as E  # noqa: F401
from oracle import oracle as O
import test_gpu_track as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 17)
mode = os.environ.get("EAO_TRACK_POLL", "1")
cap = 2048
scenes = []
for s in range(S):
    kw = dict(seed=int(rng.integers(0, 1 << 30)), n=int(rng.choice([150, 400, 900])), prior_frac=0.3, mono_frac=float(rng.choice([0.0, 0.25])))
    cur, kps, desc, depth, pts, prior = T._scene(**kw)
    d = T._device_buffers(kps, desc, depth, cap)
    trk = T._tracker(cur, cap, 2048); trk.set_local_map(pts)
    variants = []
    for v in range(4):
        pr = prior.copy()
        drop = rng.random(len(pr)) < 0.25 * v          # variant v keeps fewer of the prior matches: other edges, other matches, other outliers
        pr[drop] = -1
        variants.append(dict(prior=pr, th=float([1.0, 3.0, 5.0, 2.0][v]), nnratio=float([0.8, 0.9, 0.6, 0.7][v]), want=None))
    scenes.append(dict(kw=kw, cur=cur, kps=kps, desc=desc, depth=depth, pts=pts, dev=d, trk=trk, var=variants))


def call(sc, va):
    d_kps, d_desc, d_n, d_depth = sc["dev"]
    got = sc["trk"].track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, sc["cur"]["Tcw"], va["prior"], va["th"],
                                    va["nnratio"], torch.cuda.current_stream().cuda_stream)
    return {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in got.items()}


# reference results: first call of every (scene, variant), checked against the oracle chain
oracle_bad = 0
for sc in scenes:
    for va in sc["var"]:
        va["want"] = call(sc, va)
        want = T._chain(T._OracleCalls(O), sc["cur"], sc["kps"], sc["desc"], sc["depth"], sc["pts"], va["prior"], va["th"], va["nnratio"])
        ok, err, upd = T._pose_close(va["want"]["Tcw"], want["Tcw"], sc["cur"]["Tcw"])
        same = ok and np.array_equal(va["want"]["kp_map_point"], want["kp_map_point"]) and np.array_equal(va["want"]["kp_outlier"], want["kp_outlier"]) \
            and va["want"]["n_inliers"] == want["n_inliers"]
        oracle_bad += 0 if same else 1
distinct = sum(1 for sc in scenes for a in range(4) for b in range(a) if not np.array_equal(sc["var"][a]["want"]["kp_map_point"], sc["var"][b]["want"]["kp_map_point"]))
print("mode EAO_TRACK_POLL=%s: %d scenes x 4 variants; %d of them differ from the oracle chain; %d of %d variant pairs have different match tables" % (
    mode, S, oracle_bad, distinct, S * 6), flush=True)
bad, t0 = 0, time.time()
fields = {}
for it in range(N):
    sc = scenes[int(rng.integers(S))]
    va = sc["var"][int(rng.integers(4))]
    got = call(sc, va)
    d = [k for k in va["want"] if not (np.array_equal(va["want"][k], got[k]) if isinstance(got[k], np.ndarray) else va["want"][k] == got[k])]
    if d:
        bad += 1
        for k in d: fields[k] = fields.get(k, 0) + 1
        if bad <= 10: print("MISMATCH call %d scene %s variant th %.1f: %s" % (it, sc["kw"], va["th"], d), flush=True)
    if (it + 1) % 20000 == 0: print("  %d calls, %d mismatches, %.1f s" % (it + 1, bad, time.time() - t0), flush=True)
dt = time.time() - t0
print("stress EAO_TRACK_POLL=%s: %d calls (consecutive calls on a handle differ), %d mismatches %s, %.1f s = %.3f ms per call incl. Python" % (
    mode, N, bad, fields or "", dt, 1e3 * dt / max(N, 1)))
sys.exit(1 if bad or oracle_bad else 0)
