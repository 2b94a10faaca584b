"""Run-to-run reproducibility of eao_tracker_track_local_map: the same frame N times, every output compared bit for bit with the first run's
(and with the oracle chain once).   python tools/dbg_track_repeat.py [N]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import eao_fusion_amd as E
from oracle import oracle as O
import test_gpu_track as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cases = [dict(kw=dict(seed=929732905, n=134, prior_frac=0.4, mono_frac=0.25), th=5.0, nnratio=0.9),
         dict(kw=dict(seed=7101, n=900, prior_frac=0.3), th=1.0, nnratio=0.8)]
for c in cases:
    cur, kps, desc, depth, pts, prior = T._scene(**c["kw"])
    want = T._chain(T._OracleCalls(O), cur, kps, desc, depth, pts, prior, c["th"], c["nnratio"])
    cap = 2048
    d_kps, d_desc, d_n, d_depth = T._device_buffers(kps, desc, depth, cap)
    first, diff = None, {}
    for it in range(N):
        if it % 50 == 0:      # a fresh handle every 50 calls (the sweeps build one per frame)
            trk = T._tracker(cur, cap, 2048); trk.set_local_map(pts)
        got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, c["th"], c["nnratio"],
                                  torch.cuda.current_stream().cuda_stream)
        got = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in got.items()}
        if first is None: first = got; continue
        for k in first:
            same = np.array_equal(first[k], got[k]) if isinstance(first[k], np.ndarray) else first[k] == got[k]
            if not same: diff[k] = diff.get(k, 0) + 1
    ok, err, upd = T._pose_close(first["Tcw"], want["Tcw"], cur["Tcw"])
    print("%s: %d runs, fields that differed from the first run: %s; first run vs oracle: pose %s (|d| %.3e, update %.3e), outlier table %s, inliers %d / %d" % (
        c["kw"], N, diff or "none", "ok" if ok else "OUTSIDE", err, upd, np.array_equal(first["kp_outlier"], want["kp_outlier"]), first["n_inliers"], want["n_inliers"]))
