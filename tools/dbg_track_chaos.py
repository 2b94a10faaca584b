"""Run-to-run reproducibility over MANY ill-conditioned frames (few keypoints, 40 % random prior matches: a handful of inliers, long runs of rejected LM trials):
every frame R times on a fresh handle each, all outputs compared bit for bit with the frame's first run.   python tools/dbg_track_chaos.py [frames] [R] [seed]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import eao_fusion_amd as E
import test_gpu_track as T
F = int(sys.argv[1]) if len(sys.argv) > 1 else 150
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
bad = 0
for f in range(F):
    kw = dict(seed=int(rng.integers(0, 1 << 30)), n=int(rng.integers(60, 400)), prior_frac=float(rng.choice([0.2, 0.4])), mono_frac=float(rng.choice([0.0, 0.25, 1.0])))
    th, nnratio = float(rng.choice([1.0, 5.0])), float(rng.choice([0.6, 0.9]))
    cur, kps, desc, depth, pts, prior = T._scene(**kw)
    cap = 2048
    first = None
    for r in range(R):
        d_kps, d_desc, d_n, d_depth = T._device_buffers(kps, desc, depth, cap)
        trk = T._tracker(cur, cap, 2048); trk.set_local_map(pts)
        got = trk.track_local_map(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_depth.data_ptr(), 640, 640, 480, cur["Tcw"], prior, th, nnratio,
                                  torch.cuda.current_stream().cuda_stream)
        got = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in got.items()}
        if first is None: first = got; continue
        d = [k for k in first if not (np.array_equal(first[k], got[k]) if isinstance(first[k], np.ndarray) else first[k] == got[k])]
        if d:
            bad += 1
            print("NOT REPRODUCIBLE %s th %.1f ratio %.1f run %d: %s (inliers %d / %d)" % (kw, th, nnratio, r, d, got["n_inliers"], first["n_inliers"]), flush=True)
print("chaos: %d frames x %d runs, %d runs differed from their frame's first" % (F, R, bad))
