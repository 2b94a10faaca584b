"""(tools/ab_gba_job.sh runs the three steps below on the GPU box against gpurun_ab/libeaofusion_hip_head.so, a build of the previous commit.)
Bit-for-bit A/B of the map-scale BundleAdjustment between two builds of the library (EAO_LIB_PATH selects one): poses, points and the LM trace of a few maps.
    EAO_LIB_PATH=gpurun_ab/libeaofusion_hip_head.so python tools/ab_gba_bits.py dump gpurun_out/gba_head.npz; python tools/ab_gba_bits.py dump gpurun_out/gba_new.npz
    python tools/ab_gba_bits.py cmp gpurun_out/gba_head.npz gpurun_out/gba_new.npz"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not (a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes())]
    print("%d arrays, %d differ bit for bit%s" % (len(a.files), len(bad), (": " + ", ".join(bad[:12])) if bad else ""))
    sys.exit(1 if bad else 0)
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
out = {}
cases = [dict(n_free=45, n_fixed=2, n_points=2000, seed=5402, band=4), dict(n_free=60, n_fixed=1, n_points=3000, seed=5401, band=7), dict(n_free=40, n_fixed=1, n_points=1500, seed=5403),
         dict(n_free=200, n_fixed=1, n_points=20000, seed=5300), dict(n_free=400, n_fixed=1, n_points=20000, seed=5404, band=11)]
for i, kw in enumerate(cases):
    p = synth.synth_ba(**kw)
    t = time.perf_counter(); r = E.Optimizer.BundleAdjustment(p, 10, bRobust=False); dt = time.perf_counter() - t
    r = E.Optimizer.BundleAdjustment(p, 10, bRobust=False)
    out["c%d_poses" % i] = np.asarray(r["poses"]); out["c%d_points" % i] = np.asarray(r["points"])
    for k, v in r["trace"].items(): out["c%d_t_%s" % (i, k)] = np.asarray(v)
    print(kw, "iters", list(r["iters"]), "%.2f ms first call" % (dt * 1e3), flush=True)
np.savez(sys.argv[2], **out)
print("dumped %d arrays to %s" % (len(out), sys.argv[2]))
