# map-scale BundleAdjustment: timing of the k_bal_* path vs the CPU oracle (EAO_DBG_KF / EAO_DBG_PTS / EAO_DBG_ORACLE)
import os, sys, time; sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
nkf = int(os.environ.get("EAO_DBG_KF", "200")); npts = int(os.environ.get("EAO_DBG_PTS", "20000")); its = int(os.environ.get("EAO_DBG_ITS", "10"))
t = time.perf_counter(); p = synth.synth_ba(n_free=nkf, n_fixed=1, n_points=npts, seed=5300); print("synth %.1f s, E = %d" % (time.perf_counter() - t, len(p["edge_cam"])), flush=True)
for k in range(3):
    t = time.perf_counter(); r = E.Optimizer.BundleAdjustment(p, its, bRobust=False); dt = time.perf_counter() - t
    print("GBA %d KF x %d MP: %.2f ms wall (device %.2f ms), iters %s, trials %s" % (nkf, npts, dt * 1e3, r["timing"]["device_ms"], list(r["iters"]), list(r["trace"]["trials"])), flush=True)
if os.environ.get("EAO_DBG_ORACLE"):
    from oracle import oracle as O
    t = time.perf_counter(); o = O.bundle_adjustment(p, its, False); dt = time.perf_counter() - t
    print("oracle %.1f ms, iters %s" % (dt * 1e3, list(o["iters"])))
    upd = np.abs(o["points"] - p["points"]).max()
    print("max |gpu - cpu| points %.3e (update scale %.3e), poses %.3e" % (np.abs(r["points"] - o["points"]).max(), upd, np.abs(r["poses"] - o["poses"]).max()))
