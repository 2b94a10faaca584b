# map-scale BundleAdjustment on a BANDED map (each keyframe covisible with its +-(band - 1) neighbours): timing, device memory, parity on a small case
#   EAO_DBG_KF (1000) EAO_DBG_PTS (50000) EAO_DBG_BAND (11) EAO_DBG_ITS (10)
import os, sys, time; sys.path.insert(0, '.')
import numpy as np
import torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
nkf = int(os.environ.get("EAO_DBG_KF", "1000")); npts = int(os.environ.get("EAO_DBG_PTS", "50000")); its = int(os.environ.get("EAO_DBG_ITS", "10")); band = int(os.environ.get("EAO_DBG_BAND", "11"))
if os.environ.get("EAO_DBG_ORACLE", "1") != "0":
    from oracle import oracle as O
    for kw in (dict(n_free=60, n_fixed=1, n_points=3000, seed=5401, band=7), dict(n_free=45, n_fixed=2, n_points=2000, seed=5402, band=4)):
        p = synth.synth_ba(**kw)
        r = E.Optimizer.BundleAdjustment(p, 10, bRobust=False); o = O.bundle_adjustment(p, 10, False)
        upd = np.abs(o["points"] - p["points"]).max()
        print("parity %s: iters %s / %s, max |gpu - cpu| points %.3e (update %.3e), poses %.3e" % (kw, list(r["iters"]), list(o["iters"]), np.abs(r["points"] - o["points"]).max(), upd,
              np.abs(r["poses"] - o["poses"]).max()), flush=True)
t = time.perf_counter(); p = synth.synth_ba(n_free=nkf, n_fixed=1, n_points=npts, seed=5400, band=band); print("synth %.1f s, E = %d" % (time.perf_counter() - t, len(p["edge_cam"])), flush=True)
free0 = torch.cuda.mem_get_info()[0]
for k in range(3):
    t = time.perf_counter(); r = E.Optimizer.BundleAdjustment(p, its, bRobust=False); dt = time.perf_counter() - t
    used = (free0 - torch.cuda.mem_get_info()[0]) / 1e6
    print("banded GBA %d KF x %d MP (band %d): %.2f ms wall (device %.2f ms), iters %s, trials %s, device memory of the call %.0f MB" % (
        nkf, npts, band, dt * 1e3, r["timing"]["device_ms"], list(r["iters"]), list(r["trace"]["trials"]), used), flush=True)
