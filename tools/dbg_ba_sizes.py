# LocalBundleAdjustment wall time vs number of free keyframes, default solver choice vs EAO_BA_SOLVER=big (run twice)
import os, sys, time; sys.path.insert(0, '.')
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
for nf in [int(x) for x in os.environ.get("EAO_DBG_NF", "20,31,40,50,64,80").split(",")]:
    p = synth.synth_ba(n_free=nf, n_fixed=4, n_points=150 * nf, seed=3000 + nf)
    for i in range(2):
        r = E.Optimizer.LocalBundleAdjustment(p)
    ts = []
    for i in range(7):
        t0 = time.perf_counter(); r = E.Optimizer.LocalBundleAdjustment(p); ts.append((time.perf_counter() - t0) * 1e3)
    print("solver=%s nFree %3d E %6d iters %s trials %d  wall ms min %.3f med %.3f" % (os.environ.get("EAO_BA_SOLVER", "default"), nf, len(p["edge_cam"]),
          list(r["iters"]), int(sum(r["trace"]["trials"])), min(ts), sorted(ts)[3]), flush=True)
