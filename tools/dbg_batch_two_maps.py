"""Two DIFFERENT map-scale windows in one eao_local_ba_batch call, prepared by ONE set-up worker (EAO_BA_BATCH_THREADS=1): each must come out as its own single call."""
import os, sys; sys.path.insert(0, '.')
os.environ.setdefault("EAO_BA_BATCH_THREADS", "1")
import numpy as np, torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
A = synth.synth_ba(n_free=70, n_fixed=2, n_points=2500, seed=5611)
B = synth.synth_ba(n_free=110, n_fixed=1, n_points=4400, seed=5610, band=3)
S = synth.synth_ba(seed=6200)
single = [E.Optimizer.LocalBundleAdjustment(p) for p in (A, S, B)]
batch = E.Optimizer.LocalBundleAdjustmentBatch([A, S, B])
for name, s, b in zip("ASB", single, batch):
    same = np.array_equal(s["poses"], b["poses"]) and np.array_equal(s["points"], b["points"]) and list(s["iters"]) == list(b["iters"])
    print(name, "iters", list(s["iters"]), list(b["iters"]), "identical" if same else "DIFFERENT: max |dpose| %.3e" % np.abs(s["poses"] - b["poses"]).max())
