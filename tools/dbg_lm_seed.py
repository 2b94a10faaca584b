"""LM trace of one far-off window (tests/test_gpu_lm.py::test_local_ba_rejected_trials) next to the oracle's: python tools/dbg_lm_seed.py <seed> [more seeds]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
from oracle import oracle as O
for seed in [int(a) for a in sys.argv[1:]] or [3037]:
    p = synth.synth_ba(n_free=5, n_fixed=2, n_points=200, seed=seed, rot_noise_deg=25, trans_noise=0.8, point_noise=1.0, mono_frac=0.7)
    r = E.Optimizer.LocalBundleAdjustment(p); o = O.local_ba(p)
    print("seed", seed, "iters gpu", list(r["iters"]), "oracle", list(o["iters"]))
    tg, to = r["trace"], o["trace"]
    for k in range(max(len(tg["chi2"]), len(to["chi2"]))):
        g = (tg["trials"][k], tg["chi2"][k], tg["lam"][k]) if k < len(tg["chi2"]) else None
        c = (to["trials"][k], to["chi2"][k], to["lam"][k]) if k < len(to["chi2"]) else None
        print("  it %2d gpu %s  oracle %s" % (k, g, c))
    upd = np.abs(o["points"] - p["points"]).max()
    print("  points |gpu - oracle| %.3e of update %.3e; outlier tables equal: %s" % (np.abs(r["points"] - o["points"]).max(), upd, np.array_equal(r["edge_outlier"], o["edge_outlier"])))
