import sys; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
from oracle import oracle as O
for seed in [int(a) for a in sys.argv[1:]]:
    p = synth.synth_ba(n_free=5, n_fixed=2, n_points=200, seed=seed, rot_noise_deg=25, trans_noise=0.8, point_noise=1.0, mono_frac=0.7)
    r = E.Optimizer.LocalBundleAdjustment(p); o = O.local_ba(p)
    print(seed, "iters", list(map(int, r["iters"])), list(map(int, o["iters"])))
    print("  trials gpu", list(map(int, r["trace"]["trials"]))); print("  trials cpu", list(map(int, o["trace"]["trials"])))
    print("  chi gpu", ["%.6g" % c for c in r["trace"]["chi2"]]); print("  chi cpu", ["%.6g" % c for c in o["trace"]["chi2"]])
    print("  lam gpu", ["%.4g" % c for c in r["trace"]["lam"]]); print("  lam cpu", ["%.4g" % c for c in o["trace"]["lam"]])
