import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
for kw in [dict(), dict(n_free=5, n_fixed=2, n_points=300), dict(mono_frac=0.4, seed=3001), dict(n_free=3, n_fixed=0, n_points=120, seed=3002),
           dict(n_free=20, n_fixed=4, n_points=3000, sigma=0.0, outlier_frac=0.0), dict(n_free=7, n_fixed=2, n_points=400, seed=3003),
           dict(n_free=28, n_fixed=3, n_points=1500, seed=3006), dict(n_free=34, n_fixed=2, n_points=1500, seed=3005),
           dict(n_free=6, n_fixed=2, n_points=300, seed=3010, sigma=3.0, outlier_frac=0.3)]:
    p = synth.synth_ba(**kw)
    if kw.get("n_fixed", 4) == 0: p["fixed"][0] = 1
    r = E.Optimizer.LocalBundleAdjustment(p)
    print(kw, list(r["iters"]), list(r["trace"]["trials"]))
