#!/usr/bin/env python3
"""How far the LM trace (lambda, chi2 per iteration) of the GPU is from the oracle's on the parity problems of
tests/test_gpu_lm.py, next to how far the ORACLE's own trace moves when its inputs are perturbed by one float32 ulp --
the quantity behind _check_trace's tolerances (chi2 1e-6 / 1e-4 relative, lambda 2e-3 relative, comparison stops at the
first stalled iteration).  lambda is multiplied by max(1/3, 1 - (2 rho - 1)^3) every accepted step, with
rho = (chi_old - chi_new) / scale a ratio of small differences: a relative error eps on chi becomes eps * chi / (chi_old -
chi_new) on rho, so lambda is looser than chi2 by construction.  Run on a GPU box:
    python3 tools/lm_trace_sensitivity.py > profiles/r02_lm_trace_sensitivity.txt"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
from oracle import oracle as O

O.build()
CASES = [dict(), dict(n_free=5, n_fixed=2, n_points=300), dict(mono_frac=0.4, seed=3001), dict(n_free=3, n_fixed=0, n_points=120, seed=3002),
         dict(n_free=20, n_fixed=4, n_points=3000, sigma=0.0, outlier_frac=0.0), dict(n_free=7, n_fixed=2, n_points=400, seed=3003),
         dict(n_free=30, n_fixed=3, n_points=1500, seed=3004), dict(n_free=34, n_fixed=2, n_points=1500, seed=3005)]


def ulp(p, seed):
    rng = np.random.default_rng(seed)
    q = dict(p)
    for k in ("points", "obs", "poses"):
        v = p[k].copy()
        d = rng.integers(-1, 2, v.shape)
        if k == "poses":
            d[:, 3, :] = 0
        w = np.where(d > 0, np.nextafter(v, np.float32(np.inf)), np.where(d < 0, np.nextafter(v, np.float32(-np.inf)), v)).astype(np.float32)
        q[k] = np.where(p[k] < 0, p[k], w) if k == "obs" else w
    return q


def trace_dev(a, b):
    """max relative deviation of chi2 / lambda over the well-conditioned prefix (the rule of tests/test_gpu_lm.py::_check_trace)"""
    ta, tb = a["trace"], b["trace"]
    n = min(len(ta["chi2"]), len(tb["chi2"]))
    dc = dl = 0.0
    prev = None
    used = 0
    for k in range(n):
        c = tb["chi2"][k]
        if (prev is not None and abs(prev - c) <= 1e-6 * max(abs(prev), 1e-12)) or c < 1e-6:
            break
        dc = max(dc, abs(ta["chi2"][k] - c) / abs(c))
        dl = max(dl, abs(ta["lam"][k] - tb["lam"][k]) / abs(tb["lam"][k]))
        prev = c
        used += 1
    return dc, dl, used, n


print(__doc__)
print("%-58s | %-34s | %-34s" % ("LocalBundleAdjustment problem (synth_ba arguments)", "oracle + 1 ulp vs oracle", "GPU vs oracle"))
print("%-58s | %-10s %-10s %-11s | %-10s %-10s %-11s" % ("", "chi2", "lambda", "iterations", "chi2", "lambda", "iterations"))
worst = [0, 0, 0, 0]
for kw in CASES:
    p = synth.synth_ba(**kw)
    if kw.get("n_fixed", 4) == 0:
        p["fixed"][0] = 1
    a = O.local_ba(p)
    bc = bl = 0.0
    for draw in range(3):
        c, l, used, n = trace_dev(O.local_ba(ulp(p, 77 + draw)), a)
        bc, bl = max(bc, c), max(bl, l)
    g = E.Optimizer.LocalBundleAdjustment(p)
    gc, gl, gused, gn = trace_dev(g, a)
    worst = [max(worst[0], bc), max(worst[1], bl), max(worst[2], gc), max(worst[3], gl)]
    print("%-58s | %-10.2e %-10.2e %-11s | %-10.2e %-10.2e %-11s" % (str(kw)[:58], bc, bl, "%d of %d" % (used, n), gc, gl, "%d of %d" % (gused, gn)))
print()
print("worst case: oracle + 1 ulp: chi2 %.2e lambda %.2e;  GPU: chi2 %.2e lambda %.2e  (test bounds: chi2 1e-6, lambda 2e-3)" % tuple(worst))
