#!/usr/bin/env python3
"""Which seeds of tests/test_gpu_lm.py::test_local_ba_rejected_trials are held to the oracle's own one-ulp band (CHAOTIC_BAND) instead of 1e-4.

For every seed of the family (far-off starts: 25 degrees, 0.8 m, 1 m on the points, 70 % monocular edges) three runs:
  A  the CPU oracle on the problem as generated
  B  the SAME oracle on the problem with EVERY input (points, poses, observations) moved by a random -1 / 0 / +1 float32 ulp
     -- what a different summation order or a differently rounded reciprocal does to the intermediate values of another
     implementation, applied at the inputs; three draws, the worst is reported
  G  the GPU path on the problem as generated
and, for B and G against A: do the LM schedules agree (iterations of both optimize() calls, outlier table), and how far are
the optimised poses / points apart relative to the size of the update.  A seed where B -- the same code, the same order of
operations, a 1-ulp input change -- already disagrees with A by more than the parity bar cannot be used to compare ANY two
implementations: its first LM iterations amplify rounding (points flip behind cameras, chi2 ~ 1e6).  Run on a GPU box:
    python3 tools/lm_chaotic_seeds.py > profiles/r02_lm_chaotic_seeds.txt"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
from oracle import oracle as O

O.build()


def rel(a, b, old):
    upd = max(np.abs(a.astype(np.float64) - old.astype(np.float64)).max(), 1e-6)
    return np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / upd


def same_schedule(x, y):
    return list(x["iters"]) == list(y["iters"]) and np.array_equal(x["edge_outlier"], y["edge_outlier"])


print(__doc__)
print("%-6s | %-34s | %-34s | verdict" % ("seed", "B: oracle + 1 ulp vs oracle", "G: GPU vs oracle"))
print("%-6s | %-10s %-11s %-11s | %-10s %-11s %-11s |" % ("", "schedule", "poses", "points", "schedule", "poses", "points"))
seeds = list(range(3030, 3060))
n_bad_b = n_bad_g = n_unexplained = 0
for sd in seeds:
    p = synth.synth_ba(n_free=5, n_fixed=2, n_points=200, seed=sd, rot_noise_deg=25, trans_noise=0.8, point_noise=1.0, mono_frac=0.7)
    a = O.local_ba(p)
    sb, rb = True, (0.0, 0.0)
    for draw in range(3):
        rng = np.random.default_rng(100 * sd + draw)
        q = dict(p)
        for k in ("points", "obs"):
            v = p[k].copy()
            d = rng.integers(-1, 2, v.shape)
            v = np.where(d > 0, np.nextafter(v, np.float32(np.inf)), np.where(d < 0, np.nextafter(v, np.float32(-np.inf)), v)).astype(np.float32)
            q[k] = np.where(p[k] < 0, p[k], v) if k == "obs" else v          # (a negative uR marks a monocular edge: leave it)
        v = p["poses"].copy()
        d = rng.integers(-1, 2, v.shape)
        d[:, 3, :] = 0                                                          # (the last row stays 0 0 0 1)
        q["poses"] = np.where(d > 0, np.nextafter(v, np.float32(np.inf)), np.where(d < 0, np.nextafter(v, np.float32(-np.inf)), v)).astype(np.float32)
        b = O.local_ba(q)
        sb = sb and same_schedule(a, b)
        rb = (max(rb[0], rel(a["poses"], b["poses"], p["poses"])), max(rb[1], rel(a["points"], b["points"], p["points"])))
    g = E.Optimizer.LocalBundleAdjustment(p)
    sg = same_schedule(a, g)
    rg = (rel(a["poses"], g["poses"], p["poses"]), rel(a["points"], g["points"], p["points"]))
    bad_b = (not sb) or max(rb) > 1e-4
    bad_g = (not sg) or max(rg) > 1e-4
    n_bad_b += bad_b; n_bad_g += bad_g; n_unexplained += (bad_g and not bad_b)
    verdict = "chaotic (1 ulp moves the oracle itself)" if bad_b else ("GPU DIFFERS" if bad_g else "ok")
    print("%-6d | %-10s %-11.3e %-11.3e | %-10s %-11.3e %-11.3e | %s" % (sd, "same" if sb else "DIFFERENT", rb[0], rb[1], "same" if sg else "DIFFERENT", rg[0], rg[1], verdict))
print()
print("%d of %d seeds are chaotic by the 1-ulp criterion; the GPU differs from the oracle beyond the 1e-4 bar on %d of them, %d of which are NOT chaotic by this criterion"
      % (n_bad_b, len(seeds), n_bad_g, n_unexplained))
