"""Randomised sweep of eao_local_ba_batch / eao_pose_optimization_batch against the single calls: random compositions of window
sizes (tile solver, beyond it, far-off starts whose trials get rejected, windows without edges) and batch sizes 1..40.  Every
window of a batch must equal its own eao_local_ba call within the parity bar (same LM schedule, updates <= 1e-4); every frame
of a pose batch must be BIT-identical to its single call.  Not part of the test suite: run by hand on a GPU box."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
REL = 1e-4
def close(a, b, old):
    upd = max(np.abs(b.astype(np.float64) - old.astype(np.float64)).max(), 1e-6)
    ulp = np.spacing(np.abs(b).max().astype(np.float32))
    return np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= REL * upd + 2 * ulp
bad = 0
for it in range(N):
    nb = int(rng.choice([1, 2, 3, 5, 8, 13, 25, 40]))
    probs = []
    for w in range(nb):
        kind = rng.random()
        sd = int(rng.integers(0, 1 << 30))
        if kind < 0.6:
            probs.append(synth.synth_ba(n_free=int(rng.integers(2, 24)), n_fixed=int(rng.integers(1, 4)), n_points=int(rng.integers(100, 1500)), seed=sd,
                                        mono_frac=float(rng.choice([0.0, 0.3]))))
        elif kind < 0.75:
            probs.append(synth.synth_ba(n_free=int(rng.integers(31, 40)), n_fixed=2, n_points=1200, seed=sd))        # beyond the tile solver
        elif kind < 0.9:
            probs.append(synth.synth_ba(n_free=5, n_fixed=2, n_points=200, seed=int(rng.choice([3030, 3032, 3033, 3035, 3036, 3037, 3038, 3041, 3043, 3044])),
                                        rot_noise_deg=25, trans_noise=0.8, point_noise=1.0, mono_frac=0.7))                # rejected trials
        else:
            probs.append(synth.synth_ba(n_free=20, n_fixed=4, n_points=3000, seed=sd))
    single = [E.Optimizer.LocalBundleAdjustment(p) for p in probs]
    res = E.Optimizer.LocalBundleAdjustmentBatch(probs)
    ok = True
    for w, (p, a, b) in enumerate(zip(probs, res, single)):
        good = list(a["iters"]) == list(b["iters"]) and close(a["poses"], b["poses"], p["poses"]) and close(a["points"], b["points"], p["points"]) \
            and np.array_equal(a["edge_outlier"], b["edge_outlier"])
        if not good:
            upd = max(np.abs(b["poses"].astype(np.float64) - p["poses"].astype(np.float64)).max(), 1e-6)
            print("   window %d (%d cams, %d points, %d edges): iters %s vs %s, pose diff / update %.3e, outlier tables equal %s" % (
                w, len(p["poses"]), len(p["points"]), len(p["edge_cam"]), list(a["iters"]), list(b["iters"]),
                np.abs(a["poses"].astype(np.float64) - b["poses"].astype(np.float64)).max() / upd, np.array_equal(a["edge_outlier"], b["edge_outlier"])), flush=True)
            dpt = np.abs(a["points"].astype(np.float64) - b["points"].astype(np.float64)).max(axis=1)
            updp = max(np.abs(b["points"].astype(np.float64) - p["points"].astype(np.float64)).max(), 1e-6)
            worst = int(np.argmax(dpt))
            nobs = int((p["edge_point"] == worst).sum())
            print("      points: worst diff / update %.3e at point %d (%d observations, |value| %.3f, its own update %.3e)" % (
                dpt[worst] / updp, worst, nobs, np.abs(b["points"][worst]).max(), np.abs(b["points"][worst].astype(np.float64) - p["points"][worst]).max()), flush=True)
            from oracle import oracle as O
            O.build()
            o = O.local_ba(p)
            band = 0.0
            for draw in range(3):        # the oracle's own sensitivity: every input moved by -1 / 0 / +1 float32 ulp
                r2 = np.random.default_rng(1000 + draw)
                q = dict(p)
                for key in ("points", "obs", "poses"):
                    v = p[key].copy()
                    dd = r2.integers(-1, 2, v.shape)
                    if key == "poses":
                        dd[:, 3, :] = 0
                    wv = np.where(dd > 0, np.nextafter(v, np.float32(np.inf)), np.where(dd < 0, np.nextafter(v, np.float32(-np.inf)), v)).astype(np.float32)
                    q[key] = np.where(p[key] < 0, p[key], wv) if key == "obs" else wv
                o2 = O.local_ba(q)
                band = max(band, np.abs(o2["points"].astype(np.float64) - o["points"].astype(np.float64)).max() / updp)
            print("      the oracle against itself with one-ulp input noise: %.3e" % band, flush=True)
            print("      against the fp64 oracle: batch %.3e, single call %.3e (points, relative to the update); iters oracle %s" % (
                np.abs(a["points"].astype(np.float64) - o["points"].astype(np.float64)).max() / updp,
                np.abs(b["points"].astype(np.float64) - o["points"].astype(np.float64)).max() / updp, list(o["iters"])), flush=True)
            # a window on which one ulp moves the oracle itself by more than the difference seen is ill-conditioned (typically a
            # two-observation point that travels metres), not a disagreement of the two entry points
            if list(a["iters"]) == list(b["iters"]) and np.array_equal(a["edge_outlier"], b["edge_outlier"]) and close(a["poses"], b["poses"], p["poses"]) \
                    and dpt[worst] / updp <= band:
                print("      -> inside the oracle's own one-ulp band: ill-conditioned window, not counted", flush=True)
                good = True
        ok = ok and good
    # pose batch
    npz = int(rng.choice([1, 3, 17, 64]))
    pp = [synth.synth_pose(n=int(rng.integers(3, 2300)), seed=int(rng.integers(0, 1 << 30)), mono_frac=float(rng.choice([0.0, 0.5, 1.0]))) for _ in range(npz)]
    ps = [E.Optimizer.PoseOptimization(p) for p in pp]
    pb = E.Optimizer.PoseOptimizationBatch(pp)
    for q, (a, b) in enumerate(zip(pb, ps)):
        good = a["n_inliers"] == b["n_inliers"] and np.array_equal(a["outlier"], b["outlier"]) and np.array_equal(a["Tcw"].view(np.uint32), b["Tcw"].view(np.uint32))
        if not good:
            print("   pose frame %d (n = %d): inliers %d vs %d" % (q, len(pp[q]["points"]), a["n_inliers"], b["n_inliers"]), flush=True)
        ok = ok and good
    bad += not ok
    print("%s  %d windows, %d pose frames" % ("ok      " if ok else "MISMATCH", nb, npz), flush=True)
print("sweep done: %d batches, %d mismatches" % (N, bad))
