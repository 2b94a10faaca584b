"""Randomised parity sweep of LocalBundleAdjustment / BundleAdjustment (with and without map planes) against the CPU oracle over
window sizes -- register-tile solver up to 30 free keyframes, the map-scale path beyond (every padding case of its 32-column
panels and 64 x 64 tiles) -- observation mixes and seeds.  Not part of the test suite: run by hand on a GPU box."""
import os, sys; sys.path.insert(0, '.')
import numpy as np, torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
REL = 1e-4
def close(a, b, old):
    upd = max(np.abs(b.astype(np.float64) - old.astype(np.float64)).max(), 1e-6)
    ulp = np.spacing(np.abs(b).max().astype(np.float32))
    return np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= REL * upd + 2 * ulp
bad = 0
for it in range(N):
    nf = int(rng.integers(2, 100)); nx = int(rng.integers(1, 4)); npts = int(nf * rng.integers(40, 140))
    kw = dict(n_free=nf, n_fixed=nx, n_points=npts, seed=int(rng.integers(0, 1 << 30)), mono_frac=float(rng.choice([0.0, 0.3, 1.0])),
              outlier_frac=float(rng.choice([0.0, 0.05])))
    if nf > 34 and rng.random() < 0.4:      # (round 5) a trajectory map: every keyframe covisible with its neighbours only -- block-sparse tiles, symbolic fill-in; up to 160 keyframes
        kw.update(n_free=int(rng.integers(35, 160)), band=int(rng.integers(3, 12)), mono_frac=0.0, outlier_frac=0.0)
        kw["n_points"] = kw["n_free"] * int(rng.integers(30, 60))
    p = synth.synth_ba(**kw)
    mode = int(rng.integers(0, 3))
    if "band" in kw: mode = int(rng.integers(0, 2)); os.environ["EAO_BA_SETUP_THREADS"] = str(int(rng.choice([1, 3, 8])))      # (the set-up on 1 / 3 / 8 workers of the host crew)
    else: os.environ.pop("EAO_BA_SETUP_THREADS", None)
    # (round 6) the map-scale path's elimination order: the library's choice, natural order, or a forced number of segments (read per call)
    if kw["n_free"] > 30: os.environ["EAO_BA_ND"] = str(int(rng.choice([0, 0, 1, 2, 3, 5, 8]))); kw_nd = os.environ["EAO_BA_ND"]
    else: os.environ.pop("EAO_BA_ND", None); kw_nd = "-"
    try:
        if mode == 0:
            r, o = E.Optimizer.LocalBundleAdjustment(p), O.local_ba(p)
            ok = list(r["iters"]) == list(o["iters"]) and np.array_equal(r["edge_outlier"], o["edge_outlier"])
        else:
            if mode == 2:
                p = synth.add_ba_planes(p, n_planes=int(rng.integers(1, 7)), seed=int(rng.integers(0, 1 << 30)))
            rob = bool(rng.integers(0, 2))
            r, o = E.Optimizer.BundleAdjustment(p, 8, bRobust=rob), O.bundle_adjustment(p, 8, rob)
            ok = int(r["iters"][0]) == int(o["iters"][0])
            if mode == 2:
                ok = ok and close(r["planes"], o["planes"], p["planes"])
        ok = ok and close(r["poses"], o["poses"], p["poses"]) and close(r["points"], o["points"], p["points"])
    except Exception as ex:  # noqa: BLE001
        ok = False; print("   exception", repr(ex))
    if not ok:
        bad += 1
        print("MISMATCH mode %d %s EAO_BA_ND=%s" % (mode, kw, kw_nd), flush=True)
        try:      # what differs, and how far the ORACLE itself moves when the input points move by one float32 ulp (an ill-conditioned landmark amplifies the last bit)
            def rel(a, b, old):
                upd = max(np.abs(b.astype(np.float64) - old.astype(np.float64)).max(), 1e-6)
                return np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / upd
            keys = ("poses", "points") + (("planes",) if mode == 2 else ())
            band = {k: 0.0 for k in keys}
            for name, towards in (("points", np.inf), ("points", -np.inf), ("obs", np.inf), ("obs", -np.inf), ("poses", np.inf), ("poses", -np.inf)):
                q = dict(p); q[name] = np.nextafter(p[name], np.float32(towards)).astype(np.float32)
                o2 = O.local_ba(q) if mode == 0 else O.bundle_adjustment(q, 8, rob)
                for k in keys: band[k] = max(band[k], rel(o2[k], o[k], p[k]))
            print("   iters %s / %s; |gpu - oracle| / update: %s; the ORACLE under one float32 ulp of input noise (points, observations, poses, either way): %s" % (
                  list(r["iters"]), list(o["iters"]), ", ".join("%s %.2e" % (k, rel(r[k], o[k], p[k])) for k in keys), ", ".join("%s %.2e" % (k, band[k]) for k in keys)), flush=True)
        except Exception as ex:  # noqa: BLE001
            print("   (details failed: %r)" % ex)
print("sweep done: %d problems, %d mismatches" % (N, bad))
