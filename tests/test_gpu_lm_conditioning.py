"""The "up to conditioning" clause of the LM parity claim, checked by the driver-run suite.  north_star holds the GPU's pose / point updates to 1e-4 of
the update against the oracle (tests/lm_tolerances.py UPDATE_REL).  Rounds 4-5's randomised sweeps (tools/sweep_lm.py, builder-run) found 5 of 1 200 problems on which ONE
weakly constrained landmark leaves that bar (1.1 - 1.9e-4) -- each inside the band the ORACLE ITSELF spans when its float32 inputs move by one ulp.  Here: those five problems
(profiles/r05_sweeps.txt, seeds 509 and 611, regenerated from the sweep's own draws) and fifty fresh draws of the same generator.  Every problem: identical LM schedule and
outlier table; then EITHER the updates are within UPDATE_REL, OR the offending entry is named and its excursion is at most the oracle's own one-ulp band, measured right here
(twelve perturbed oracle runs: six uniform one-ulp shifts, six with a random sign per entry) -- and the number of such banded problems is bounded by lm_tolerances.CONDITIONING_BANDED_MAX.  Nothing outside the bar AND outside the band passes."""
import numpy as np
import pytest

from eao_fusion_amd import synth
from lm_tolerances import CONDITIONING_BANDED_MAX, UPDATE_REL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    return E

# (mode: 0 = LocalBundleAdjustment, 1 = BundleAdjustment, 2 = BundleAdjustment with map planes; the sweep's excursions as its generator drew them)
KNOWN = [
    (0, dict(n_free=104, n_fixed=2, n_points=6032, seed=1071982673, mono_frac=0.0, outlier_frac=0.0, band=7), None, None),
    (0, dict(n_free=30, n_fixed=3, n_points=1500, seed=893619570, mono_frac=0.3, outlier_frac=0.05), None, None),
    (2, dict(n_free=4, n_fixed=2, n_points=276, seed=404533642, mono_frac=0.3, outlier_frac=0.05), (6, 220420340), False),
    (0, dict(n_free=32, n_fixed=3, n_points=1376, seed=261914007, mono_frac=0.3, outlier_frac=0.05), None, None),
    (0, dict(n_free=27, n_fixed=1, n_points=1161, seed=925472987, mono_frac=0.3, outlier_frac=0.0), None, None),
]


def fresh_draws(seed, count):
    """tools/sweep_lm.py's generator (sizes capped so that fifty oracle runs fit the suite)"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        nf = int(rng.integers(2, 60)); nx = int(rng.integers(1, 4)); npts = int(nf * rng.integers(40, 100))
        kw = dict(n_free=nf, n_fixed=nx, n_points=npts, seed=int(rng.integers(0, 1 << 30)), mono_frac=float(rng.choice([0.0, 0.3, 1.0])), outlier_frac=float(rng.choice([0.0, 0.05])))
        if nf > 34 and rng.random() < 0.4:
            kw.update(n_free=int(rng.integers(35, 120)), band=int(rng.integers(3, 12)), mono_frac=0.0, outlier_frac=0.0)
            kw["n_points"] = kw["n_free"] * int(rng.integers(30, 50))
        mode = int(rng.integers(0, 3))
        if "band" in kw:
            mode = int(rng.integers(0, 2))
        planes = (int(rng.integers(1, 7)), int(rng.integers(0, 1 << 30))) if mode == 2 else None
        rob = bool(rng.integers(0, 2)) if mode else None
        out.append((mode, kw, planes, rob))
    return out


def rel(a, b, old):
    """largest |a - b| relative to the largest update, its flat index"""
    upd = max(np.abs(b.astype(np.float64) - old.astype(np.float64)).max(), 1e-6)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return float(d.max() / upd), int(d.argmax())


def within_bar(a, b, old):
    upd = max(np.abs(b.astype(np.float64) - old.astype(np.float64)).max(), 1e-6)
    ulp = np.spacing(np.abs(b).max().astype(np.float32))
    return np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= UPDATE_REL * upd + 2 * ulp


def test_updates_within_the_bar_or_within_the_oracles_own_one_ulp_band(gpu, oracle, capsys):
    banded, log = 0, []
    problems = [("sweep excursion %d" % i, c) for i, c in enumerate(KNOWN)] + [("fresh draw %d" % i, c) for i, c in enumerate(fresh_draws(2026, 50))]
    for name, (mode, kw, planes, rob) in problems:
        p = synth.synth_ba(**kw)
        if mode == 2:
            p = synth.add_ba_planes(p, n_planes=planes[0], seed=planes[1])
        run_g = (lambda q: gpu.Optimizer.LocalBundleAdjustment(q)) if mode == 0 else (lambda q: gpu.Optimizer.BundleAdjustment(q, 8, bRobust=rob))
        run_o = (lambda q: oracle.local_ba(q)) if mode == 0 else (lambda q: oracle.bundle_adjustment(q, 8, rob))
        r, o = run_g(p), run_o(p)
        if mode == 0:
            assert list(r["iters"]) == list(o["iters"]) and np.array_equal(r["edge_outlier"], o["edge_outlier"]), "%s %s: LM schedule / outlier table differ" % (name, kw)
        else:
            assert int(r["iters"][0]) == int(o["iters"][0]), "%s %s: LM schedule differs" % (name, kw)
        keys = ("poses", "points") + (("planes",) if mode == 2 else ())
        if all(within_bar(r[k], o[k], p[k]) for k in keys):
            continue
        # outside the bar: how far does the ORACLE move when its float32 inputs move by one ulp?  Six uniform shifts (points / observations / poses, either way: the
        # round-5 sweep's measure) and six draws in which every entry of the three arrays moves one ulp up or down at random (a uniform shift of all entries is a
        # correlated perturbation that largely cancels in a two-view landmark; rounding noise is not correlated)
        band = {k: 0.0 for k in keys}
        trials = [(arr, towards) for arr in ("points", "obs", "poses") for towards in (np.inf, -np.inf)] + [("random", s) for s in range(6)]
        for arr, towards in trials:
            q = dict(p)
            if arr != "random":
                q[arr] = np.nextafter(p[arr], np.float32(towards)).astype(np.float32)
            else:
                rs = np.random.default_rng(9000 + towards)
                for a2 in ("points", "obs", "poses"):
                    up = rs.integers(0, 2, size=p[a2].shape).astype(bool)
                    q[a2] = np.where(up, np.nextafter(p[a2], np.float32(np.inf)), np.nextafter(p[a2], np.float32(-np.inf))).astype(np.float32)
                q["obs"][:, 2] = np.where(p["obs"][:, 2] < 0, p["obs"][:, 2], q["obs"][:, 2])      # (ur < 0 marks a monocular observation: not a number to perturb)
                q["poses"][:, 3, :] = p["poses"][:, 3, :]
            o2 = run_o(q)
            for k in keys:
                band[k] = max(band[k], rel(o2[k], o[k], p[k])[0])
        for k in keys:
            if within_bar(r[k], o[k], p[k]):
                continue
            exc, at = rel(r[k], o[k], p[k])
            what = "%s[%d] (entry %d of %s %d)" % (k, at, at % (3 if k == "points" else 16 if k == "poses" else 4), {"points": "landmark", "poses": "camera", "planes": "plane"}[k],
                                                    at // (3 if k == "points" else 16 if k == "poses" else 4))
            log.append("%s %s: %s is %.2e of the largest update from the oracle; the oracle's own one-ulp band there: %.2e" % (name, kw, what, exc, band[k]))
            assert exc <= band[k], log[-1] + " -- OUTSIDE the band"
        banded += 1
    with capsys.disabled():
        print("\n[conditioning] %d problems, %d outside the %.0e bar but inside the oracle's one-ulp band (at most %d allowed)" % (len(problems), banded, UPDATE_REL, CONDITIONING_BANDED_MAX))
        for ln in log:
            print("   " + ln)
    assert banded <= CONDITIONING_BANDED_MAX, log
