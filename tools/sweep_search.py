"""Randomised bit-exactness sweep of the guided searches (SearchByProjection sim3 / keyframe / frame-to-frame / points, SearchByBoW,
SearchForTriangulation, SearchForInitialization, Fuse, SearchBySim3) against the CPU oracle over scene sizes, descriptor noise,
clutter, monocular fractions, vocabulary sizes, radii and ratio thresholds.  Not part of the test suite: run by hand on a GPU box.
    python tools/sweep_search.py [seed] [scenes] [handles]      (handles: the same calls through keyframe handles, eao_kf_*, round 5)"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import search, synth
from oracle import oracle as O
import test_gpu_search as T
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25
g, o = (search.product_handles() if len(sys.argv) > 3 and sys.argv[3] == "handles" else search.product()), O.search_binding()
bad = calls = 0
def same(name, a, b, kw):
    global bad, calls
    calls += 1
    ok = a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
    if not ok:
        bad += 1
        print("MISMATCH %s %s: %s vs %s" % (name, kw, a[0], b[0]), flush=True)
for it in range(N):
    kw = dict(n=int(rng.choice([rng.integers(30, 200), rng.integers(200, 900), rng.integers(900, 2500)])), seed=int(rng.integers(0, 1 << 30)),
              flip=float(rng.choice([0.0, 0.03, 0.08, 0.15])), clutter=float(rng.choice([0.0, 0.15, 0.5])), mono_frac=float(rng.choice([0.0, 0.3, 1.0])),
              n_nodes=int(rng.choice([1, 12, 60, 400])))
    try:
        sc = synth.synth_search_scene(**kw)
        P = sc["points"]
        kf = dict(sc["K2"]); kf["occupied"] = (rng.random(len(kf["kp_x"])) < rng.choice([0.0, 0.1, 0.5])).astype(np.uint8)
        th = int(rng.choice([3, 4, 10, 15]))
        same("projection_sim3", g.search_by_projection_sim3(kf, sc["Scw"], sc["K"], P, th), o.search_by_projection_sim3(kf, sc["Scw"], sc["K"], P, th), kw)
        ang = rng.uniform(0, 360, len(P["active"])).astype(np.float32)
        chk, od = bool(rng.integers(0, 2)), int(rng.choice([50, 64, 100]))
        same("projection_kf", g.search_by_projection_kf(kf, sc["T2w"], sc["K"], P, ang, th, od, chk), o.search_by_projection_kf(kf, sc["T2w"], sc["K"], P, ang, th, od, chk), kw)
        s1, s2 = T._sides(sc)
        for mode in (0, 1):
            ratio = float(rng.choice([0.6, 0.7, 0.9])); chk = bool(rng.integers(0, 2))
            same("bow%d" % mode, g.search_by_bow(mode, s1, s2, ratio, chk), o.search_by_bow(mode, s1, s2, ratio, chk), kw)
        k1, k2 = dict(sc["K1"]), dict(sc["K2"])
        k1["occupied"] = ((sc["mp1"] >= 0) & (rng.random(len(sc["mp1"])) < 0.5)).astype(np.uint8)
        k2["occupied"] = ((sc["mp2"] >= 0) & (rng.random(len(sc["mp2"])) < 0.3)).astype(np.uint8)
        for only_stereo in (0, 1):
            args = (k1, sc["fv1"], k2, sc["fv2"], sc["F12"], sc["ex"], sc["ey"], only_stereo, bool(rng.integers(0, 2)))
            same("triangulation", g.search_for_triangulation(*args), o.search_for_triangulation(*args), kw)
        f1, f2 = sc["K1"], sc["K2"]
        pm = np.stack([f1["kp_x"], f1["kp_y"]], 1)
        w, ratio, chk = int(rng.choice([20, 40, 100])), float(rng.choice([0.8, 0.9])), bool(rng.integers(0, 2))
        same("initialization", g.search_for_initialization(f1, f2, pm, w, ratio, chk), o.search_for_initialization(f1, f2, pm, w, ratio, chk), kw)
        Tm = sc["T2w"].astype(np.float64)
        pose = np.concatenate([Tm[:3, :3].ravel(), Tm[:3, 3], -Tm[:3, :3].T @ Tm[:3, 3]]).astype(np.float32)
        fth = float(rng.choice([2.5, 3.0, 4.0]))
        same("fuse", g.fuse_search(sc["K2"], 0, pose, sc["K"], sc["bf"], P, fth), o.fuse_search(sc["K2"], 0, pose, sc["K"], sc["bf"], P, fth), kw)
        same("fuse_sim3", g.fuse_search(sc["K2"], 1, sc["Scw"], sc["K"], sc["bf"], P, fth), o.fuse_search(sc["K2"], 1, sc["Scw"], sc["K"], sc["bf"], P, fth), kw)
        P1, P2 = T._pts_of(sc, sc["mp1"]), T._pts_of(sc, sc["mp2"])
        args = (sc["K1"], sc["T1w"], P1, sc["K2"], sc["T2w"], P2, sc["K"], 1.0, sc["R12"], sc["t12"], float(rng.choice([5.0, 7.5, 10.0])))
        same("sim3", g.search_by_sim3(*args), o.search_by_sim3(*args), kw)
    except Exception as e:
        bad += 1
        print("EXCEPTION %s: %s: %s" % (kw, type(e).__name__, e), flush=True)
print("search sweep: %d scenes, %d calls, %d mismatches" % (N, calls, bad))
