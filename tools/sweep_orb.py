"""Randomised bit-exactness sweep of the GPU ORB extractor against the CPU oracle over image sizes, feature counts, scale
factors, level counts, thresholds and textures (not part of the test suite: run by hand on a GPU box)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for it in range(N):
    w = int(rng.integers(320, 1400)); h = int(rng.integers(200, min(w, 900) + 1))     # landscape: portrait divides by zero upstream
    nfeat = int(rng.choice([100, 300, 500, 1000, 1500, 2000, 3000]))
    sf = float(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0]))
    top = int(np.floor(np.log(min(w, h) / 90.0) / np.log(sf))) + 1                     # keep the top level above ~90 px
    nlev = int(rng.integers(2, max(3, min(8, top) + 1)))
    ini = int(rng.choice([10, 20, 30, 50])); mn = int(rng.choice([3, 5, 7, 9]))
    mn = min(mn, ini)
    n_rect = int(rng.choice([0, 5, 40, 400])); n_small = int(rng.choice([0, 100, 1000]))
    seed = int(rng.integers(0, 1 << 30))
    try:
        img = synth.synth_frame(seed, w, h, n_rect, n_small)
        if rng.random() < 0.15:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)          # pure noise: every pixel a corner candidate
        orc = O.OrbOracle(nfeat, sf, nlev, ini, mn)
        try:
            ok_cfg = True
            k0, d0 = orc.extract(img)
        except Exception as ex:
            ok_cfg = False; msg = repr(ex)
        try:
            ext = E.ORBextractor(nfeat, sf, nlev, ini, mn)
            k1, d1 = ext(img)
            gpu_ok = True
        except Exception as ex:
            gpu_ok = False; gmsg = repr(ex)
        if not ok_cfg or not gpu_ok:
            print("cfg", (w, h, nfeat, sf, nlev, ini, mn), "oracle ok" if ok_cfg else "oracle: " + msg[:80], "gpu ok" if gpu_ok else "gpu: " + gmsg[:120])
            continue
        same = len(k0) == len(k1) and np.array_equal(k0, k1) and np.array_equal(d0, d1)
        if not same:
            bad += 1
            print("MISMATCH", (w, h, nfeat, sf, nlev, ini, mn, n_rect, n_small, seed), len(k0), len(k1))
    except Exception as ex:
        print("error", (w, h, nfeat, sf, nlev, ini, mn), repr(ex)[:200])
print("sweep done:", N, "configs,", bad, "mismatches")
