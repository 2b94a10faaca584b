"""Times the guided searches through host arrays, through keyframe handles and on the oracle (one CPU thread): python3 tools/dbg_search_handles.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench  # noqa: E402
from eao_fusion_amd import search as SR, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

g, gH, o = SR.product(), SR.product_handles(), O.search_binding()
sc, pose15, cases = bench.search_cases(synth)
out = {}
for name, fn in cases:
    a, ra = bench.time_calls(lambda: fn(g))
    h, rh = bench.time_calls(lambda: fn(gH))
    c, rc = bench.time_calls(lambda: fn(o), reps=8)
    assert ra[0] == rc[0] == rh[0] and np.array_equal(ra[1], rc[1]) and np.array_equal(rh[1], rc[1]), name
    out[name] = {"arrays_ms": round(a, 4), "handles_ms": round(h, 4), "cpu_ms": round(c, 4), "handles_over_cpu": round(h / c, 2)}
h1, h2 = gH.handle(sc["K1"], sc["fv1"]), gH.handle(sc["K2"], sc["fv2"])
nb = 10
t, _ = bench.time_calls(lambda: gH.search_for_triangulation_h(h1, [h2] * nb, [sc["F12"]] * nb, [sc["ex"]] * nb, [sc["ey"]] * nb, 0, True))
out["triangulation_batch10_handles_ms"] = round(t, 4)
t, _ = bench.time_calls(lambda: gH.fuse_search_h([h2] * nb, 0, [pose15] * nb, sc["K"], sc["bf"], sc["points"], 3.0))
out["fuse_batch10_handles_ms"] = round(t, 4)
print(json.dumps(out, indent=1))
