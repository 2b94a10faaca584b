"""50 SearchByBoW(KF, Frame) calls through keyframe handles (for a kernel trace: tools/trace_py.sh tools/dbg_bow_trace.py <tag>)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench  # noqa: E402
from eao_fusion_amd import search as SR, synth  # noqa: E402

gH = SR.product_handles()
sc, pose15, cases = bench.search_cases(synth)
h1, h2 = gH.handle(sc["K1"], sc["fv1"]), gH.handle(sc["K2"], sc["fv2"])
v1 = (sc["mp1"] >= 0).astype(np.uint8)
for _ in range(5):
    gH.search_by_bow_h(0, h1, v1, h2, None, 0.75, True)
ts = []
for _ in range(50):
    t0 = time.perf_counter()
    gH.search_by_bow_h(0, h1, v1, h2, None, 0.75, True)
    ts.append(time.perf_counter() - t0)
print("bow kf-frame through handles: median %.1f us, min %.1f us" % (np.median(ts) * 1e6, min(ts) * 1e6))
import ctypes as C
out, n = np.full(h1.n, -1, np.int32), C.c_int32(0)
L = gH.lib
ts = []
for _ in range(50):
    t0 = time.perf_counter()
    L.eao_kf_search_by_bow(0, h1.h, v1.ctypes.data, h2.h, None, 0.75, 1, out.ctypes.data, C.byref(n))
    ts.append(time.perf_counter() - t0)
print("  the C call alone: median %.1f us, min %.1f us" % (np.median(ts) * 1e6, min(ts) * 1e6))
