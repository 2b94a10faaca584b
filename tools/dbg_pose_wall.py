import sys, time, ctypes as C; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
for n in (300, 1000):
    p = synth.synth_pose(n=n)
    for i in range(20): r = E.Optimizer.PoseOptimization(p)
    ts = []
    for i in range(200):
        t0 = time.perf_counter(); r = E.Optimizer.PoseOptimization(p); ts.append((time.perf_counter() - t0) * 1e3)
    print("n=%d python-mirror wall ms min %.4f med %.4f device %.4f" % (n, min(ts), sorted(ts)[100], r['timing']['device_ms']))
