"""Timing of eao_pose_optimization_batch against a loop of eao_pose_optimization (host wall clock, inputs on the host):
the call at the C-ABI with the argument records packed once, the same call through the Python mirror (which builds the
records per call), and single calls."""
import time
import numpy as np
import torch  # noqa: F401
import sys; sys.path.insert(0, "."); import eao_fusion_amd as E
from eao_fusion_amd import _lib, synth

L = _lib.load()
for nb in (1, 8, 32, 128, 256, 1024):
    probs = [synth.synth_pose(n=300, seed=7000 + k) for k in range(nb)]
    pk = E.Optimizer.pack_pose_batch(probs)
    for _ in range(3):
        _lib.check(L.eao_pose_optimization_batch(pk["P"], nb, pk["R"]))
    ts = []
    for _ in range(9):
        t = time.perf_counter(); _lib.check(L.eao_pose_optimization_batch(pk["P"], nb, pk["R"])); ts.append(time.perf_counter() - t)
    tc = float(np.median(ts))
    t = time.perf_counter(); reps = 3
    for _ in range(reps):
        outs = E.Optimizer.PoseOptimizationBatch(probs)
    tb = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for p in probs[:32]:
        E.Optimizer.PoseOptimization(p)
    t1 = (time.perf_counter() - t) / min(nb, 32)
    print("frames %4d: C-ABI %.3f ms (%.2f us/frame, %.0f frames/s)   through the Python mirror %.3f ms   single call %.3f ms/frame"
          % (nb, tc * 1e3, tc / nb * 1e6, nb / tc, tb * 1e3, t1 * 1e3), flush=True)
