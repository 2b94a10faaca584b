"""Timing of eao_pose_optimization_batch against a loop of eao_pose_optimization (host wall clock, inputs on the host)."""
import time
import numpy as np
import eao_fusion_amd as E
from eao_fusion_amd import synth

for nb in (1, 8, 32, 128, 256, 1024):
    probs = [synth.synth_pose(n=300, seed=7000 + k) for k in range(nb)]
    E.Optimizer.PoseOptimizationBatch(probs)
    t = time.perf_counter(); reps = 5
    for _ in range(reps):
        outs = E.Optimizer.PoseOptimizationBatch(probs)
    tb = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for p in probs[:32]:
        E.Optimizer.PoseOptimization(p)
    ts = (time.perf_counter() - t) / min(nb, 32)
    print("frames %4d: batch %.3f ms (%.1f us/frame, %.0f frames/s)   single call %.3f ms/frame" % (nb, tb * 1e3, tb / nb * 1e6, nb / tb, ts * 1e3), flush=True)
