import sys, time; sys.path.insert(0, '.')
import torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
p = synth.synth_pose()
for i in range(3): r = E.Optimizer.PoseOptimization(p)
ts = []
for i in range(10):
    t0 = time.perf_counter(); r = E.Optimizer.PoseOptimization(p); ts.append((time.perf_counter() - t0) * 1e3)
print({k: r[k] for k in r if k in ('timing', 'iters', 'n_inliers')}, 'wall ms min/med: %.3f %.3f' % (min(ts), sorted(ts)[len(ts) // 2]))
