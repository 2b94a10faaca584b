import sys, time; sys.path.insert(0, '.')
import torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
p = synth.synth_ba()
for i in range(3):
    r = E.Optimizer.LocalBundleAdjustment(p)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    t0 = time.perf_counter(); r = E.Optimizer.LocalBundleAdjustment(p); ts.append((time.perf_counter() - t0) * 1e3)
print(r['timing'], r['iters'], 'wall ms min/med: %.3f %.3f' % (min(ts), sorted(ts)[len(ts) // 2]))
