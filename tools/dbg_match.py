import sys, time; sys.path.insert(0, '.')
import torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
curf, lastf, mpsf = synth.synth_tracking()
mt = E.ORBmatcher(0.8, True)
for _ in range(3):
    mt.SearchByProjectionPoints(curf, mpsf, 1.0); mt.SearchByProjectionFrames(curf, lastf, 7.0, False)
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    n1, _m = mt.SearchByProjectionPoints(curf, mpsf, 1.0)
    n2, _m = mt.SearchByProjectionFrames(curf, lastf, 7.0, False)
    ts.append((time.perf_counter() - t0) * 1e3)
print('both searches ms min/med: %.3f %.3f' % (min(ts), sorted(ts)[10]), n1, n2)
