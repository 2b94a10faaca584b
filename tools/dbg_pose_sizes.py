"""PoseOptimization at the C-ABI for several correspondence counts: device span and time per linearisation (the register kernels
hold two edges per thread up to 1024 correspondences, four up to 2048)."""
import sys, time; sys.path.insert(0, '.')
import ctypes as C
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth, _lib
L = _lib.load()
for n in (300, 714, 1000, 1100, 1500):
    pk = E.Optimizer.pack_pose_batch([synth.synth_pose(n=n, seed=4242)])
    ds = []
    dm, li = C.c_float(), C.c_int32()
    for _ in range(12):
        _lib.check(L.eao_pose_optimization(C.byref(pk["P"][0]), C.byref(pk["R"][0])))
        L.eao_last_lm_timing(C.byref(dm), C.byref(li)); ds.append(dm.value)
    print("n %4d: device %.3f ms, %d linearisations -> %.2f us per linearisation" % (n, np.median(ds[2:]), li.value, np.median(ds[2:]) * 1e3 / max(li.value, 1)))
