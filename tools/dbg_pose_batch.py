"""Where a call of eao_pose_optimization_batch spends its time: wall clock at the C-ABI vs the device span (upload + kernels,
HIP events) for several batch sizes."""
import ctypes as C
import time
import numpy as np
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import _lib, synth

L = _lib.load()
for nb in (1, 2, 8, 32, 64, 128, 256, 512):
    probs = [synth.synth_pose(n=300, seed=7000 + k) for k in range(nb)]
    pk = E.Optimizer.pack_pose_batch(probs)
    for _ in range(3):
        _lib.check(L.eao_pose_optimization_batch(pk["P"], nb, pk["R"]))
    ts, ds = [], []
    dm, li = C.c_float(), C.c_int32()
    for _ in range(11):
        t = time.perf_counter(); _lib.check(L.eao_pose_optimization_batch(pk["P"], nb, pk["R"])); ts.append(time.perf_counter() - t)
        L.eao_last_lm_timing(C.byref(dm), C.byref(li)); ds.append(dm.value)
    print("frames %4d: wall %.3f ms, device span (upload + kernels) %.3f ms, host outside it %.3f ms" % (nb, np.median(ts) * 1e3, np.median(ds), np.median(ts) * 1e3 - np.median(ds)), flush=True)
