# wall time of eao_local_ba_batch at the C-ABI: the 25 windows of BASELINE configs[4] (arguments packed once)
import sys, time, os, ctypes as C; sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import _lib, synth
n = int(os.environ.get("EAO_DBG_WINDOWS", "25"))
probs = [synth.synth_ba(seed=6000 + w) for w in range(n)]
pk = E.Optimizer.pack_batch(probs)
L = _lib.load()
for _ in range(3): _lib.check(L.eao_local_ba_batch(pk["P"], n, None, pk["R"]))
ts = []
for _ in range(20):
    t0 = time.perf_counter(); _lib.check(L.eao_local_ba_batch(pk["P"], n, None, pk["R"])); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
dm = C.c_float(); li = C.c_int32(); L.eao_last_lm_timing(C.byref(dm), C.byref(li))
E_ = sum(len(p["edge_cam"]) for p in probs) / n
print("eao_local_ba_batch, %d windows: min %.3f median %.3f ms per call (device %.3f ms) = %.3f ms per window; %d linearisations -> %.3e residual blocks/s"
      % (n, ts.min(), np.median(ts), dm.value, np.median(ts) / n, li.value, E_ * li.value / (np.median(ts) * 1e-3)))
print("iters", [list(pk["R"][w].iters[:]) for w in range(min(n, 25))])
