# wall time of eao_local_ba measured at the C-ABI (arguments prepared once; no Python wrapper work inside the loop)
import sys, time, ctypes as C; sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import _lib, synth
p = synth.synth_ba()
L = _lib.load()
cams = np.ascontiguousarray(p["poses"], np.float32); fixed = np.ascontiguousarray(p["fixed"], np.uint8)
pts = np.ascontiguousarray(p["points"], np.float32); ec = np.ascontiguousarray(p["edge_cam"], np.int32)
ep = np.ascontiguousarray(p["edge_point"], np.int32); obs = np.ascontiguousarray(p["obs"], np.float32)
inv = np.ascontiguousarray(p["inv_sigma2"], np.float32)
P = _lib.BAProblem(len(cams), len(pts), len(ec), _lib.ptr(cams), _lib.ptr(fixed), _lib.ptr(pts), _lib.ptr(ec), _lib.ptr(ep), _lib.ptr(obs),
                   _lib.ptr(inv), p["fx"], p["fy"], p["cx"], p["cy"], p["bf"], 5, 10)
co, po, ol = np.zeros_like(cams), np.zeros_like(pts), np.zeros(len(ec), np.uint8)
R = _lib.BAResult(); R.cam_Tcw, R.points, R.edge_outlier = _lib.ptr(co), _lib.ptr(po), _lib.ptr(ol)
for _ in range(5): _lib.check(L.eao_local_ba(C.byref(P), None, C.byref(R)))
ts = []
for _ in range(40):
    t0 = time.perf_counter(); L.eao_local_ba(C.byref(P), None, C.byref(R)); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
dm = C.c_float(); li = C.c_int32(); L.eao_last_lm_timing(C.byref(dm), C.byref(li))
print("eao_local_ba at the C-ABI: min %.3f  median %.3f ms  (device %.3f ms, iters %s)" % (ts.min(), np.median(ts), dm.value, list(R.iters)))
t0 = time.perf_counter()
for _ in range(20): E.Optimizer.LocalBundleAdjustment(p)
print("through the Python mirror: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
