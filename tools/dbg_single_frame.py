"""Latency of one eao_orb_extract call (pageable host buffers in and out), min / median of 200 calls."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
ext = E.ORBextractor(1000, 1.2, 8, 20, 7)
img = synth.synth_frame(1234, 640, 480)
one = img[None]
for _ in range(5): k, d = ext.extract_batch(one)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); k, d = ext.extract_batch(one); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("eao_orb_extract (through the ctypes mirror): min %.4f median %.4f ms, %d keypoints" % (ts.min(), np.median(ts), len(k[0])))
