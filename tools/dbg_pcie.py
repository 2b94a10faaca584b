"""PCIe-inclusive ORB rate: eao_orb_extract_batch with pageable host buffers in and out (never used as bench `value`)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
imgs = synth.synth_frames(64)
ext = E.ORBextractor(1000, 1.2, 8, 20, 7)
ext.extract_batch(imgs)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); k, d = ext.extract_batch(imgs); ts.append(time.perf_counter() - t0)
n = sum(len(x) for x in k)
print("host in/out: %.3f ms per 64-frame batch, %.1f M kpts/s (%d keypoints)" % (min(ts) * 1e3, n / min(ts) / 1e6, n))
