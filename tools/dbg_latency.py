"""Single-frame latencies at the host API, as Tracking.cc would see them: ORBextractor::operator() on one 640x480 frame
(host image in, host keypoints / descriptors out), two extractors from two threads (stereo), and small batches."""
import sys, time, threading; sys.path.insert(0, '.')
import numpy as np, torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
imgs = synth.synth_frames(64)
ext = E.ORBextractor(1000, 1.2, 8, 20, 7)
for B in (1, 2, 4, 8, 16, 64):
    sub = imgs[:B]
    for _ in range(3): ext.extract_batch(sub)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); k, d = ext.extract_batch(sub); ts.append(time.perf_counter() - t0)
    print("batch %2d: min %.3f ms  median %.3f ms per call (%.3f ms per frame), stages %s" % (B, min(ts) * 1e3, sorted(ts)[15] * 1e3, min(ts) * 1e3 / B, ""), flush=True)
ext.set_profiling(True)
ext.extract_batch(imgs[:1]); ext.extract_batch(imgs[:1])
print("single-frame stage times (each kernel alone, HIP events):", {k: round(v, 4) for k, v in ext.last_timing().items()})
