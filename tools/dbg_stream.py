"""Where the time of the streaming host API goes: PCIe copies alone (pinned), the pipeline with / without a pause between submits."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
fr = np.stack([synth.synth_frame(1000 + f, 640, 480) for f in range(64)])
# 1. the link: pinned -> device and back, alone
hp = torch.from_numpy(fr).pin_memory(); dv = torch.empty_like(hp, device="cuda")
out_d = torch.empty(64 * 1032 * 60, dtype=torch.uint8, device="cuda"); out_h = torch.empty(64 * 1032 * 60, dtype=torch.uint8).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(3): dv.copy_(hp, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): dv.copy_(hp, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("H2D 19.66 MB pinned: %.3f ms = %.1f GB/s" % (dt * 1e3, fr.nbytes / dt / 1e9))
t0 = time.perf_counter()
for _ in range(20):
    with torch.cuda.stream(s1): dv.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2): out_h.copy_(out_d, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("H2D 19.66 MB + D2H 3.96 MB on two streams: %.3f ms" % (dt * 1e3))
ext = E.ORBextractor(1000, 1.2, 8, 20, 7)
sl = ext.stream_create(640, 480, 64, 3)
for s in range(3): sl[s]["frames"][:] = fr
for rep in range(2):
    for s in range(3): ext.stream_submit(s)
    for s in range(3): ext.stream_wait(s)
def pipeline(nb, pause):
    t0 = time.perf_counter()
    for k in range(nb + 2):
        if k < nb:
            if pause: time.sleep(pause)
            ext.stream_submit(k % 3)
        if k >= 2: ext.stream_wait((k - 2) % 3)
    return (time.perf_counter() - t0) / nb
for pause in (0, 0.0002, 0.0004):
    print("pipeline, pause %.1f ms before each submit: %.3f ms per 64 frames" % (pause * 1e3, pipeline(30, pause) * 1e3))
# one slot at a time (no overlap): upload + extraction + download in sequence
t0 = time.perf_counter()
for k in range(20):
    ext.stream_submit(0); ext.stream_wait(0)
print("one slot, submit + wait: %.3f ms per 64 frames" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for k in range(20):
    ext.stream_submit(k % 2)
    if k: ext.stream_wait((k - 1) % 2)
ext.stream_wait(19 % 2)
print("two slots: %.3f ms per 64 frames" % ((time.perf_counter() - t0) / 20 * 1e3))
# host-side cost of the calls in the three-slot pipeline
ts, tw = [], []
for k in range(32):
    if k < 30:
        t0 = time.perf_counter(); ext.stream_submit(k % 3); ts.append(time.perf_counter() - t0)
    if k >= 2:
        t0 = time.perf_counter(); ext.stream_wait((k - 2) % 3); tw.append(time.perf_counter() - t0)
print("host: submit median %.3f ms (max %.3f), wait median %.3f ms" % (np.median(ts) * 1e3, max(ts) * 1e3, np.median(tw) * 1e3))
