"""Throughput of independent LocalBundleAdjustment windows issued from several host threads (one HIP stream each)."""
import sys, time, threading; sys.path.insert(0, '.')
import torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
p = synth.synth_ba()
E.Optimizer.LocalBundleAdjustment(p)
for nth in (1, 2, 4, 8, 16):
    reps = 20
    def work():
        for _ in range(reps): E.Optimizer.LocalBundleAdjustment(p)
    ths = [threading.Thread(target=work) for _ in range(nth)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("threads %2d: %.3f ms per window, %.0f windows/s" % (nth, dt * 1e3 / (nth * reps), nth * reps / dt))
