"""Two 64-frame ORB batches in flight: two extractor handles on two streams, steps issued alternately -- does the tail of one
step (orientation + description alone on the chip, the cross-stream join in front of it) overlap the head of the next?"""
import os, sys, time
sys.path.insert(0, ".")
import torch
import numpy as np
import eao_fusion_amd as E
from eao_fusion_amd import synth, sequence as SQ
B = int(os.environ.get("EAO_DBG_BATCH", "64"))
fr = [torch.from_numpy(np.stack([synth.synth_frame(1000 + 64 * k + f, 640, 480) for f in range(B)])).cuda() for k in range(2)]
sh = [SQ.SequenceShard(B), SQ.SequenceShard(B)]
st = [torch.cuda.Stream(), torch.cuda.Stream()]
K = int(os.environ.get("EAO_DBG_STEPS", "100"))
def run(two):
    for _ in range(10):
        for k in range(2):
            with torch.cuda.stream(st[k if two else 0]):
                sh[k].extract(fr[k])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(K):
            k = i & 1
            with torch.cuda.stream(st[k if two else 0]):
                sh[k].extract(fr[k])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    return best
a = run(False); b = run(True)
print("batch %d, two handles: one stream %.4f ms per step, two streams %.4f ms per step (%.3e kpts/s)" % (B, a, b, B * 1005.5 / (b * 1e-3)))
