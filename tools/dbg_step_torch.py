"""Device-resident ORB step under PyTorch's bundled HIP runtime (what bench.py runs on): ms per 64-frame step and host enqueue time."""
import os, sys, time
sys.path.insert(0, ".")
import torch
import numpy as np
import eao_fusion_amd as E
from eao_fusion_amd import synth, sequence as SQ
B = int(os.environ.get("EAO_DBG_BATCH", "64"))
fr = torch.from_numpy(np.stack([synth.synth_frame(1000 + f, 640, 480) for f in range(B)])).cuda()
sh = SQ.SequenceShard(B)
for _ in range(5):
    sh.extract(fr)
torch.cuda.synchronize()
K = int(os.environ.get("EAO_DBG_STEPS", "100"))
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        sh.extract(fr)
    te = time.perf_counter() - t0
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / K * 1e3)
print("batch %d: step %.4f ms (host enqueue %.4f ms)" % (B, best, te / K * 1e3))
