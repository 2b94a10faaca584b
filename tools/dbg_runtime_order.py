"""Which HIP runtime the process ends up with, and the ORB step time, when libeaofusion_hip.so (rpath /opt/rocm/lib) is loaded
BEFORE torch (EAO_LOAD_FIRST=1) or after it."""
import os, sys, time
sys.path.insert(0, ".")
first = os.environ.get("EAO_LOAD_FIRST") == "1"
if first:
    import eao_fusion_amd as E
    E.load()
import torch
import eao_fusion_amd as E
import numpy as np
from eao_fusion_amd import synth
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l or "librccl" in l})
print("load first:", first, libs)
x = torch.ones(1 << 20, device="cuda")
print("torch ok:", float(x.sum()), torch.version.hip)
fr = torch.from_numpy(np.stack([synth.synth_frame(1000 + f, 640, 480) for f in range(64)])).cuda()
from eao_fusion_amd import sequence as SQ
sh = SQ.SequenceShard(64)
for _ in range(5):
    sh.extract(fr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    sh.extract(fr)
torch.cuda.synchronize()
print("step ms: %.4f" % ((time.perf_counter() - t0) / 50 * 1e3))
a = torch.randn(2048, 2048, device="cuda"); b = a @ a; print("gemm ok", float(b[0, 0]))
