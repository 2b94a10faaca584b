"""Quick numbers: ORB step (64 frames, device-resident), BA batch (25 windows), one BA window, tracked frame, pose."""
import sys, time, ctypes as C, subprocess; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import _lib, synth, sequence
dev = torch.device("cuda")
frames = np.stack([synth.synth_frame(1000 + f, 640, 480) for f in range(64)])
d_img = torch.from_numpy(frames).to(dev)
seq = sequence.SequenceShard(64, 640, 480, dev)
for _ in range(60): seq.extract(d_img)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): seq.extract(d_img)
torch.cuda.synchronize(); print("ORB step 64 frames: %.4f ms" % ((time.perf_counter() - t0) / 100 * 1e3))
L = _lib.load()
probs = [synth.synth_ba(seed=6000 + w) for w in range(25)]
pk = E.Optimizer.pack_batch(probs)
for _ in range(4): _lib.check(L.eao_local_ba_batch(pk["P"], 25, None, pk["R"]))
ts = []
for _ in range(15):
    t0 = time.perf_counter(); _lib.check(L.eao_local_ba_batch(pk["P"], 25, None, pk["R"])); ts.append(time.perf_counter() - t0)
print("BA batch 25 windows: median %.3f min %.3f ms" % (np.median(ts) * 1e3, min(ts) * 1e3))
p = synth.synth_ba()
for _ in range(3): E.Optimizer.LocalBundleAdjustment(p)
ts = []
for _ in range(15):
    t0 = time.perf_counter(); E.Optimizer.LocalBundleAdjustment(p); ts.append(time.perf_counter() - t0)
print("BA one window (python mirror): median %.3f ms" % (np.median(ts) * 1e3))
out = subprocess.run([sys.executable, "tools/dbg_track.py"], capture_output=True, text=True).stdout.strip().split("\n")[-1]
print(out)
