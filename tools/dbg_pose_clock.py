"""Does a single-workgroup kernel run at another clock after an idle spell than right behind chip-filling work?  PoseOptimization (1000 correspondences) timed
after 0.5 s of sleep, right after 300 ORB steps of 64 frames, and interleaved with them."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth, sequence
dev = torch.device("cuda")
frames = np.stack([synth.synth_frame(1000 + f, 640, 480) for f in range(64)])
d_img = torch.from_numpy(frames).to(dev)
seq = sequence.SequenceShard(64, 640, 480, dev)
p = synth.synth_pose(n=1000)
def pose(k):
    ts = []
    for i in range(k):
        r = E.Optimizer.PoseOptimization(p); ts.append(r['timing']['device_ms'])
    return ts
pose(10)
for rep in range(2):
    time.sleep(0.5)
    a = pose(30)
    for _ in range(300): seq.extract(d_img)
    torch.cuda.synchronize()
    b = pose(30)
    c = []
    for i in range(30):
        for _ in range(5): seq.extract(d_img)
        torch.cuda.synchronize()
        c += pose(1)
    print("device ms: after idle first %.4f median %.4f | after 300 ORB steps first %.4f median %.4f | interleaved median %.4f" % (a[0], np.median(a), b[0], np.median(b), np.median(c)))
