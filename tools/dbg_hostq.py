import sys, time, os; sys.path.insert(0,'.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import sequence, synth
dev=torch.device("cuda")
B=64
frames=np.stack([synth.synth_frame(1000+f) for f in range(B)])
d_img=torch.from_numpy(frames).to(dev)
seq=sequence.SequenceShard(B,640,480,dev)
for _ in range(5): seq.extract(d_img)
torch.cuda.synchronize()
K=200
t=time.perf_counter()
for _ in range(K): seq.extract(d_img)
tq=time.perf_counter()-t
torch.cuda.synchronize()
print("torch runtime: host enqueue %.4f ms/step, total %.4f ms/step" % (tq/K*1e3, (time.perf_counter()-t)/K*1e3))
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5): seq.extract(d_img)
    torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(K): seq.extract(d_img)
    tq=time.perf_counter()-t
    torch.cuda.synchronize()
    print("torch side stream: host enqueue %.4f ms/step, total %.4f ms/step" % (tq/K*1e3, (time.perf_counter()-t)/K*1e3))
