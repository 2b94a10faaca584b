import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
fr = np.stack([synth.synth_frame(1000 + f, 640, 480) for f in range(64)])
ext = E.ORBextractor(1000, 1.2, 8, 20, 7)
sl = ext.stream_create(640, 480, 64, 3)
for s in range(3): sl[s]["frames"][:] = fr
for k in range(12 + 2):
    if k < 12: ext.stream_submit(k % 3)
    if k >= 2: ext.stream_wait((k - 2) % 3)
