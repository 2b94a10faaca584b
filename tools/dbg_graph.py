import sys, faulthandler; faulthandler.enable(); sys.path.insert(0,'.')
import os
if os.environ.get("WITH_TORCH","1")=="1":
    import torch
    print("torch hip", torch.version.hip)
import numpy as np
import eao_fusion_amd as E
from eao_fusion_amd import synth
imgs = synth.synth_frames(8)
ext = E.ORBextractor(1000,1.2,8,20,7)
for i in range(3):
    k,d = ext.extract_batch(imgs)
    print(i, [len(x) for x in k][:4])
