import sys; sys.path.insert(0,'.')
import torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
p=synth.synth_ba()
for i in range(3):
    r=E.Optimizer.LocalBundleAdjustment(p)
print(r['timing'], r['iters'])
