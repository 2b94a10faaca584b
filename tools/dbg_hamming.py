# Hamming 1000 x 1000 x 64 pairs on device-resident descriptors: matrix mode (write-bound) and best-2 mode
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
L = E.load()
pairs = 64
g = torch.Generator().manual_seed(2000)
dA = torch.randint(0, 256, (pairs, 1000, 32), dtype=torch.uint8, generator=g).cuda()
dB = torch.randint(0, 256, (pairs, 1000, 32), dtype=torch.uint8, generator=g).cuda()
dD = torch.empty((pairs, 1000, 1000), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(): E._lib.check(L.eao_hamming_matrix_device(dA.data_ptr(), 1000, dB.data_ptr(), 1000, pairs, dD.data_ptr(), st))
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("matrix: %.4f ms per launch, %.0f GB/s (%.1f %% of 8 TB/s)" % (ms, pairs * 2.064e6 / ms / 1e6, pairs * 2.064e6 / ms / 1e6 / 80))
ref = ((dA[0, :8].cpu().numpy()[:, None, :] ^ dB[0].cpu().numpy()[None, :, :]))
ref = np.unpackbits(ref, axis=2).sum(2)
assert np.array_equal(ref, dD[0, :8].cpu().numpy()), "mismatch"

import ctypes as C
dO = torch.empty((pairs, 1000, 4), dtype=torch.int32, device="cuda")
def run2(): E._lib.check(L.eao_hamming_best2_device(dA.data_ptr(), 1000, dB.data_ptr(), 1000, pairs, None, dO.data_ptr(), st))
for _ in range(5): run2()
torch.cuda.synchronize()
e0.record()
for _ in range(50): run2()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("best2: %.4f ms per launch, %.0f G distances/s" % (ms, pairs * 1e6 / ms / 1e6))
