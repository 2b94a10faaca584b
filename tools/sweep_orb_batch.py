"""Randomised sweep of the BATCH paths of the ORB extractor (fused pyramid up to 32 frames / chain beyond, 1024- / 256-thread
quad-trees, pinned / DMA result copies, the device API on a torch stream): every frame of a batch must come out bit-identical
to the CPU oracle's single-frame result.  Not part of the test suite: run by hand on a GPU box."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for it in range(N):
    w, h = [(640, 480), (752, 480), (1241, 376), (320, 240), (533, 401)][int(rng.integers(0, 5))]
    nfeat = int(rng.choice([300, 1000, 2000])); sf = float(rng.choice([1.2, 1.5])); nlev = (int(rng.choice([3, 5, 8])) if sf < 1.3 else int(rng.choice([3, 4]))) if min(w, h) > 300 else 3
    B = int(rng.choice([1, 2, 7, 31, 32, 33, 47, 48, 64, 80]))
    seeds = rng.integers(0, 1 << 30, B)
    imgs = np.stack([synth.synth_frame(int(sd), w, h, int(rng.choice([5, 40, 400])), int(rng.choice([0, 100, 1000]))) for sd in seeds])
    if B > 2:
        imgs[int(rng.integers(0, B))] = 90                      # a frame without corners
    ext = E.ORBextractor(nfeat, sf, nlev, 20, 7)
    orc = O.OrbOracle(nfeat, sf, nlev, 20, 7)
    try:
        kps, desc = ext.extract_batch(imgs)
    except E.EaoError as ex:      # (a top level smaller than one FAST cell: refused loudly, upstream divides by zero there)
        print("skipped   %dx%d sf %.2f levels %d: %s" % (w, h, sf, nlev, str(ex)[:60]), flush=True)
        continue
    # the device API on torch's stream, twice in a row (the second call reuses every scratch buffer)
    cap = ext.max_keypoints(w, h)
    d_img = torch.from_numpy(imgs).cuda()
    d_k = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(2):
        ext.extract_batch_device(d_img.data_ptr(), w, h, w, w * h, B, d_k.data_ptr(), d_d.data_ptr(), cap, d_n.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    hn = d_n.cpu().numpy(); hk = d_k.cpu().numpy(); hd = d_d.cpu().numpy()
    ok = True
    for f in sorted(set([0, B - 1] + list(rng.integers(0, B, 4)))):
        ok_k, ok_d = orc.extract(imgs[f])
        n = len(ok_k)
        good = np.array_equal(kps[f], ok_k) and np.array_equal(desc[f], ok_d) and hn[f] == n and np.array_equal(hk[f, :n].reshape(-1).view(E.KP_DTYPE), ok_k) and np.array_equal(hd[f, :n], ok_d)
        ok = ok and good
    bad += not ok
    print("%s  %dx%d nfeat %d sf %.2f levels %d batch %d" % ("ok      " if ok else "MISMATCH", w, h, nfeat, sf, nlev, B), flush=True)
print("sweep done: %d batches, %d mismatches" % (N, bad))
