"""LocalBundleAdjustment looping on one thread while a map-scale BundleAdjustment loops on another: is every LBA result bit-identical?  EAO_BA_ND as set by the caller."""
import sys, threading, time; sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
p = synth.synth_ba()
nkf = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mode = sys.argv[3] if len(sys.argv) > 3 else "gba"      # the neighbour: gba | orb | batch | lba (a second LBA loop on another window)
ref = E.Optimizer.LocalBundleAdjustment(p)
stop = False
gdiff = []
if mode == "gba":
    g = synth.synth_ba(n_free=nkf, n_fixed=1, n_points=50 * nkf, seed=5400, band=11)
    gref = E.Optimizer.BundleAdjustment(g, 10, bRobust=False)
elif mode == "batch":
    wins = [synth.synth_ba(seed=6000 + w) for w in range(25)]
    bref = E.Optimizer.LocalBundleAdjustmentBatch(wins)
elif mode == "lba":
    p2 = synth.synth_ba(seed=6007)
    l2 = E.Optimizer.LocalBundleAdjustment(p2)
def gloop():
    k = 0
    if mode == "orb":
        dev = torch.device("cuda", 0)
        from eao_fusion_amd import sequence
        frames = np.stack([synth.synth_frame(1000 + f) for f in range(64)])
        d_img = torch.from_numpy(frames).to(dev)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            seq = sequence.SequenceShard(64, 640, 480, dev)
            while not stop:
                seq.extract(d_img); k += 1
                if k % 8 == 0: st.synchronize()
            st.synchronize()
        print("ORB batches", k, flush=True)
        return
    while not stop:
        if mode == "gba":
            r = E.Optimizer.BundleAdjustment(g, 10, bRobust=False)
            if not (np.array_equal(r["poses"], gref["poses"]) and np.array_equal(r["points"], gref["points"])):
                gdiff.append((k, float(np.abs(r["points"] - gref["points"]).max())))
        elif mode == "batch":
            rb = E.Optimizer.LocalBundleAdjustmentBatch(wins)
            if not all(np.array_equal(rb[w]["points"], bref[w]["points"]) and np.array_equal(rb[w]["poses"], bref[w]["poses"]) for w in range(25)):
                gdiff.append((k, [w for w in range(25) if not np.array_equal(rb[w]["points"], bref[w]["points"])]))
        else:
            r2 = E.Optimizer.LocalBundleAdjustment(p2)
            if not np.array_equal(r2["points"], l2["points"]):
                gdiff.append((k, float(np.abs(r2["points"] - l2["points"]).max())))
        k += 1
    print("neighbour (%s) calls" % mode, k, "differing", len(gdiff), gdiff[:5], flush=True)
t = threading.Thread(target=gloop); t.start()
bad = []
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 12:
    r = E.Optimizer.LocalBundleAdjustment(p)
    n += 1
    if not (np.array_equal(r["poses"], ref["poses"]) and np.array_equal(r["points"], ref["points"]) and np.array_equal(r["edge_outlier"], ref["edge_outlier"])):
        dp = np.abs(r["points"].astype(np.float64) - ref["points"]); dc = np.abs(r["poses"].astype(np.float64) - ref["poses"])
        bad.append((n, int((dp > 0).sum()), float(dp.max()), int((dc > 0).sum()), float(dc.max()), list(r["iters"]), list(ref["iters"]), int((r["edge_outlier"] != ref["edge_outlier"]).sum())))
stop = True; t.join()
print("LBA calls", n, "differing", len(bad))
for b in bad[:10]:
    print("   call %d: %d point entries differ (max %.3e), %d pose entries (max %.3e), iters %s vs %s, %d outlier flags" % b)
