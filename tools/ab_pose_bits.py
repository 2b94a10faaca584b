"""(tools/ab_pose_job.sh runs the three steps below on the GPU box against gpurun_ab/libeaofusion_hip_head.so, a build of the previous commit.)
Bit-for-bit A/B of PoseOptimization between two builds of the library: dumps (pose, outlier table, LM trace) of a seeded set of problems over every launch
geometry to an .npz; run once per build (EAO_LIB_PATH selects it), then with both files to compare.
    EAO_LIB_PATH=gpurun_ab/libeaofusion_hip_head.so python tools/ab_pose_bits.py dump gpurun_out/pose_head.npz
    python tools/ab_pose_bits.py dump gpurun_out/pose_new.npz
    python tools/ab_pose_bits.py cmp gpurun_out/pose_head.npz gpurun_out/pose_new.npz"""
import sys; sys.path.insert(0, '.')
import numpy as np
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not (a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes())]
    print("%d arrays, %d differ bit for bit%s" % (len(a.files), len(bad), (": " + ", ".join(bad[:12])) if bad else ""))
    sys.exit(1 if bad else 0)
import torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
out = {}
rng = np.random.default_rng(77)
cases = [dict(n=n, seed=4000 + i) for i, n in enumerate((5, 40, 64, 200, 256, 300, 512, 700, 1000, 1024, 1500, 2048, 2500))]
for i in range(40):
    kw = dict(n=int(rng.integers(3, 2600)), seed=int(rng.integers(0, 1 << 30)), sigma=float(rng.choice([0.0, 0.5, 1.0, 2.0])), outlier_frac=float(rng.choice([0.0, 0.1, 0.3])),
              mono_frac=float(rng.choice([0.0, 0.3, 1.0])))
    if rng.random() < 0.3: kw["n_planes"] = int(rng.integers(1, 9))
    cases.append(kw)
for i, kw in enumerate(cases):
    r = E.Optimizer.PoseOptimization(synth.synth_pose(**kw))
    out["c%d_T" % i] = np.asarray(r["Tcw"]); out["c%d_o" % i] = np.asarray(r["outlier"]); out["c%d_n" % i] = np.asarray([r["n_inliers"]])
    for k, v in r["trace"].items(): out["c%d_t_%s" % (i, k)] = np.asarray(v)
    if "plane_outlier" in r: out["c%d_po" % i] = np.asarray(r["plane_outlier"])
# the batch entry point (one workgroup per frame)
ps = [synth.synth_pose(n=300, seed=9000 + k) for k in range(16)]
for k, r in enumerate(E.Optimizer.PoseOptimizationBatch(ps)):
    out["b%d_T" % k] = np.asarray(r["Tcw"]); out["b%d_o" % k] = np.asarray(r["outlier"])
np.savez(sys.argv[2], **out)
print("dumped %d arrays to %s" % (len(out), sys.argv[2]))
