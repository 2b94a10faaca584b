"""k_pose_optimization (launch record by value) against k_pose_optimization_batch (record read through a pointer) on the same frames: run under
    rocprofv3 --kernel-trace --stats -- python3 tools/dbg_pose_value_vs_pointer.py
and compare the two kernels' average durations."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch  # noqa: F401
import eao_fusion_amd as E
from eao_fusion_amd import synth
for n in (300, 700, 1000):
    p = synth.synth_pose(n=n)
    for i in range(25):
        a = E.Optimizer.PoseOptimization(p)
        b = E.Optimizer.PoseOptimizationBatch([p])[0]
    assert np.array_equal(a["Tcw"], b["Tcw"]) and np.array_equal(a["outlier"], b["outlier"])
print("done")
