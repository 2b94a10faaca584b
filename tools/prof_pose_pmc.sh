#!/bin/bash
# SQ counters of the single-workgroup PoseOptimization kernel (1000 correspondences): instructions per wave and per pass.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; rm -rf $O/pose_pmc
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/pose_pmc -o p -- python3 tools/dbg_pose_waves.py 1000 > $O/pose_pmc.log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pose_pmc/p_counter_collection.csv")))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "k_pose_optimization" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
