#!/bin/bash
# kernel timeline of one ORB batch with the stages serialised (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export EAO_DBG_STEPS=4 EAO_DBG_PROF=1
rm -rf gpurun_out/trace_orb
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_orb -o t -- python3 tools/dbg_lanes.py > gpurun_out/trace_orb.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_orb/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
# last batch
n = 0
for i in range(len(rows) - 1, -1, -1):
    if "k_orient" in rows[i]["Kernel_Name"]:
        n += 1
        if n == 2: start = i + 1; break
prev = None
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print("%-20s grid %6s x %5s x %3s  dur %8.2f us  gap %6.2f us" % (name[:20], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3))
    prev = e
PY
