#!/bin/bash
# kernel timeline of one ORB batch with the stages serialised (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export EAO_DBG_STEPS=4 EAO_DBG_PROF=${EAO_DBG_PROF-1}
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
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print("%-20s grid %6s x %5s x %3s  start %8.2f  end %8.2f  dur %8.2f us  queue %s" % (name[:20], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "")))
PY
