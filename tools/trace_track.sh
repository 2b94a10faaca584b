#!/bin/bash
# kernel timeline of one tracked frame (tools/dbg_track.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tr_track
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_track -o t -- python3 tools/dbg_track.py > gpurun_out/tr_track.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/tr_track/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_pose_optimization" in r["Kernel_Name"]]      # the last kernel of a tracked frame
a, b = idx[-3] + 1, idx[-2] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    m = re.search(r"(k_\w+|__amd\w+)", r["Kernel_Name"])
    print("  %-24s %8.1f -> %8.1f  (%.1f)" % (m.group(1) if m else r["Kernel_Name"][:24], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
