#!/bin/bash
# durations of the chain schedule's three FAST launches when each runs ALONE (profiled calls run the stages one after the other)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tr_fs
EAO_FAST_SPLIT_PROF=1 EAO_DBG_PROF=1 EAO_DBG_STEPS=10 EAO_DBG_STREAM=own rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_fs -o t -- python3 tools/dbg_lanes.py > gpurun_out/tr_fs.log 2>&1
python3 - <<'EOF'
import csv, glob
f = glob.glob("gpurun_out/tr_fs/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
fs = [r for r in rows if "k_fast_cells" in r["Kernel_Name"]][-9:]
for r in fs: print("k_fast_cells grid %s: %.1f us" % (r.get("Grid_Size_X") or r.get("Grid_Size"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
EOF
