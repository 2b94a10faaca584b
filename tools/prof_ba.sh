#!/bin/bash
# usage (on the GPU box): bash tools/prof_ba.sh <tag> ; prints per-kernel average durations of 3 LocalBundleAdjustment calls
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-ba}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 tools/dbg_ba.py > gpurun_out/prof_$tag.log 2>&1
tail -1 gpurun_out/prof_$tag.log
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 tools/ba_timeline.py $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e3:9.1f} us")
PY
