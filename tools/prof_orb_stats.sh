#!/bin/bash
# per-kernel durations of the ORB stages, each stage alone (EAO_DBG_PROF=1 runs them one after the other), 64-frame batch:
#   bash tools/prof_orb_stats.sh [tag]  ->  gpurun_out/orb_stats_<tag>.csv (+ the condensed table on stdout)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/orb_stats_$tag
EAO_DBG_PROF=1 EAO_DBG_STEPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/orb_stats_$tag -o s -- python3 tools/dbg_lanes.py > gpurun_out/orb_stats_$tag.log 2>&1
f=$(find gpurun_out/orb_stats_$tag -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" gpurun_out/orb_stats_$tag.csv "EAO_DBG_PROF=1 python3 tools/dbg_lanes.py (64 frames, stages sequential)" | head -14
EAO_DBG_STEPS=50 python3 tools/dbg_lanes.py | tail -1
