#!/bin/bash
# usage (on the GPU box): EAO_DBG_KF=500 EAO_DBG_PTS=50000 bash tools/prof_gba.sh <tag> ; per-kernel durations of 3 map-scale BundleAdjustment calls
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-gba}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 tools/dbg_gba.py > gpurun_out/prof_$tag.log 2>&1
tail -2 gpurun_out/prof_$tag.log
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" gpurun_out/prof_${tag}_summary.csv "python3 tools/dbg_gba.py (EAO_DBG_KF=$EAO_DBG_KF EAO_DBG_PTS=$EAO_DBG_PTS)"
