#!/bin/bash
# usage (on the GPU box): bash tools/prof_py.sh <tag> <script.py> ; rocprofv3 kernel stats of one python tool, summary in gpurun_out/prof_<tag>_summary.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; script=$2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 $script > gpurun_out/prof_$tag.log 2>&1
grep -v rocprofv3 gpurun_out/prof_$tag.log | tail -3
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" gpurun_out/prof_${tag}_summary.csv "python3 $script"
