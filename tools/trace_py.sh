#!/bin/bash
# kernel trace of a python tool on the GPU box:  bash tools/trace_py.sh tools/<script>.py <tag>  ->  gpurun_out/<tag>_trace/
S=$1; T=${2:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${T}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_trace -o t -- python3 $S > gpurun_out/${T}_trace.log 2>&1
f=$(find gpurun_out/${T}_trace -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" gpurun_out/${T}_kernel_stats.csv "$S under rocprofv3" | head -30
