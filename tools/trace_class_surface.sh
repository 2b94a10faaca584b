#!/bin/bash
# Kernel + memory-copy trace of tests/cpp/adapter_bench (the class-surface calls), on the GPU box:
#   bash tools/trace_class_surface.sh <tag>   ->  gpurun_out/<tag>_cs_trace/ (csv), gpurun_out/<tag>_cs_kernel_stats.csv
T=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O/${T}_cs && rm -rf $O/${T}_cs_trace
g++ -O2 -std=c++17 -DEAOFUSION_FORCE_CV_COMPAT -I include tests/cpp/adapter_bench.cpp -o $O/${T}_cs/adapter_bench -L eao_fusion_amd -leaofusion_hip \
    -Wl,-rpath,$GRAFT_REPO_ROOT/eao_fusion_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
python3 -c "
import sys; sys.path.insert(0, '.')
import bench
from eao_fusion_amd import synth
bench.class_surface_problem('$O/${T}_cs/problem.bin', synth)"
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/${T}_cs_trace -o cs -- $O/${T}_cs/adapter_bench $O/${T}_cs/problem.bin > $O/${T}_cs_under_rocprof.json 2> $O/${T}_cs_trace.err
f=$(find $O/${T}_cs_trace -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" $O/${T}_cs_kernel_stats.csv "adapter_bench under rocprofv3" | head -40
