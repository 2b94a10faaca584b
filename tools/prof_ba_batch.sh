#!/bin/bash
# per-kernel durations of eao_local_ba_batch (25 windows of BASELINE configs[4]):  bash tools/prof_ba_batch.sh [tag]
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ba_batch_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ba_batch_$tag -o s -- python3 tools/dbg_ba_batch.py > gpurun_out/ba_batch_$tag.log 2>&1
f=$(find gpurun_out/ba_batch_$tag -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" gpurun_out/ba_batch_$tag.csv "python3 tools/dbg_ba_batch.py (25 windows per call, 23 calls)" | head -20
tail -2 gpurun_out/ba_batch_$tag.log
