#!/bin/bash
# kernel trace of eao_local_ba_batch (25 windows) + per-queue timeline of the last call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tr_ba_batch
EAO_BA_BATCH_GROUPS=${1:-4} rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_ba_batch -o t -- python3 tools/dbg_ba_batch.py > gpurun_out/tr_ba_batch.log 2>&1
tail -2 gpurun_out/tr_ba_batch.log
python3 tools/ba_batch_timeline.py $(find gpurun_out/tr_ba_batch -name "*kernel_trace.csv" | head -1) ${1:-4}
