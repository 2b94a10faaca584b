#!/bin/bash
# kernel timeline of one single-frame ORB step under torch's HIP runtime
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tr_single
EAO_DBG_BATCH=1 EAO_DBG_STEPS=20 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_single -o t -- python3 tools/dbg_step_torch.py > gpurun_out/tr_single.log 2>&1
python3 tools/print_step_timeline.py gpurun_out/tr_single
