#!/bin/bash
# kernel timeline of one 64-frame ORB step under the environment given as arguments:  bash tools/trace_env.sh EAO_ORB_QT_EARLY=0 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/tr_env
EAO_DBG_STEPS=20 EAO_DBG_STREAM=own rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_env -o t -- python3 tools/dbg_lanes.py > gpurun_out/tr_env.log 2>&1
tail -1 gpurun_out/tr_env.log
python3 tools/print_step_timeline.py gpurun_out/tr_env
