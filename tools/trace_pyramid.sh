#!/bin/bash
# kernel timeline of one 64-frame ORB step with the pyramid as ONE launch (EAO_ORB_PYRAMID=fused) and as the chain
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in ${1:-fused}; do
rm -rf gpurun_out/tr_pyr_$mode
EAO_ORB_PYRAMID=$mode EAO_DBG_STEPS=20 EAO_DBG_STREAM=own rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_pyr_$mode -o t -- python3 tools/dbg_lanes.py > gpurun_out/tr_pyr_$mode.log 2>&1
tail -1 gpurun_out/tr_pyr_$mode.log
python3 tools/print_step_timeline.py gpurun_out/tr_pyr_$mode
done
