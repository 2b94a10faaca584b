#!/bin/bash
# kernel timeline of one 64-frame ORB step: under torch's bundled HIP runtime and under /opt/rocm's
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tr_torch gpurun_out/tr_rocm
EAO_DBG_STEPS=20 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_torch -o t -- python3 tools/dbg_step_torch.py > gpurun_out/tr_torch.log 2>&1
EAO_DBG_STEPS=20 EAO_DBG_STREAM=own rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_rocm -o t -- python3 tools/dbg_lanes.py > gpurun_out/tr_rocm.log 2>&1
tail -1 gpurun_out/tr_torch.log; tail -1 gpurun_out/tr_rocm.log
