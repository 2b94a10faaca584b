#!/bin/bash
# round 6: kernel timeline of the mixed-load harness (device chain, idle + beside a looping LBA), with and without stream priorities
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06b; mkdir -p $O
python - <<'P' || exit 1
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from eao_fusion_amd import synth
bench.mixed_load_inputs("/tmp", synth)
P
/opt/rocm/bin/hipcc -O2 -std=c++17 -DEAOFUSION_FORCE_CV_COMPAT -I include tests/cpp/mixed_load.cpp -o /tmp/mixed_load -L eao_fusion_amd -leaofusion_hip -Wl,-rpath,$PWD/eao_fusion_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
for mode in on off; do
  if [ $mode = off ]; then export EAO_STREAM_PRIORITY=0; fi
  rm -rf $O/tr_$mode
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr_$mode -o t -- /tmp/mixed_load /tmp/problem.bin /tmp/windows.bin /tmp/map.bin 600 2000 3 1 > $O/mixed_$mode.json 2> $O/mixed_$mode.err || { tail -5 $O/mixed_$mode.err; exit 1; }
  python tools/analyze_mixed_trace.py $O/tr_$mode > $O/analysis_$mode.txt 2>&1
  rm -rf $O/tr_$mode
done
cat $O/analysis_on.txt
