#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O
python - <<'P' || exit 1
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from eao_fusion_amd import synth
bench.mixed_load_inputs("/tmp", synth)
P
/opt/rocm/bin/hipcc -O2 -std=c++17 -DEAOFUSION_FORCE_CV_COMPAT -I include tests/cpp/mixed_load.cpp -o /tmp/mixed_load -L eao_fusion_amd -leaofusion_hip -Wl,-rpath,$PWD/eao_fusion_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
for i in 1 2 3; do
  /tmp/mixed_load /tmp/problem.bin /tmp/windows.bin /tmp/map.bin 800 2000 12 3 > $O/run_$i.json 2> $O/run_$i.err
  EAO_STREAM_PRIORITY=0 /tmp/mixed_load /tmp/problem.bin /tmp/windows.bin /tmp/map.bin 800 2000 12 3 > $O/noprio_$i.json 2> $O/noprio_$i.err
done
grep -o '"beside[a-z_0-9]*": {\|"distinct_results": {[^}]*}' $O/*.json | cut -c1-300
