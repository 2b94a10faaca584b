#!/bin/bash
O=gpurun_out/r06cs; mkdir -p $O
python3 - <<'P'
import sys; sys.path.insert(0, '.')
import bench
from eao_fusion_amd import synth
bench.mixed_load_inputs('gpurun_out/r06cs', synth)
P
for V in "" "-DEAO_BENCH_EDITED_MAPPOINT"; do
g++ -O2 -std=c++17 -DEAOFUSION_FORCE_CV_COMPAT $V -I include tests/cpp/adapter_bench.cpp -o $O/ab -L eao_fusion_amd -leaofusion_hip -Wl,-rpath,$PWD/eao_fusion_amd -Wl,-rpath,/opt/rocm/lib -pthread
for M in gba-walk gba gba-walk gba; do echo "$V $M: $($O/ab $O/map.bin $M | grep call_ms | cut -c1-110)"; done
done
