#!/bin/bash
# The class-surface bench alone (tests/cpp/adapter_bench.cpp through bench.measure_class_surface), on the GPU box:
#   bash tools/run_class_surface.sh <tag>   ->  gpurun_out/<tag>_class_surface.json
T=${1:-r05}
cd "$GRAFT_REPO_ROOT"
python3 - > gpurun_out/${T}_class_surface.json 2> gpurun_out/${T}_class_surface.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import torch  # noqa: F401
import bench
from eao_fusion_amd import synth
print(json.dumps(bench.measure_class_surface(synth), indent=1))
PY
cat gpurun_out/${T}_class_surface.json
