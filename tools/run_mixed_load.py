"""Runs tests/cpp/mixed_load.cpp as bench.py does (extra.mixed_load) and prints the JSON: python tools/run_mixed_load.py [frames] [period_us]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eao_fusion_amd import synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
period = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
print(json.dumps(bench.measure_mixed_load(synth, frames, period), indent=1))
