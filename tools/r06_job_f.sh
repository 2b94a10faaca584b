#!/bin/bash
set -o pipefail
O=gpurun_out/r06f; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -8 $O/tests.log
bash tools/prof_fast_census.sh || exit 1
python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r06f/bench_line.json').read().strip().splitlines()[-1])
print(json.dumps({k:v for k,v in d['roofline'].items() if not isinstance(v,(dict,list))}, indent=0))
print(d['value'], d['ms_per_step'], d.get('ms_per_step_cold'))
cs=d['extra']['class_surface']; print(cs.get('local_bundle_adjustment'), cs.get('local_bundle_adjustment_with_accessors'))
print({k:d['extra'][k].get('ms_per_call') for k in ('bundle_adjustment_map_scale','bundle_adjustment_map_scale_banded')})
P
