#!/bin/bash
# round 6, first GPU job: the suite, the mixed-load harness (both priority modes), A/B of the per-call event
set -o pipefail
mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -x -q > gpurun_out/r06a/tests.log 2>&1 || { tail -30 gpurun_out/r06a/tests.log; exit 1; }
tail -3 gpurun_out/r06a/tests.log
python tools/run_mixed_load.py > gpurun_out/r06a/mixed.json 2> gpurun_out/r06a/mixed.err || { tail -20 gpurun_out/r06a/mixed.err; exit 1; }
for i in 1 2 3; do
  python bench.py --no-extra --no-cpu-baseline > gpurun_out/r06a/bench_default_$i.json 2>> gpurun_out/r06a/bench.err || exit 1
  EAO_ORB_LAST_EVENT=always python bench.py --no-extra --no-cpu-baseline > gpurun_out/r06a/bench_always_$i.json 2>> gpurun_out/r06a/bench.err || exit 1
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06a/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d.get('ms_per_step_cold'))
P
