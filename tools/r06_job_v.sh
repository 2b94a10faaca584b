#!/bin/bash
# after the orb.hip split: the whole GPU suite, then the step
O=gpurun_out/r06v; mkdir -p $O
python3 -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
python3 bench.py --no-extra --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r06v/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('valu_frac'))"
