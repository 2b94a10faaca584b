#!/bin/bash
# A/B of the headline step: stream classes on / off, alternating
O=gpurun_out/r06g; mkdir -p $O
for i in 1 2 3 4; do
  python bench.py --no-extra --no-cpu-baseline > $O/prio_$i.json 2>> $O/err.log || exit 1
  EAO_STREAM_PRIORITY=0 python bench.py --no-extra --no-cpu-baseline > $O/noprio_$i.json 2>> $O/err.log || exit 1
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06g/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_cold'), d['roofline']['avg_launch_ms'])
P
