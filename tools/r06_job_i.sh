#!/bin/bash
O=gpurun_out/r06i2; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_orb.py tests/test_gpu_sequence.py tests/test_gpu_track.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2 3; do
  python bench.py --no-extra --no-cpu-baseline > $O/prio_$i.json 2>> $O/err.log || exit 1
  EAO_STREAM_PRIORITY=0 python bench.py --no-extra --no-cpu-baseline > $O/noprio_$i.json 2>> $O/err.log || exit 1
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06i2/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_cold'), d['roofline']['avg_launch_ms'])
P
python tools/run_mixed_load.py 1500 2000 > $O/mixed.json 2> $O/mixed.err || { tail -5 $O/mixed.err; exit 1; }
python - <<'P'
import json
d=json.load(open('gpurun_out/r06i2/mixed.json'))
for mode in ('priorities','no_priorities'):
    m=d[mode]
    for v in ('device_chain','class_surface'):
        for sc,r in m[v].items():
            f=r['frame_ms']; print(mode, v, sc, 'p50 %.3f p90 %.3f p99 %.3f max %.3f same %s'%(f['p50'],f['p90'],f['p99'],f['max'],r['results_identical']))
P
