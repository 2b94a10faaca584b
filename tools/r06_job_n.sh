#!/bin/bash
O=gpurun_out/r06n; mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r06n/bench_line.json').read().strip().splitlines()[-1])
print({k:v for k,v in d['roofline'].items() if not isinstance(v,(dict,list))})
print(d['value'], d['ms_per_step'], d.get('ms_per_step_cold'))
ml=d['extra']['mixed_load']['priorities']
for v in ('device_chain','class_surface'):
    for sc,r in ml[v].items(): print(v,sc,r['frame_ms'], r['results_identical'])
P
python - <<'P' || exit 1
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from eao_fusion_amd import synth
bench.mixed_load_inputs("/tmp", synth)
P
/opt/rocm/bin/hipcc -O2 -std=c++17 -DEAOFUSION_FORCE_CV_COMPAT -I include tests/cpp/mixed_load.cpp -o /tmp/mixed_load -L eao_fusion_amd -leaofusion_hip -Wl,-rpath,$PWD/eao_fusion_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
for g in 4 2 1; do
  EAO_BA_BATCH_GROUPS=$g /tmp/mixed_load /tmp/problem.bin /tmp/windows.bin /tmp/map.bin 1200 2000 5 1 > $O/groups_$g.json 2> $O/groups_$g.err
  python - <<P
import json
d=json.load(open('gpurun_out/r06n/groups_$g.json'))
for sc,r in d['device_chain'].items():
    if isinstance(r,dict): print('groups $g', sc, r['frame_ms'], {k:r[k]['p50'] for k in r if k=='lba_batch25_ms'}, 'alone', d['alone']['lba_batch25_ms']['p50'])
P
done
