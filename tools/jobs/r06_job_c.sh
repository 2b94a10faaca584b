#!/bin/bash
# round 6: where T's slow frames fall in L's cycle; runtime knobs
set -o pipefail
O=gpurun_out/r06c; mkdir -p $O
python - <<'P' || exit 1
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from eao_fusion_amd import synth
bench.mixed_load_inputs("/tmp", synth)
P
/opt/rocm/bin/hipcc -O2 -std=c++17 -DEAOFUSION_FORCE_CV_COMPAT -I include tests/cpp/mixed_load.cpp -o /tmp/mixed_load -L eao_fusion_amd -leaofusion_hip -Wl,-rpath,$PWD/eao_fusion_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
run() { name=$1; shift; env "$@" /tmp/mixed_load /tmp/problem.bin /tmp/windows.bin /tmp/map.bin 1500 2000 3 1 > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; }; }
run default A=1
run default2 A=1
run nointr HSA_ENABLE_INTERRUPT=0
run hwq8 GPU_MAX_HW_QUEUES=8
run noprio EAO_STREAM_PRIORITY=0
run nopoll EAO_TRACK_POLL=0
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06c/*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'bad', e); continue
    v=d['device_chain']
    for sc in ('idle','beside_lba'):
        r=v[sc]; fm=r['frame_ms']
        print(f.split('/')[-1], sc, 'p50 %.3f p90 %.3f p99 %.3f max %.3f'%(fm['p50'],fm['p90'],fm['p99'],fm['max']), 'ext p99 %.3f mm p99 %.3f lm p99 %.3f'%(r['extract_ms']['p99'],r['motion_model_ms']['p99'],r['local_map_ms']['p99']), r.get('slow_frames_by_lba_phase',''))
P
