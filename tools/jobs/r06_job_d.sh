#!/bin/bash
# round 6: the lm.hip split is a pure move (A/B bits against the previous commit's build), the suite, the mixed load with the spin wait, the CU-reserve experiment
set -o pipefail
O=gpurun_out/r06d; mkdir -p $O
bash tools/ab_pose_job.sh > $O/ab_pose.txt 2>&1; head -4 $O/ab_pose.txt
bash tools/ab_gba_job.sh > $O/ab_gba.txt 2>&1; head -4 $O/ab_gba.txt
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python - <<'P' || exit 1
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from eao_fusion_amd import synth
bench.mixed_load_inputs("/tmp", synth)
P
/opt/rocm/bin/hipcc -O2 -std=c++17 -DEAOFUSION_FORCE_CV_COMPAT -I include tests/cpp/mixed_load.cpp -o /tmp/mixed_load -L eao_fusion_amd -leaofusion_hip -Wl,-rpath,$PWD/eao_fusion_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
run() { name=$1; mask=$2; shift; shift; env "$@" /tmp/mixed_load /tmp/problem.bin /tmp/windows.bin /tmp/map.bin 1500 2000 $mask 1 > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; }; }
run default 15 A=1
run default2 15 A=1
run nospin 3 EAO_SPIN_WAIT=0
run reserve1 15 EAO_BULK_CU_RESERVE=1
run reserve2 15 EAO_BULK_CU_RESERVE=2
run noprio 15 EAO_STREAM_PRIORITY=0
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06d/*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'bad', e); continue
    v=d['device_chain']
    for sc,r in v.items():
        if not isinstance(r,dict): continue
        fm=r['frame_ms']
        bg={k:(r[k]['p50']) for k in r if k in ('lba_class_surface_ms','lba_batch25_ms','map_ba_ms')}
        print('%-14s %-22s p50 %.3f p90 %.3f p99 %.3f max %.3f | ext p99 %.3f mm p99 %.3f lm p99 %.3f same %s %s'%(f.split('/')[-1][:-5], sc, fm['p50'],fm['p90'],fm['p99'],fm['max'],r['extract_ms']['p99'],r['motion_model_ms']['p99'],r['local_map_ms']['p99'], r['results_identical'], bg))
P
