#!/bin/bash
O=gpurun_out/r06o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_lm.py tests/test_gpu_threads.py tests/test_gpu_sequence.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/run_mixed_load.py 1200 2000 > $O/mixed.json 2> $O/mixed.err || { tail -5 $O/mixed.err; exit 1; }
python - <<'P'
import json
d=json.load(open('gpurun_out/r06o/mixed.json'))
for mode in ('priorities','no_priorities'):
    m=d[mode]
    for v in ('device_chain','class_surface'):
        for sc,r in m[v].items():
            f=r['frame_ms']; bg={k:r[k]['p50'] for k in r if k in ('lba_class_surface_ms','lba_batch25_ms','map_ba_ms')}
            print(mode, v, sc, 'p50 %.3f p90 %.3f p99 %.3f max %.3f same %s'%(f['p50'],f['p90'],f['p99'],f['max'],r['results_identical']), bg)
P
