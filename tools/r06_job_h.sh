#!/bin/bash
# side-stream priority fix (headline A/B), by-value look-ahead records (GBA timing), LM tests
O=gpurun_out/r06h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_lm.py tests/test_gpu_orb.py tests/test_gpu_adapters.py tests/test_integration_snippets.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2 3; do
  python bench.py --no-extra --no-cpu-baseline > $O/prio_$i.json 2>> $O/err.log || exit 1
  EAO_STREAM_PRIORITY=0 python bench.py --no-extra --no-cpu-baseline > $O/noprio_$i.json 2>> $O/err.log || exit 1
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06h/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_cold'), d['roofline']['avg_launch_ms'])
P
EAO_DBG_ORACLE=0 EAO_DEBUG_STAMPS=1 python3 tools/dbg_gba_banded.py 2>&1 | grep -E "banded GBA|map-scale wall" | cut -c1-200
EAO_DEBUG_STAMPS=1 python3 tools/dbg_gba.py 2>&1 | grep -E "^GBA|map-scale wall" | cut -c1-200
