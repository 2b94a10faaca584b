#!/bin/bash
# the LM engine beside another thread's LM calls: every result bit-identical?  (tools/dbg_lba_beside_gba.py)
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/dbg_lba_beside_gba.py 1000 ${SECS:-150} ${MODE:-lba} 2>&1 | grep -v amdgpu.ids | grep -E "calls|differ" | head -6; }
timeout -k 10 300 python -m pytest tests/test_gpu_lm.py -x -q -m gpu 2>&1 | tail -2
run EAO_STREAM_PRIORITY=0
run A=1
run EAO_STREAM_PRIORITY=0
MODE=batch SECS=100 run A=1
MODE=gba SECS=60 run A=1
