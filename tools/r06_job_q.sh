#!/bin/bash
O=gpurun_out/r06q; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_lm.py tests/test_gpu_threads.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
EAO_DBG_ORACLE=0 EAO_DEBUG_STAMPS=1 python3 tools/dbg_gba_banded.py 2>&1 | grep -E "banded GBA|map-scale wall|host set-up\] (observer|covis)" | cut -c1-330 | tail -8
EAO_DEBUG_STAMPS=1 python3 tools/dbg_gba.py 2>&1 | grep -E "^GBA|map-scale wall|host set-up\] (observer|covis)" | cut -c1-330 | tail -8
