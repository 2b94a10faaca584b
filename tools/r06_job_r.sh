#!/bin/bash
# host set-up as a crew session: parity tests of the LM paths, then the map-scale stamps
O=gpurun_out/r06r; mkdir -p $O
python3 -m pytest tests/test_gpu_lm.py tests/test_gpu_threads.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py 2>&1 | grep -E 'map-scale plan|host set-up|map-scale wall|banded GBA' | cut -c1-420 > $O/banded.txt
EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba.py 2>&1 | grep -E 'map-scale plan|host set-up|map-scale wall|^GBA' | cut -c1-420 > $O/gba.txt
tail -n 4 $O/banded.txt | cut -c1-330; tail -n 4 $O/gba.txt | cut -c1-330
python3 tools/dbg_gba.py | tail -n 2; python3 tools/dbg_gba_banded.py | tail -n 2
