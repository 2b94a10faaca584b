#!/bin/bash
# the round's closing capture: the GPU suite, the default bench line, the map-scale stamps (everything else: tools/prof_round.sh r06)
O=gpurun_out
python3 -m pytest tests -m gpu -q > $O/r06_gputests.log 2>&1; tail -3 $O/r06_gputests.log
python3 bench.py > $O/r06_bench_line.json 2> $O/r06_bench_line.err; tail -c 300 $O/r06_bench_line.json; echo
EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py 2>&1 | grep -E 'map-scale plan|host set-up|map-scale wall|banded GBA' | cut -c1-420 > $O/r06_gba_banded_host_stamps.txt
EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba.py 2>&1 | grep -E 'map-scale plan|host set-up|map-scale wall|^GBA' | cut -c1-420 > $O/r06_gba_host_stamps.txt
tail -2 $O/r06_gba_banded_host_stamps.txt $O/r06_gba_host_stamps.txt | cut -c1-200
