#!/bin/bash
# pass timings of the map-scale set-up session (third call of each benchmark map)
for T in dbg_gba_banded.py dbg_gba.py; do
  EAO_DEBUG_CREW=1 EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/$T 2>&1 | grep -E 'crew\]|host set-up\] (obs|covis)|map-scale wall' | tail -n 11 | cut -c1-330
done
