#!/bin/bash
O=gpurun_out/r06s; mkdir -p $O
for T in 12; do
  EAO_DEBUG_CREW=1 EAO_BA_SETUP_THREADS=$T EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py 2>&1 | grep -E 'crew\]|host set-up\] (obs|covis)|map-scale wall' | tail -n 14 | cut -c1-330 > $O/banded_$T.txt
  echo "T=$T"; cat $O/banded_$T.txt
done
