#!/bin/bash
# same-box A/B of the map-scale host set-up: gpurun_ab/libeaofusion_hip_head.so (the previous commit) against the tree's library, alternating, three rounds
for R in 1 2 3; do for L in head new; do
  if [ $L = head ]; then export EAO_LIB_PATH=gpurun_ab/libeaofusion_hip_head.so; else unset EAO_LIB_PATH; fi
  for T in dbg_gba_banded.py dbg_gba.py; do
    EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/$T 2>&1 | grep -E 'map-scale wall' | tail -n 2 | sed "s/^/$L $T /" | cut -c1-160
  done
done; done
