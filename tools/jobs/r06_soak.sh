#!/bin/bash
# LocalBundleAdjustment looping beside a neighbour for 120 s each (1000-keyframe map BA with its crew sessions; a 25-window batch; a second LBA loop): every result bit-identical?
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_soak.txt; : > $O
for M in gba batch lba; do echo "## python3 tools/dbg_lba_beside_gba.py 1000 120 $M" >> $O; timeout -k 10 300 python3 tools/dbg_lba_beside_gba.py 1000 120 $M 2>&1 | grep -E "calls|differ" >> $O; tail -n 2 $O; done
