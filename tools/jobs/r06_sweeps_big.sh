#!/bin/bash
# the round's large randomised sweeps with the final code (fresh seeds)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_sweeps_big.txt
echo "# ---- large sweeps with the round's final code (fresh seeds; tools/jobs/r06_sweeps_big.sh)" > $O
run() { echo "## tools/$1 ${@:2}" >> $O; timeout -k 10 900 python3 tools/$1 "${@:2}" 2>&1 | grep -v "amdgpu.ids" | grep -E "MISMATCH|iters|sweep|EXCEPTION" | tail -12 | cut -c1-420 >> $O; echo "[$(date +%T)] $1 done: $(tail -n 1 $O)"; }
run sweep_lm.py 811 900
run sweep_lm_batch.py 812 60
run sweep_pose.py 813 4000
run sweep_track.py 814 2000
run sweep_track_stages.py 815 800
run sweep_search.py 816 400 handles
run sweep_match.py 818 300
run sweep_orb.py 819 300
