#!/bin/bash
# the randomised parity sweeps of a round in one call (on the GPU box):  bash tools/run_sweeps.sh r04 [scale]   -> gpurun_out/r04_sweeps.txt
R=${1:-r04}; S=${2:-1}; PART=${3:-AB}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${R}_sweeps_${PART}.txt
echo "# randomised parity sweeps against the CPU oracle with the round's final code (tools/sweep_*.py, one MI355X; seed, cases per line)" > $O
run() { echo "## tools/$1 ${@:2}" >> $O; timeout -k 10 500 python3 tools/$1 "${@:2}" 2>&1 | grep -v "amdgpu.ids" | tail -4 >> $O; echo "[$(date +%T)] $1 done: $(tail -1 $O)"; }
if [[ $PART == *A* ]]; then
run sweep_orb.py 401 $((60 * S))
run sweep_lm.py 402 $((80 * S))
run sweep_lm_batch.py 403 $((10 * S))
run sweep_pose.py 404 $((600 * S))
fi
if [[ $PART == *B* ]]; then
run sweep_track.py 405 $((800 * S))
run sweep_track_stages.py 406 $((300 * S))
run sweep_search.py 407 $((100 * S))
run sweep_match.py 408 $((100 * S))
fi
