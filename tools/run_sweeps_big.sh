cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_sweeps_big.txt; : > $O
run() { echo "## tools/$1 ${@:2}" >> $O; timeout -k 10 400 python3 tools/$1 "${@:2}" 2>&1 | grep -v "amdgpu.ids" | grep -E "MISMATCH|iters|sweep|EXCEPTION" | tail -12 | cut -c1-420 >> $O; echo "[$(date +%T)] $1 done: $(tail -1 $O | cut -c1-160)"; }
run sweep_lm.py 611 900
run sweep_lm_batch.py 612 120
run sweep_pose.py 613 8000
run sweep_track.py 614 4000
run sweep_track_stages.py 615 1500
run sweep_search.py 616 800 handles
run sweep_search.py 617 500
run sweep_match.py 618 600
run sweep_orb.py 619 500
