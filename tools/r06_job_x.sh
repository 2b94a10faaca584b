#!/bin/bash
# fresh seeds of the LM / ORB sweeps with the set-up sessions, the device pair lists and the staged uploads
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_sweeps_late.txt
echo "# ---- after the set-up sessions of the host crew, k_bal_pair_fill and the pinned upload staging (two fresh seeds each; sweep_lm draws EAO_BA_ND and EAO_BA_SETUP_THREADS)" > $O
run() { echo "## tools/$1 ${@:2}" >> $O; timeout -k 10 500 python3 tools/$1 "${@:2}" 2>&1 | grep -v "amdgpu.ids" | grep -E "MISMATCH|iters|sweep|EXCEPTION" | tail -8 | cut -c1-420 >> $O; echo "[$(date +%T)] $1 done: $(tail -1 $O)"; }
run sweep_lm.py 801 300
run sweep_lm.py 802 300
run sweep_lm_batch.py 803 20
run sweep_orb.py 804 120
run sweep_orb_batch.py 805 40
