#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/run_sweeps.sh r06 3 B
O=gpurun_out/r06_sweeps_lm.txt; : > $O
for seed in 701 702; do
  echo "## tools/sweep_lm.py $seed 400" >> $O
  timeout -k 10 500 python3 tools/sweep_lm.py $seed 400 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|iters|sweep|exception" | cut -c1-420 >> $O
done
cat $O
