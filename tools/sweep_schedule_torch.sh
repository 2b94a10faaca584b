#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for mid in 0 3 4; do for qe in 0 1; do
echo -n "mid=$mid qtearly=$qe: "; EAO_ORB_MID=$mid EAO_ORB_QT_EARLY=$qe python3 tools/dbg_step_torch.py 2>&1 | tail -1
done; done
echo -n "fused pyramid: "; EAO_ORB_PYRAMID=fused python3 tools/dbg_step_torch.py 2>&1 | tail -1
