#!/bin/bash
# window groups of eao_local_ba_batch (each group a chain on its own stream): wall time of the 25-window call
cd "$GRAFT_REPO_ROOT"
for g in ${GROUPS_LIST:-1 2 3 4}; do
  echo "== EAO_BA_BATCH_GROUPS=$g"
  EAO_BA_BATCH_GROUPS=$g python3 tools/dbg_ba_batch.py 2>&1 | grep eao_local_ba_batch
done
