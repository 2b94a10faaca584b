#!/bin/bash
# round 6: the map-scale path in nested-dissection order: LM tests, timing of the two benchmark maps per segment count, kernel stats
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_lm.py tests/test_gpu_threads.py -x -q > $O/tests_lm.log 2>&1 || { tail -40 $O/tests_lm.log; exit 1; }
tail -2 $O/tests_lm.log
for nd in 1 0 8 16 24; do
  echo "== EAO_BA_ND=$nd" >> $O/banded.txt
  EAO_BA_ND=$nd EAO_DBG_ORACLE=0 EAO_DEBUG_STAMPS=1 python3 tools/dbg_gba_banded.py 2>&1 | grep -E "banded GBA|map-scale plan|host set-up|map-scale wall" | cut -c1-400 >> $O/banded.txt
done
for nd in 1 0; do
  echo "== EAO_BA_ND=$nd (200 KF cyclic)" >> $O/gba200.txt
  EAO_BA_ND=$nd EAO_DEBUG_STAMPS=1 python3 tools/dbg_gba.py 2>&1 | grep -E "GBA|map-scale plan|host set-up|map-scale wall" | cut -c1-400 >> $O/gba200.txt
done
cat $O/banded.txt $O/gba200.txt
export EAO_DBG_ORACLE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_banded -o p -- python3 tools/dbg_gba_banded.py > $O/prof_banded.log 2>&1 || true
F=$(find $O/prof_banded -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py $F $O/gba_banded_kernel_stats.csv "EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py (1000 KF x 50 000 MP, band 11, three calls)" > /dev/null 2>&1 || true
rm -rf $O/prof_banded
head -16 $O/gba_banded_kernel_stats.csv
