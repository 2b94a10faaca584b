#!/bin/bash
# Round-5 extras of the evidence run (on the GPU box): per-pass cycle stamps of PoseOptimization, cycle stamps + host-phase stamps + rocprofv3 kernel summaries of the
# two map-scale BundleAdjustment benchmarks.  Outputs gpurun_out/r05_* (copy into profiles/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python3 tools/dbg_pose_waves.py > $O/r05_pose_stamps.txt 2>&1
EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba.py > $O/r05_bal_step_stamps.txt 2>&1
rm -rf $O/r05_gba $O/r05_gbab
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_gba -o s -- python3 tools/dbg_gba.py > $O/r05_gba.log 2>&1
f=$(find $O/r05_gba -name "*kernel_stats.csv" | head -1); python3 tools/summarize_rocprof.py "$f" $O/r05_gba_kernel_stats.csv "python3 tools/dbg_gba.py (200 KF x 20000 MP, 3 calls)" | head -14
EAO_DBG_ORACLE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_gbab -o s -- python3 tools/dbg_gba_banded.py > $O/r05_gbab.log 2>&1
f=$(find $O/r05_gbab -name "*kernel_stats.csv" | head -1); python3 tools/summarize_rocprof.py "$f" $O/r05_gba_banded_kernel_stats.csv "EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py (1000 KF x 50000 MP band 11, 3 calls)" | head -14
EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py 2>&1 | grep -E 'host set-up|map-scale wall|banded GBA' | tail -5 > $O/r05_gba_banded_host_stamps.txt
grep -E 'host set-up|map-scale wall|^GBA' $O/r05_bal_step_stamps.txt | tail -5 > $O/r05_gba_host_stamps.txt
cat $O/r05_gba_host_stamps.txt $O/r05_gba_banded_host_stamps.txt | cut -c1-300
