cd "$GRAFT_REPO_ROOT"; O=gpurun_out
EAO_LIB_PATH=gpurun_ab/libeaofusion_hip_head.so python3 tools/ab_gba_bits.py dump $O/gba_head.npz 2>&1 | tail -1
python3 tools/ab_gba_bits.py dump $O/gba_new.npz 2>&1 | tail -1
python3 tools/ab_gba_bits.py cmp $O/gba_head.npz $O/gba_new.npz
EAO_DEBUG_STAMPS=1 python3 tools/dbg_gba.py 2>&1 | grep -E "host set-up|map-scale wall|GBA" | tail -3 | cut -c1-330
EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py 2>&1 | grep -E "host set-up|map-scale wall|GBA" | tail -3 | cut -c1-330
timeout -k 10 400 python3 -m pytest tests/test_gpu_lm.py -x -q 2>&1 | tail -3
