cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for T in 256 512 1024; do
echo -n "EAO_BA_LIN_THREADS=$T: "; EAO_BA_LIN_THREADS=$T python3 tools/dbg_ba_cabi.py 2>/dev/null | tail -2 | head -1 | cut -c1-120
done
python3 tools/dbg_ba_cabi.py 2>/dev/null | tail -2 | head -1 | cut -c1-120
EAO_BA_LIN_THREADS=256 python -m pytest tests/test_gpu_lm.py -x -q 2>&1 | tail -1
python -m pytest tests/test_gpu_lm.py tests/test_golden.py tests/test_gpu_adapters.py -x -q 2>&1 | tail -1
