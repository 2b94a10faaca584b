for cfg in "0 0" "1 0" "1 1"; do set -- $cfg; for r in 1 2; do
echo -n "blur_stream=$1 split=$2: "; EAO_ORB_BLUR_STREAM=$1 EAO_ORB_SPLIT=$2 EAO_DBG_STEPS=300 python3 tools/dbg_lanes.py | tail -1
done; done
