for mode in chain fused; do
echo -n "pyramid=$mode: "; EAO_ORB_PYRAMID=$mode EAO_DBG_STEPS=200 python3 tools/dbg_lanes.py | tail -1
done
