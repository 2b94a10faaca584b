for tf in 0 3 4 5 6; do for r in 1 2; do
echo -n "top_from=$tf: "; EAO_ORB_TOP_FROM=$tf EAO_DBG_STEPS=300 python3 tools/dbg_lanes.py | tail -1
done; done
