for l in 1 2 4; do for mid in 0 3; do for qe in 0 1; do
echo -n "lanes=$l mid=$mid qtearly=$qe: "; EAO_ORB_LANES=$l EAO_ORB_MID=$mid EAO_ORB_QT_EARLY=$qe EAO_DBG_STEPS=200 python3 tools/dbg_lanes.py | tail -1
done; done; done
