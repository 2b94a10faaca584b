for u in 0 4 5 6; do for r in 1 2; do
echo -n "up_split=$u: "; EAO_ORB_UP_SPLIT=$u EAO_DBG_STEPS=300 python3 tools/dbg_lanes.py | tail -1
done; done
