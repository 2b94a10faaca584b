for k in 0 4 8 16; do for r in 1 2; do
echo -n "kpw=$k: "; EAO_ORB_KPW=$k EAO_DBG_STEPS=300 python3 tools/dbg_lanes.py | tail -1
done; done
