#!/bin/bash
# ORB batch throughput for 1 / 2 / 4 concurrent stream lanes (EAO_ORB_LANES)
for l in 1 2 4; do
  EAO_ORB_LANES=$l python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null > /tmp/b_$l.json
  python - "$l" /tmp/b_$l.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read())
print("lanes", sys.argv[1], d["value"], d["ms_per_step"])
PY
done
