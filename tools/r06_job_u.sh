#!/bin/bash
# mixed load, 5000 frames per scenario: the frame uploads through the handle's pinned staging buffer (default) against the runtime's pageable path
O=gpurun_out/r06u; mkdir -p $O
for V in 1 0; do
EAO_ORB_PINNED_IN=$V python3 tools/run_mixed_load.py 5000 > $O/mixed_in$V.json 2> $O/mixed_in$V.err
python3 - $V <<'P'
import json,sys
d=json.load(open('gpurun_out/r06u/mixed_in%s.json'%sys.argv[1]))
for mode in ('priorities',):
    for var,v in d.get(mode,{}).items():
        if not isinstance(v,dict): continue
        for sc,s in v.items():
            if isinstance(s,dict) and 'frame_ms' in s:
                f=s['frame_ms']; e=s['extract_ms']
                print("pinned_in=%s %-13s %-22s p50 %.3f p99 %.3f max %.3f | extract p50 %.3f p99 %.3f max %.3f same %s"%(sys.argv[1],var,sc,f['p50'],f['p99'],f['max'],e['p50'],e['p99'],e['max'],s.get('results_identical')))
P
done
