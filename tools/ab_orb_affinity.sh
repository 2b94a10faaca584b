cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for cfg in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $cfg
  for rep in 1 2; do
    EAO_RESIZE_AFFINITY=$1 EAO_BLUR_AFFINITY=$2 python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('resize_aff=$1 blur_aff=$2  ms_per_step %.4f cold %.4f  stages' % (d['ms_per_step'], d.get('ms_per_step_cold', 0)), r['stage_ms'])"
  done
done
# traffic with both on (default) 
bash tools/prof_pmc_traffic.sh > $O/traffic_aff.log 2>&1
python3 -c "
import json
t = json.load(open('gpurun_out/pmc_traffic.json'))
for k, v in t['kernels'].items(): print(k, v['kernel'], 'fetch %.1f MB write %.1f MB corrected %.1f MB' % (v['fetch_bytes_per_step'] / 1e6, v['write_bytes_per_step'] / 1e6, v['hbm_bytes_per_step_corrected'] / 1e6))"
