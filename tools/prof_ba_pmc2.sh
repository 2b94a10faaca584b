#!/bin/bash
# SQ / memory counters of the batched local-BA kernels (one group, 25 windows), one rocprofv3 --pmc pass per counter set (never combined with traces):
#   bash tools/prof_ba_pmc2.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/ba_pmc_${tag}_$i
  EAO_BA_BATCH_GROUPS=1 EAO_DBG_WINDOWS=${EAO_DBG_WINDOWS:-25} rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/ba_pmc_${tag}_$i -o p -- python3 tools/dbg_ba_batch.py > gpurun_out/ba_pmc_${tag}_$i.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, collections, re, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob("gpurun_out/ba_pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_ba_\w+)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1); c = r["Counter_Name"]; acc[k][c] += float(r["Counter_Value"]); disp[k][c].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
print("per launch (average over the launches of each kernel):")
print("%-24s" % "kernel" + "".join("%22s" % c for c in names))
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    print("%-24s" % k + "".join("%22.4g" % (acc[k][c] / max(len(disp[k][c]), 1)) for c in names))
PY
