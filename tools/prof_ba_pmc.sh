#!/bin/bash
# SQ counters of the batched local-BA kernels (one group, 25 windows): scalar vs vector instruction counts, busy cycles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ba_pmc
EAO_BA_BATCH_GROUPS=1 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
    --output-format csv -d gpurun_out/ba_pmc -o p -- python3 tools/dbg_ba_batch.py > gpurun_out/ba_pmc.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/ba_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_ba_\w+)", r["Kernel_Name"])
    if not m: continue
    k = m.group(1); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
print("%-22s %8s %10s %12s %12s %12s %12s" % ("kernel", "launches", "waves", "VALU/wave", "SALU/wave", "LDS/wave", "VMEM/wave"))
for k in sorted(acc, key=lambda k: -acc[k]["SQ_INSTS_VALU"]):
    n = len(disp[k]); w = acc[k]["SQ_WAVES"] / n
    print("%-22s %8d %10.0f %12.1f %12.1f %12.1f %12.1f" % (k, n, w, acc[k]["SQ_INSTS_VALU"] / n / w, acc[k]["SQ_INSTS_SALU"] / n / w, acc[k]["SQ_INSTS_LDS"] / n / w,
                                                  (acc[k]["SQ_INSTS_VMEM_RD"] + acc[k]["SQ_INSTS_VMEM_WR"]) / n / w))
PY
