#!/bin/bash
# SQ counters of the ORB kernels (run on the GPU box through gpurun): two --pmc passes over tools/dbg_lanes.py
#   bash tools/prof_orb_pmc.sh  ->  gpurun_out/pmc_orb_{a,b}/ + a per-kernel table on stdout
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export EAO_DBG_STEPS=4
rm -rf gpurun_out/pmc_orb_a gpurun_out/pmc_orb_b
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d gpurun_out/pmc_orb_a -o a -- python3 tools/dbg_lanes.py > gpurun_out/pmc_orb_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d gpurun_out/pmc_orb_b -o b -- python3 tools/dbg_lanes.py > gpurun_out/pmc_orb_b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    fs = glob.glob("gpurun_out/pmc_orb_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: print("no counter file for pass", tag); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen: seen.add(key); calls[k] += 1
    names = sorted({c for k in acc for c in acc[k]})
    print("pass %s (per launch)" % tag)
    print("kernel".ljust(20) + "".join(n.replace("SQ_", "").rjust(18) for n in names))
    for k in sorted(acc):
        print(k[:20].ljust(20) + "".join(("%.4g" % (acc[k][n] / calls[k])).rjust(18) for n in names))
PY
