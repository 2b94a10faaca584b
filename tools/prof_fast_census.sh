#!/bin/bash
# Per-phase DYNAMIC instruction census of k_fast_cells: the kernel leaves after phase 1 / 2 / 3 (EAO_FAST_STOP_AFTER) or runs in full; every variant's
# SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS per whole-stage launch (64 frames) and its duration; differences = the phases.  -> gpurun_out/r06_fast_census.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_fast; mkdir -p $O
for stop in 1 2 3 0; do
  rm -rf $O/pmc_$stop $O/tr_$stop
  EAO_FAST_STOP_AFTER=$stop rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_$stop -o c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_$stop.log 2>&1 || { tail -5 $O/pmc_$stop.log; exit 1; }
  EAO_FAST_STOP_AFTER=$stop rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$stop -o t -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $O/tr_$stop.log 2>&1 || { tail -5 $O/tr_$stop.log; exit 1; }
done
python3 - <<'PY' | tee gpurun_out/r06_fast_census.txt
import csv, glob, collections
names = {1: "staging (+ launch, cell set-up)", 2: "+ candidate test (8 px / lane, packed compass test, compaction)", 3: "+ arc values of the survivors", 0: "+ NMS and ordered emission = full kernel"}
rows = {}
for stop in (1, 2, 3, 0):
    f = glob.glob("gpurun_out/r06_fast/pmc_%d/**/*counter_collection.csv" % stop, recursive=True)[0]
    acc = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(f)):
        if "k_fast_cells<true" not in r["Kernel_Name"].replace("(bool)1", "true").replace("<1", "<true"): 
            if not ("k_fast_cells" in r["Kernel_Name"] and ("<true" in r["Kernel_Name"] or "(bool)1" in r["Kernel_Name"])): continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(len(disp), 1)
    g = glob.glob("gpurun_out/r06_fast/tr_%d/**/*kernel_stats.csv" % stop, recursive=True)[0]
    us = None
    for r in csv.DictReader(open(g)):
        if "k_fast_cells" in r["Name"] and ("<true" in r["Name"] or "(bool)1" in r["Name"]): us = float(r["AverageNs"]) / 1e3
    rows[stop] = ({k: v / n for k, v in acc.items()}, us, n)
print("k_fast_cells<true, 48>, whole-stage launch over 64 frames (52 160 cells, one wavefront each); wave instructions per launch, cumulative and per phase")
print("%-68s %12s %12s %12s %10s | %12s %12s %10s" % ("kernel leaves after", "VALU", "SALU", "LDS", "us", "d VALU", "d SALU", "d us"))
prev = ({}, 0.0)
for stop in (1, 2, 3, 0):
    c, us, n = rows[stop]
    print("%-68s %12.4g %12.4g %12.4g %10.1f | %12.4g %12.4g %10.1f   (%d launches)" % (names[stop], c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0), us or 0,
          c.get("SQ_INSTS_VALU", 0) - prev[0].get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0) - prev[0].get("SQ_INSTS_SALU", 0), (us or 0) - prev[1], n))
    prev = (c, us or 0)
PY
