#!/bin/bash
# kernel + memory-copy timeline of the streaming host API (12 batches through three slots)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tr_stream
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tr_stream -o t -- python3 tools/dbg_stream_trace.py > gpurun_out/tr_stream.log 2>&1
python3 - <<'PY'
import csv, glob
kt = glob.glob("gpurun_out/tr_stream/**/*kernel_trace.csv", recursive=True)[0]
mc = glob.glob("gpurun_out/tr_stream/**/*memory_copy_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-40:]))
for r in csv.DictReader(open(mc)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
big = [e for e in ev if e[2].startswith("C") and (e[1] - e[0]) > 100000]
t0 = big[4][0]
with open("gpurun_out/r03_stream_timeline.txt", "w") as f:
    f.write("# kernel + copy timeline of eao_orb_stream_* (three slots, 64 frames each), microseconds from the fifth upload\n")
    for s, e, n in ev:
        if t0 - 50000 <= s <= t0 + 1500000:
            f.write("%9.1f %9.1f %7.1f  %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
print(open("gpurun_out/r03_stream_timeline.txt").read()[:6000])
PY
