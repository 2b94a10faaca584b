"""Kernel timeline of the last full ORB step in a rocprofv3 kernel trace (tools/trace_step.sh):  python3 tools/print_step_timeline.py DIR"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_orient_describe" in r["Kernel_Name"]]
a, b = idx[-3] + 1, idx[-2] + 1
t0 = int(rows[a]["Start_Timestamp"])
print("step span %.1f us (previous step's last kernel end -> this step's last kernel end)" % ((int(rows[b - 1]["End_Timestamp"]) - int(rows[idx[-3]]["End_Timestamp"])) / 1e3))
for r in rows[a:b]:
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    g = r.get("Grid_Size_X", "") or r.get("Grid_Size", "")
    print("  %-20s q%-3s grid %-8s %8.1f -> %8.1f  (%.1f)" % (m.group(1) if m else "?", r.get("Queue_Id", "?"), g, (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                      (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
