"""Kernel timeline of tests/cpp/mixed_load under rocprofv3 --kernel-trace (+ --memory-copy-trace): per tracker-side kernel its duration distribution and the gap to
the previous tracker-side kernel, while the LM kernels (k_ba_*) are absent / present on the device.   python tools/analyze_mixed_trace.py <dir>"""
import csv
import glob
import re
import sys

import numpy as np

d = sys.argv[1]
kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(kf)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(k_\w+)", n)
    return m.group(1) if m else n[:32]


T_NAMES = ("k_track", "k_match", "k_pose", "k_resize", "k_pyramid", "k_fast", "k_quadtree", "k_blur", "k_orient", "k_stream", "k_undist")
lm = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if short(r["Kernel_Name"]).startswith(("k_ba_", "k_bal_"))]
lm_s = np.array([a for a, _ in lm]) if lm else np.zeros(0)
lm_e = np.array([b for _, b in lm]) if lm else np.zeros(0)


def lm_near(t0, t1, pad=300000):      # an LM kernel within 0.3 ms of [t0, t1]
    if not len(lm_s):
        return False
    i = np.searchsorted(lm_s, t1 + pad)
    return bool(i > 0 and (lm_e[:i] >= t0 - pad).any()) if i < 2000 else bool((lm_e[max(0, i - 2000):i] >= t0 - pad).any())


tk = [r for r in rows if short(r["Kernel_Name"]).startswith(T_NAMES)]
print("%d kernels, %d tracker-side, %d LM" % (len(rows), len(tk), len(lm)))
queues = {}
for r in tk:
    queues.setdefault(r.get("Queue_Id", "?"), 0)
    queues[r.get("Queue_Id", "?")] += 1
print("tracker-side kernels per queue:", queues)
lq = {}
for r in rows:
    if short(r["Kernel_Name"]).startswith(("k_ba_", "k_bal_")):
        lq[r.get("Queue_Id", "?")] = lq.get(r.get("Queue_Id", "?"), 0) + 1
print("LM kernels per queue:", lq)
stat = {}
prev_end = None
for r in tk:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    near = lm_near(s, e)
    st = stat.setdefault((n, near), {"dur": [], "gap": []})
    st["dur"].append((e - s) / 1e3)
    if prev_end is not None and s - prev_end < 2e6:
        st["gap"].append((s - prev_end) / 1e3)
    prev_end = e
print("%-28s %-6s %6s %9s %9s %9s | gap to the previous tracker-side kernel: %9s %9s %9s" % ("kernel", "LM", "n", "dur p50", "p99", "max", "p50", "p99", "max"))
for (n, near), st in sorted(stat.items()):
    du, g = np.array(st["dur"]), np.array(st["gap"] or [0])
    print("%-28s %-6s %6d %9.1f %9.1f %9.1f | %50.1f %9.1f %9.1f" % (n, "beside" if near else "idle", len(du), np.percentile(du, 50), np.percentile(du, 99), du.max(),
                                                                      np.percentile(g, 50), np.percentile(g, 99), g.max()))
mf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
if mf:
    mr = list(csv.DictReader(open(mf[0])))
    by = {}
    for r in mr:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = (r.get("Direction", "?"), "beside" if lm_near(s, e) else "idle")
        by.setdefault(k, []).append((e - s) / 1e3)
    for k, v in sorted(by.items()):
        v = np.array(v)
        print("copy %-28s %-6s n %6d dur p50 %8.1f p99 %8.1f max %8.1f us" % (k[0], k[1], len(v), np.percentile(v, 50), np.percentile(v, 99), v.max()))
