"""Summarise a rocprofv3 kernel trace of tools/dbg_ba.py: per-kernel totals of the LAST LocalBundleAdjustment call and
every idle gap above 3 us on its stream.  usage: python tools/ba_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'k_\w+', n)
    return m.group(0) if m else n[:28]
idx = [i for i, r in enumerate(rows) if 'k_ba_errors' in r['Kernel_Name']]
start = idx[-2] if len(idx) >= 2 else 0          # two k_ba_errors per call
# include the copies just before
while start > 0 and 'k_ba' not in rows[start - 1]['Kernel_Name']: start -= 1
call = rows[start:]
t0 = int(call[0]['Start_Timestamp'])
tot = {}
prev = None
for r in call:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = short(r['Kernel_Name'])
    tot.setdefault(n, [0, 0.0]); tot[n][0] += 1; tot[n][1] += (e - s) / 1e3
    if prev is not None and s - prev > 3000:
        print(f"  gap {(s - prev) / 1e3:7.1f} us before {n} at {(s - t0) / 1e3:8.1f}")
    prev = e
span = (int(call[-1]['End_Timestamp']) - t0) / 1e3
busy = sum(v[1] for v in tot.values())
print(f"span {span:.1f} us, busy {busy:.1f} us, {len(call)} dispatches")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:24s} x{v[0]:3d}  {v[1]:8.1f} us  avg {v[1] / v[0]:6.1f}")
