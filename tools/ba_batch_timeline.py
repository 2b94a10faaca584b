"""Per-stream picture of the LAST eao_local_ba_batch call in a rocprofv3 kernel trace of tools/dbg_ba_batch.py: for every hardware
queue the kernels it ran, their summed duration, the idle time between them and the span.   python tools/ba_batch_timeline.py <kernel_trace.csv>"""
import csv, re, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'k_\w+', n)
    return m.group(0) if m else n[:24]
# the last call: everything after the last-but-one call's final k_ba_finish (G finishes per call, G = argv[2], default 4 groups)
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fin = [i for i, r in enumerate(rows) if 'k_ba_finish' in r['Kernel_Name']]
ends = sorted(fin, key=lambda i: int(rows[i]['End_Timestamp']))
prev_end = int(rows[ends[-G - 1]]['End_Timestamp']) if len(ends) > G else 0
call = [r for r in rows if int(r['Start_Timestamp']) >= prev_end and 'k_ba' in r['Kernel_Name'] or (int(r['Start_Timestamp']) >= prev_end and 'copy' in r['Kernel_Name'].lower())]
t0 = min(int(r['Start_Timestamp']) for r in call)
tend = max(int(r['End_Timestamp']) for r in call)
print("call span %.1f us, %d dispatches" % ((tend - t0) / 1e3, len(call)))
byq = collections.defaultdict(list)
for r in call: byq[r['Queue_Id']].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e3
    first, lastt = int(rs[0]['Start_Timestamp']) - t0, int(rs[-1]['End_Timestamp']) - t0
    gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rs, rs[1:])]
    big = sorted(gaps)[-5:]
    print("queue %s: %3d kernels, %.1f -> %.1f us, busy %.1f, idle between kernels %.1f (median gap %.2f, five largest %s)" % (
        q, len(rs), first / 1e3, lastt / 1e3, busy, sum(g for g in gaps if g > 0), sorted(gaps)[len(gaps) // 2] if gaps else 0, ["%.1f" % g for g in big]))
    tot = collections.OrderedDict()
    for r in rs:
        n = short(r['Kernel_Name']); d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot.setdefault(n, [0, 0.0]); tot[n][0] += 1; tot[n][1] += d
    print("    " + ", ".join("%s x%d %.0f (avg %.1f)" % (n, v[0], v[1], v[1] / v[0]) for n, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:6]))
