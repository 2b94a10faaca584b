#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into the per-kernel table committed under profiles/.
usage: summarize_rocprof.py <kernel_stats.csv> <out.csv> "<command that was profiled>" """
import csv
import re
import sys


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+(?:<[A-Za-z0-9_, ]+>)?)", name)
    return m.group(1) if m else name.split("(")[0][-60:]


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = list(csv.DictReader(open(src)))
    out = ["# rocprofv3 --kernel-trace --stats -- %s   (1x MI355X)" % cmd, "# durations in microseconds",
           "kernel,calls,avg_us,min_us,max_us,total_ms,pct"]
    for r in rows:
        out.append("%s,%s,%.2f,%.2f,%.2f,%.3f,%s" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                     float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:24]))


if __name__ == "__main__":
    main()
