#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per byte streamed, by load width (tools/ubench/fetch_calib.hip) -> gpurun_out/r05_fetch_calib.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; rm -rf $O/fc_f $O/fc_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fc_f -o f -- ./tools/ubench/fetch_calib > $O/fc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/fc_w -o w -- ./tools/ubench/fetch_calib > $O/fc_w.log 2>&1
python3 - <<'PY' | tee gpurun_out/r05_fetch_calib.txt
import csv, re
B = 768 << 20
exp = {"k_read<unsigned": B, "k_read<uint2": B, "k_read<u3": B // 12 * 12, "k_read<uint4": B, "k_read_u8": B // 4, "k_write<unsigned": B, "k_write<uint4": B}
print("# tools/ubench/fetch_calib.hip on MI355X: counter (KiB x 1024) / bytes the kernel streams once over a 768 MB buffer (separate --pmc passes)")
print("%-34s %-12s %14s %14s %8s" % ("kernel", "counter", "counter bytes", "streamed", "ratio"))
for tag, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for r in csv.DictReader(open("gpurun_out/fc_%s/%s_counter_collection.csv" % (tag, tag))):
        name = r["Kernel_Name"]
        for k, e in exp.items():
            if k in name.replace("(anonymous namespace)::", ""):
                v = float(r["Counter_Value"]) * 1024
                print("%-34s %-12s %14.0f %14d %8.3f" % (k + ">" * ("<" in k), ctr, v, e, v / e))
PY
