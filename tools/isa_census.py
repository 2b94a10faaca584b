#!/usr/bin/env python3
"""Static ISA mix per kernel of one HIP source (gfx950): how many VALU / SALU / LDS / VMEM instructions, and how many of the
VALU ones are 64-bit address arithmetic, integer multiplies, moves -- the census that found k_orient_describe's per-lane
pointer arithmetic.  usage: python tools/isa_census.py eao_fusion_amd/csrc/orb.hip [name-filter]"""
import collections
import re
import subprocess
import sys
import tempfile


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only",
                               "-S", src, "-o", f.name], stderr=subprocess.DEVNULL)
        lines = open(f.name).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_Z") and ": " in l]
    for i, name in starts:
        ends = [k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end")]
        if not ends or flt not in name:
            continue
        ins = [l.strip().split()[0] for l in lines[i + 1:ends[0]] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        cc = collections.Counter(ins)
        grp = lambda p: sum(v for k, v in cc.items() if k.startswith(p))
        short = re.search(r"(k_[A-Za-z0-9_]+?)(?:I[A-Z]|E[A-Z]|$)", name)
        print("%-28s total %5d  valu %5d (f64 %4d, u64 add %3d, mad_u64 %3d, mul_lo %3d, mov %3d)  salu %4d  lds %3d  vmem %3d  waitcnt %3d  branches %3d" % (
            (short.group(1) if short else name)[:28], len(ins), grp("v_"), sum(v for k, v in cc.items() if "_f64" in k), cc["v_lshl_add_u64"],
            cc["v_mad_u64_u32"], cc["v_mul_lo_u32"], cc["v_mov_b32_e32"], grp("s_") - cc["s_waitcnt"], grp("ds_"), grp("global_") + grp("buffer_") + grp("flat_"),
            cc["s_waitcnt"], grp("s_cbranch")))


if __name__ == "__main__":
    main()
