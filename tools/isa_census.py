#!/usr/bin/env python3
"""Static ISA mix per kernel of one HIP source (gfx950): how many VALU / SALU / LDS / VMEM instructions, and how many of the
VALU ones are 64-bit address arithmetic, integer multiplies, moves -- the census that found k_orient_describe's per-lane
pointer arithmetic.  usage: python tools/isa_census.py eao_fusion_amd/csrc/orb.hip [name-filter]
       python tools/isa_census.py --dpp-hazards eao_fusion_amd/csrc/gba.hip [name-filter]
--dpp-hazards: the hand-written 64-bit DPP instructions (inline assembly in csrc/gba.hip: v_fmac_f64_dpp / v_mov_b64_dpp with row_newbcast) need two wait states
between a VALU write of a VGPR and a DPP read of it; this walks every kernel's instruction stream and reports each DPP instruction whose broadcast source was
written by a VALU instruction fewer than two wait states earlier, or that follows a VALU write of EXEC (v_cmpx) by fewer than five (an `s_nop N` counts N + 1).
Exit code 1 if there is one."""
import collections
import re
import subprocess
import sys
import tempfile


def vregs(tok):
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def dpp_hazards(ins_lines):
    """ins_lines: the instruction lines of one kernel.  Returns (number of DPP instructions, list of violations)."""
    bad, n = [], 0
    for i, l in enumerate(ins_lines):
        op = l.split()[0]
        if not op.endswith("_dpp"):
            continue
        n += 1
        ops = l.split(None, 1)[1].split(",")
        src0 = vregs(ops[1].split()[0])
        ws, j = 0, i - 1
        while j >= 0 and ws < 5:
            p = ins_lines[j]
            if p.startswith("s_nop"):
                ws += int(p.split()[1]) + 1
            else:
                if ws < 2 and p.startswith("v_") and vregs(p.split(None, 1)[1].split(",")[0]) & src0:
                    bad.append((p, l))
                if p.startswith("v_cmpx") or (p.startswith("v_") and " exec" in p.split(None, 1)[1].split(",")[0]):      # a VALU write of EXEC: five wait states
                    bad.append((p, l))
                ws += 1
            j -= 1
    return n, bad


def main():
    hazards = len(sys.argv) > 1 and sys.argv[1] == "--dpp-hazards"
    if hazards:
        del sys.argv[1]
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
        if hazards:
            full = [l.strip() for l in lines[i + 1:ends[0]] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
            n, bad = dpp_hazards(full)
            if n:
                short = re.search(r"(k_[A-Za-z0-9_]+?)(?:I[A-Z]|E[A-Z]|$)", name)
                print("%-28s %5d DPP instructions, %d hazard violations" % ((short.group(1) if short else name)[:28], n, len(bad)))
                for p, l in bad[:5]:
                    print("   ", p, " ->", l)
                main.failed = main.failed or bool(bad)
            continue
        ins = [l.strip().split()[0] for l in lines[i + 1:ends[0]] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        cc = collections.Counter(ins)
        grp = lambda p: sum(v for k, v in cc.items() if k.startswith(p))
        short = re.search(r"(k_[A-Za-z0-9_]+?)(?:I[A-Z]|E[A-Z]|$)", name)
        print("%-28s total %5d  valu %5d (f64 %4d, u64 add %3d, mad_u64 %3d, mul_lo %3d, mov %3d)  salu %4d  lds %3d  vmem %3d  waitcnt %3d  branches %3d" % (
            (short.group(1) if short else name)[:28], len(ins), grp("v_"), sum(v for k, v in cc.items() if "_f64" in k), cc["v_lshl_add_u64"],
            cc["v_mad_u64_u32"], cc["v_mul_lo_u32"], cc["v_mov_b32_e32"], grp("s_") - cc["s_waitcnt"], grp("ds_"), grp("global_") + grp("buffer_") + grp("flat_"),
            cc["s_waitcnt"], grp("s_cbranch")))


main.failed = False
if __name__ == "__main__":
    main()
    sys.exit(1 if main.failed else 0)
