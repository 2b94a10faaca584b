#!/usr/bin/env python3
"""Condense the counter passes of tools/prof_r02.sh into the files committed under profiles/:
   r02_pmc_sq.json (+ .txt table)   SQ counters per launch of every ORB kernel (64-frame batch) and of the Hamming kernels
   r02_pmc_traffic.json             FETCH_SIZE / WRITE_SIZE per bench step and stage
usage: pmc_tables.py <gpurun_out dir>"""
import collections
import csv
import glob
import json
import os
import re
import sys

O = sys.argv[1]
R = sys.argv[2] if len(sys.argv) > 2 else "r02"        # round tag of the passes and of the files written


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][-40:]


def load(tag):
    fs = glob.glob(os.path.join(O, tag, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: dict({c: v / len(disp[k]) for c, v in acc[k].items()}, launches=len(disp[k])) for k in acc}


def sha16(rel):
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest()[:16]


SRC = {rel: sha16(rel) for rel in ("eao_fusion_amd/csrc/orb.hip", "eao_fusion_amd/csrc/orb_internal.h", "eao_fusion_amd/csrc/hamming.hip", "eao_fusion_amd/csrc/lm_internal.h", "eao_fusion_amd/csrc/lba.hip", "eao_fusion_amd/csrc/gba.hip", "eao_fusion_amd/csrc/lm_host.hip")}      # what bench.py checks before it derives anything
a, b, h = load(R + "_sq_a"), load(R + "_sq_b"), load(R + "_sq_h")
fe, wr = load(R + "_fetch"), load(R + "_write")
batch = int(os.environ.get("EAO_PMC_BATCH", "64"))
steps = a.get("k_blur7", {}).get("launches", 1)         # one blur launch per bench step (timed and profiled alike)
sq = {"_note": "rocprofv3 --kernel-trace --pmc <8 SQ counters> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra (two passes); "
               "values are PER LAUNCH (sum over the chip).  SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles.  k_fast_cells<true, 48> "
               "is the whole-stage launch of the profiled steps (the <false, 48> rows are the shares of the overlapped schedule).",
      "batch": batch, "source_sha16": SRC, "kernels": {}, "kernels_extra": {}}
stage = {"k_resize": "k_resize", "k_fast_cells": "k_fast_cells<true, 48>", "k_quadtree": "k_quadtree", "k_blur7": "k_blur7", "k_orient_describe": "k_orient_describe"}
lines = []
for name, row in stage.items():
    pick = lambda t: (t.get(row) or next((v for k, v in t.items() if k.startswith(name + "<true")), None) or t.get(name)
                      or next((v for k, v in t.items() if k.startswith(name + "<")), None))      # (k_quadtree<256>: a template since round 2)
    ra, rb = pick(a), pick(b)
    if not ra:
        continue
    d = {k: v for k, v in ra.items() if k != "launches"}
    if rb:
        d.update({k: v for k, v in rb.items() if k != "launches"})
    # launches per bench step (k_fast_cells<true>: one per PROFILED step, which is where its duration is measured)
    d["launches_per_step"] = 1 if name == "k_fast_cells" else round(ra["launches"] / steps, 2)
    sq["kernels"][name] = d
for k, v in h.items():
    if k.startswith("k_hamming"):
        sq["kernels_extra"][k.split("<")[0]] = {c: x for c, x in v.items()}
json.dump(sq, open(os.path.join(O, R + "_pmc_sq.json"), "w"), indent=1)
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
with open(os.path.join(O, R + "_pmc_sq.txt"), "w") as f:
    f.write("# SQ counters per launch, 64-frame batch (tools/prof_round.sh)\n")
    f.write("kernel".ljust(22) + "".join(c.replace("SQ_", "").rjust(17) for c in cols) + "\n")
    for k, d in list(sq["kernels"].items()) + list(sq["kernels_extra"].items()):
        f.write(k[:22].ljust(22) + "".join(("%.4g" % d[c]).rjust(17) if c in d else "-".rjust(17) for c in cols) + "\n")
print(open(os.path.join(O, R + "_pmc_sq.txt")).read())
# traffic
tr = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
               "--no-extra; bytes = KiB counter x 1024, RAW counters.  MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half the bytes of a streaming read on gfx950 "
               "-- bench.py doubles it; self-check: k_blur7 reads 22/16 x 60.8 MB = 83.6 MB (16-row strips + 6 halo rows), 2 x FETCH_SIZE counted below.",
      "batch": batch, "source_sha16": SRC, "kernels": {}}
names = {"pyramid": "k_resize", "fast": "k_fast_cells", "blur": "k_blur7", "quadtree": "k_quadtree", "orient_describe": "k_orient_describe"}
for st, kn in names.items():
    f_tot = sum(v["FETCH_SIZE"] * v["launches"] for k, v in fe.items() if k.startswith(kn) and "FETCH_SIZE" in v) * 1024
    w_tot = sum(v["WRITE_SIZE"] * v["launches"] for k, v in wr.items() if k.startswith(kn) and "WRITE_SIZE" in v) * 1024
    nl = sum(v["launches"] for k, v in fe.items() if k.startswith(kn))
    tr["kernels"][st] = {"kernel": kn, "launches_per_step": round(nl / steps, 2), "fetch_bytes_per_step": int(f_tot / steps), "write_bytes_per_step": int(w_tot / steps),
                         "hbm_bytes_per_step_corrected": int((2 * f_tot + w_tot) / steps)}
json.dump(tr, open(os.path.join(O, R + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(tr["kernels"], indent=1))

# ---- the BA half (round 4): FETCH_SIZE / WRITE_SIZE of every k_ba_* launch of the 25-window batch (one group) and of single windows, per LM iteration
ba = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- EAO_BA_BATCH_GROUPS=1 python3 tools/dbg_ba_batch.py (25 windows per call) "
               "and python3 tools/dbg_ba_cabi.py (one window per call); bytes = KiB counters x 1024; hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: FETCH_SIZE "
               "reports half of a streaming read on gfx950); per LM iteration = the sum over ALL k_ba_* launches of the run (set-up, error passes and the result kernel "
               "included) / the number of k_ba_backsub launches (one per LM trial).",
      "source_sha16": SRC}
for key, tagf, tagw in (("batched", R + "_ba_fetch", R + "_ba_write"), ("single_window", R + "_ba1_fetch", R + "_ba1_write")):
    fe2, wr2 = load(tagf), load(tagw)
    if not fe2 or not wr2:
        continue
    trials = sum(v["launches"] for k, v in fe2.items() if k.startswith("k_ba_backsub"))
    ker = {}
    tot_f = tot_w = 0.0
    for k in sorted(set(fe2) | set(wr2)):
        if not k.startswith("k_ba_"):
            continue
        f_ = fe2.get(k, {}).get("FETCH_SIZE", 0.0) * 1024
        w_ = wr2.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
        n_ = fe2.get(k, {}).get("launches", 0)
        ker[k] = {"launches": n_, "fetch_bytes_per_launch": int(f_), "write_bytes_per_launch": int(w_), "hbm_bytes_per_launch": int(2 * f_ + w_)}
        tot_f += f_ * n_
        tot_w += w_ * wr2.get(k, {}).get("launches", n_)
    ba[key] = {"lm_trials": trials, "fetch_bytes_per_iteration": int(tot_f / max(trials, 1)), "write_bytes_per_iteration": int(tot_w / max(trials, 1)),
               "hbm_bytes_per_iteration": int((2 * tot_f + tot_w) / max(trials, 1)), "kernels": ker}
if len(ba) > 2:
    json.dump(ba, open(os.path.join(O, R + "_ba_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: {q: v[q] for q in v if q != "kernels"} for k, v in ba.items() if isinstance(v, dict) and "lm_trials" in v}, indent=1))
