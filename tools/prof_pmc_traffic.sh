#!/bin/bash
# HBM-side traffic of the ORB kernels: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over the bench command
# (run on the GPU box through gpurun) -> gpurun_out/pmc_traffic.json + pmc_traffic_raw.json (copy into profiles/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/pmc_write.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
def load(tag, counter):
    f = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    return {k: (acc[k] / len(disp[k]), len(disp[k])) for k in acc}
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
raw = {}
for k in sorted(set(fe) | set(wr)):
    if not k.startswith("k_"): continue
    raw[k] = {"fetch_KiB_per_launch": fe.get(k, (0, 0))[0], "launches_fetch": fe.get(k, (0, 0))[1],
              "write_KiB_per_launch": wr.get(k, (0, 0))[0], "launches_write": wr.get(k, (0, 0))[1]}
json.dump(raw, open("gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
stage = {"pyramid": "k_resize", "fast": "k_fast_cells", "blur": "k_blur7", "quadtree": "k_quadtree", "orient_describe": "k_orient_describe"}
# launches of the kernel per bench step (profiled + timed steps of the run: 1 warmup + 3 steps + 3 profiled calls ...): derive from blur (1 per step)
steps = raw["k_blur7"]["launches_fetch"]
out = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py --steps 3 --warmup 1 "
                "--no-cpu-baseline --no-extra; bytes = KiB counter * 1024.  RAW counters: HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE "
                "(MI355X_MICROARCH.md; the factor holds for 1-, 4- and 12-byte loads per lane as well: profiles/r05_fetch_calib.txt).",
       "batch": int(__import__("os").environ.get("EAO_PMC_BATCH", "64")), "kernels": {}}
for st, k in stage.items():
    r = raw[k]
    f, w = r["fetch_KiB_per_launch"] * 1024, r["write_KiB_per_launch"] * 1024
    nl = r["launches_fetch"]
    out["kernels"][st] = {"kernel": k, "launches_per_step": round(nl / steps, 2), "fetch_bytes_per_step": int(f * nl / steps),
                          "write_bytes_per_step": int(w * r["launches_write"] / steps), "hbm_bytes_per_step_corrected": int((2 * f * nl + w * r["launches_write"]) / steps)}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
