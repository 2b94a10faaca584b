#!/bin/bash
# Regenerates the committed rocprofv3 summary of the default bench command (run on the GPU box through gpurun):
#   bash tools/prof_bench.sh   ->  gpurun_out/prof_final/*  (copy the summaries into profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python3 bench.py --steps 20 --warmup 3 > gpurun_out/prof_final_bench.json 2> gpurun_out/prof_final.err
f=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/prof_final_kernel_stats_raw.csv
python3 tools/summarize_rocprof.py "$f" gpurun_out/prof_final_kernel_stats.csv "python3 bench.py --steps 20 --warmup 3" | head -40
tail -c 600 gpurun_out/prof_final_bench.json
