#!/bin/bash
# Round-2 evidence run (on the GPU box through gpurun): the rocprofv3 summary of the default bench command, the SQ counter
# passes behind the VALU roofline, and the FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic.
#   bash tools/prof_r02.sh   ->  gpurun_out/r02_*   (copy into profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rm -rf $O/r02_stats $O/r02_sq_a $O/r02_sq_b $O/r02_fetch $O/r02_write $O/r02_sq_h
# 1. per-kernel durations of the same command the driver runs
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_stats -o bench -- python3 bench.py --steps 20 --warmup 3 > $O/r02_bench_line_under_rocprof.json 2> $O/r02_stats.err
f=$(find $O/r02_stats -name "*kernel_stats.csv" | head -1)
cp "$f" $O/r02_kernel_stats_raw.csv
python3 tools/summarize_rocprof.py "$f" $O/r02_kernel_stats_bench.csv "python3 bench.py --steps 20 --warmup 3" | head -30
echo "--- bench line under rocprofv3"; tail -c 400 $O/r02_bench_line_under_rocprof.json; echo
# 2. SQ counters of the ORB kernels (two passes: 8 SQ slots each)
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d $O/r02_sq_a -o a -- $BENCH > $O/r02_sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d $O/r02_sq_b -o b -- $BENCH > $O/r02_sq_b.log 2>&1
# ... and of the two Hamming kernels
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/r02_sq_h -o h -- python3 tools/dbg_hamming.py > $O/r02_sq_h.log 2>&1
# 3. HBM-side traffic (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r02_fetch -o f -- $BENCH > $O/r02_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r02_write -o w -- $BENCH > $O/r02_write.log 2>&1
python3 tools/pmc_tables.py $O
