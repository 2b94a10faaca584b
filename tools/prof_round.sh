#!/bin/bash
# Evidence run of a round (on the GPU box through gpurun): the rocprofv3 summary of the default bench command, the SQ counter passes
# behind the VALU roofline, the FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic, the per-kernel summaries of the BA calls behind
# roofline.ba, and the VALU issue-rate micro-benchmark.
#   bash tools/prof_round.sh r03   ->  gpurun_out/r03_*   (copy the summaries into profiles/ afterwards: tools/prof_round.sh does not)
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rm -rf $O/${R}_stats $O/${R}_sq_a $O/${R}_sq_b $O/${R}_fetch $O/${R}_write $O/${R}_sq_h $O/${R}_ba_batch $O/${R}_ba_single
# 1. per-kernel durations of the same command the driver runs
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_stats -o bench -- python3 bench.py --steps 20 --warmup 3 > $O/${R}_bench_line_under_rocprof.json 2> $O/${R}_stats.err
f=$(find $O/${R}_stats -name "*kernel_stats.csv" | head -1)
cp "$f" $O/${R}_kernel_stats_raw.csv
python3 tools/summarize_rocprof.py "$f" $O/${R}_kernel_stats_bench.csv "python3 bench.py --steps 20 --warmup 3" | head -30
echo "--- bench line under rocprofv3"; tail -c 400 $O/${R}_bench_line_under_rocprof.json; echo
# 2. SQ counters of the ORB kernels (two passes: 8 SQ slots each)
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d $O/${R}_sq_a -o a -- $BENCH > $O/${R}_sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d $O/${R}_sq_b -o b -- $BENCH > $O/${R}_sq_b.log 2>&1
# ... and of the two Hamming kernels
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/${R}_sq_h -o h -- python3 tools/dbg_hamming.py > $O/${R}_sq_h.log 2>&1
# 3. HBM-side traffic (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${R}_fetch -o f -- $BENCH > $O/${R}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${R}_write -o w -- $BENCH > $O/${R}_write.log 2>&1
# 3b. ... and of the LM launches (round 4: roofline.ba.*.traffic): the 25-window batch in one group, and single windows
rm -rf $O/${R}_ba_fetch $O/${R}_ba_write $O/${R}_ba1_fetch $O/${R}_ba1_write
EAO_BA_BATCH_GROUPS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${R}_ba_fetch -o f -- python3 tools/dbg_ba_batch.py > $O/${R}_ba_fetch.log 2>&1
EAO_BA_BATCH_GROUPS=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${R}_ba_write -o w -- python3 tools/dbg_ba_batch.py > $O/${R}_ba_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${R}_ba1_fetch -o f -- python3 tools/dbg_ba_cabi.py > $O/${R}_ba1_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${R}_ba1_write -o w -- python3 tools/dbg_ba_cabi.py > $O/${R}_ba1_write.log 2>&1
python3 tools/pmc_tables.py $O $R
# 4. the BA half: per-kernel durations of the 25-window batch and of the single window (the figures roofline.ba quotes)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_ba_batch -o s -- python3 tools/dbg_ba_batch.py > $O/${R}_ba_batch.log 2>&1
f=$(find $O/${R}_ba_batch -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" $O/${R}_ba_batch_kernel_stats.csv "python3 tools/dbg_ba_batch.py (25 windows per call, 23 calls)" | head -12
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_ba_single -o s -- python3 tools/dbg_ba_cabi.py > $O/${R}_ba_single.log 2>&1
f=$(find $O/${R}_ba_single -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" $O/${R}_ba_single_kernel_stats.csv "python3 tools/dbg_ba_cabi.py (one window per call, 65 calls)" | head -12
# 5. VALU issue rates at 1 / 2 / 4 / 8 waves per SIMD (bench.py's VALU_PEAK_WAVE_INSTS) and the fp64 pipes of one CU
./tools/ubench/intops_chip > $O/${R}_ubench_intops.txt 2>&1; head -8 $O/${R}_ubench_intops.txt
./tools/ubench/f64_simd > $O/${R}_ubench_f64.txt 2>&1
./tools/ubench/lds_ops > $O/${R}_ubench_lds.txt 2>&1
# 6. the default bench line itself
python3 bench.py > $O/${R}_bench_line.json 2> $O/${R}_bench_line.err; tail -c 600 $O/${R}_bench_line.json; echo
# 7. round-5 extras: pose / map-scale stamps and kernel summaries, FETCH_SIZE calibration by load width
bash tools/prof_r05_pose_gba.sh > $O/${R}_extras.log 2>&1; tail -6 $O/${R}_extras.log | cut -c1-250
bash tools/prof_fetch_calib.sh > /dev/null 2>&1; cat $O/r05_fetch_calib.txt
