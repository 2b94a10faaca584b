#!/bin/bash
# Evidence run of a round (on the GPU box through gpurun): the rocprofv3 summary of the default bench command, the SQ counter passes
# behind the VALU roofline, the FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic, the per-kernel summaries of the BA calls behind
# roofline.ba, and the VALU issue-rate micro-benchmark.
#   bash tools/prof_round.sh r03   ->  gpurun_out/r03_*   (copy the summaries into profiles/ afterwards: tools/prof_round.sh does not)
R=${1:-r06}
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
# 7. the map-scale BundleAdjustment benchmarks: host-phase stamps, the plan line, rocprofv3 kernel summaries (fresh directories: this step removes nothing)
G=$O/${R}_gba_$$; GB=$O/${R}_gbab_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $G -o s -- python3 tools/dbg_gba.py > $O/${R}_gba.log 2>&1
f=$(find $G -name "*kernel_stats.csv" | head -1); python3 tools/summarize_rocprof.py "$f" $O/${R}_gba_kernel_stats.csv "python3 tools/dbg_gba.py (200 KF x 20000 MP, 3 calls)" | head -14
EAO_DBG_ORACLE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GB -o s -- python3 tools/dbg_gba_banded.py > $O/${R}_gbab.log 2>&1
f=$(find $GB -name "*kernel_stats.csv" | head -1); python3 tools/summarize_rocprof.py "$f" $O/${R}_gba_banded_kernel_stats.csv "EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py (1000 KF x 50000 MP band 11, 3 calls)" | head -14
EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py 2>&1 | grep -E 'map-scale plan|host set-up|map-scale wall|banded GBA' | cut -c1-420 > $O/${R}_gba_banded_host_stamps.txt
EAO_BA_ND=1 EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba_banded.py 2>&1 | grep -E 'map-scale plan|map-scale wall|banded GBA' | cut -c1-420 > $O/${R}_gba_banded_natural_order.txt
EAO_DEBUG_STAMPS=1 EAO_DBG_ORACLE=0 python3 tools/dbg_gba.py 2>&1 | grep -E 'map-scale plan|host set-up|map-scale wall|^GBA' | cut -c1-420 > $O/${R}_gba_host_stamps.txt
tail -n 3 $O/${R}_gba_host_stamps.txt $O/${R}_gba_banded_host_stamps.txt | cut -c1-300
# 8. the GPU suite's log
python3 -m pytest tests -m gpu -q > $O/${R}_gputests.log 2>&1; tail -3 $O/${R}_gputests.log
