#!/bin/bash
# the round's closing capture: the GPU suite and the default bench line with the committed counter files (everything else: tools/prof_round.sh r06)
O=gpurun_out
python3 -m pytest tests -m gpu -q > $O/r06_gputests.log 2>&1; tail -n 3 $O/r06_gputests.log
python3 bench.py > $O/r06_bench_line.json 2> $O/r06_bench_line.err; tail -c 300 $O/r06_bench_line.json; echo
