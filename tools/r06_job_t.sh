#!/bin/bash
# the LBA adapter's walk (slots, radix sort, pre-sized edge arrays): adapter tests, then the class surface timed
O=gpurun_out/r06t; mkdir -p $O
python3 -m pytest tests/test_gpu_adapters.py tests/test_integration_snippets.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
python3 - <<'P' > $O/cs.json 2> $O/cs.err
import sys, json; sys.path.insert(0, '.')
import bench
from eao_fusion_amd import synth
r = bench.measure_class_surface(synth)
print(json.dumps({k: r[k] for k in r if 'bundle' in k or 'error' in k}, indent=1))
P
cat $O/cs.json | cut -c1-200; tail -3 $O/cs.err
