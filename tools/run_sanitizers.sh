#!/bin/bash
# CPU-side sanitizer pass (VERDICT r2 next #2e): the oracle (oracle/*.cpp) rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer
# and driven by the whole non-GPU suite (golden vectors, oracle self-checks, numpy re-derivations, 240 quad-tree replays ...), and
# the C++ adapters + every INTEGRATION.md snippet compiled with the same flags and -Werror=... off (compile + link only: running
# them needs the GPU, and GPU / host-in-GPU-process sanitizer runs are not available on this pool).
# Round 4: the HOST half of the product's guided searches (eao_fusion_amd/csrc/search.hip: per-query geometry + the selection loops of all eleven searches, driven
# by caller-supplied indices) is part of the pass -- tests/test_host_replay_cpu.py builds it as plain C++ with the same flags (EAO_HOST_SAN=1) against the oracle's
# candidate lists and runs the parity and the malformed-input cases under ASan + UBSan.
# Usage: bash tools/run_sanitizers.sh   [round]  -> profiles/<round>_sanitizers_cpu.txt (default r06)
set -u
cd "$(dirname "$0")/.."
OUT=profiles/${1:-r06}_sanitizers_cpu.txt
TMP=$(mktemp -d)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
{
echo "== $(date -u +%Y-%m-%dT%H:%MZ)  g++ $(g++ -dumpversion), flags: $SAN"
echo "== 1. oracle/*.cpp with ASan + UBSan"
g++ $SAN -ffp-contract=off -std=c++17 -fPIC -Wall -Wextra -pthread -shared -o $TMP/liboracle_san.so oracle/orb_cpu.cpp oracle/hamming_cpu.cpp oracle/lm_cpu.cpp \
    oracle/match_cpu.cpp oracle/search_cpu.cpp oracle/frame_cpu.cpp 2>&1 || { echo "BUILD FAILED"; exit 1; }
echo "built $TMP/liboracle_san.so"
echo "== 2. python -m pytest tests -m 'not gpu' against the instrumented oracle (halt_on_error: any report aborts the run)"
# (-O1 without -march=native: integer tables must still match the golden vectors bit for bit, fp64 LM results within 1e-9)
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
EAO_HOST_SAN=1 EAO_ORACLE_LIB=$TMP/liboracle_san.so python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
echo "pytest exit code: ${PIPESTATUS[0]}"
echo "== 3. adapters + INTEGRATION.md snippets compiled with the sanitizer flags (compile + link)"
python - <<PY
import os, subprocess, sys
sys.path.insert(0, "tests")
import test_integration_snippets as T
tree = "$TMP/checkout"
found = T.extract(tree)
units = sorted(os.path.join(tree, p) for p in found if p.endswith(".cc"))
inc = ["-I", os.path.join(tree, "include"), "-I", os.path.join(tree, "src"), "-I", "tests/cpp/integration/ref", "-I", "include"]
link = ["-L", "eao_fusion_amd", "-leaofusion_hip", "-Wl,-rpath,/opt/rocm/lib", "-pthread"]
san = "$SAN".split()
jobs = [("integration_snippets_test", ["tests/cpp/integration_snippets_test.cpp"] + units, inc)]
for t in ("adapter_test", "search_adapter_test", "frame_adapter_test"):      # (tracker_adapter_test.cpp needs hipcc: device buffers)
    jobs.append((t, ["tests/cpp/%s.cpp" % t], ["-I", "include"]))
bad = 0
for name, srcs, incs in jobs:
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Wno-unused-function", "-DEAOFUSION_FORCE_CV_COMPAT"] + san + incs + srcs + ["-o", "$TMP/" + name] + link,
                       capture_output=True, text=True)
    warn = [l for l in r.stderr.split("\n") if "warning:" in l or "error:" in l]
    print("%-28s rc %d, %d diagnostics" % (name, r.returncode, len(warn)))
    for l in warn[:20]:
        print("   ", l)
    bad += r.returncode != 0
sys.exit(bad)
PY
echo "compile exit code: $?"
echo "== 4. the LocalBundleAdjustment adapter's walk RUN under ASan + UBSan (tests/cpp/adapter_bench.cpp lba-walk: the library call replaced by an identity result -- no GPU needed), unedited MapPoint and the row-2c accessors"
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
from eao_fusion_amd import synth
bench.class_surface_problem("$TMP/problem.bin", synth)
PY
for V in "" "-DEAO_BENCH_EDITED_MAPPOINT"; do
  g++ -std=c++17 -DEAOFUSION_FORCE_CV_COMPAT $V $SAN -I include tests/cpp/adapter_bench.cpp -o $TMP/adapter_bench_san -L eao_fusion_amd -leaofusion_hip -Wl,-rpath,$PWD/eao_fusion_amd -Wl,-rpath,/opt/rocm/lib -pthread 2>&1 | grep -E "error|warning" | head -5
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $TMP/adapter_bench_san $TMP/problem.bin lba-walk | grep -o '"call_ms": [0-9.]*' | head -1
  echo "adapter_bench lba-walk ${V:-(unedited MapPoint)} exit code: ${PIPESTATUS[0]}"
done
echo "== 5. the host crew (eao_fusion_amd/csrc/host_crew.h: batch runs + polled sessions) under ThreadSanitizer (tests/cpp/host_crew_test.cpp)"
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread tests/cpp/host_crew_test.cpp -o $TMP/host_crew_tsan && TSAN_OPTIONS=halt_on_error=1 $TMP/host_crew_tsan
echo "host_crew_test (TSan) exit code: $?"
} 2>&1 | tee $OUT
rm -rf $TMP
