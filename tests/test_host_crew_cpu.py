"""The host crew's two modes (condition-variable runs for eao_local_ba_batch, polled sessions for the set-up of a map-scale BundleAdjustment) as plain C++ under
ThreadSanitizer: eao_fusion_amd/csrc/host_crew.h has no HIP in it, tests/cpp/host_crew_test.cpp drives it -- every chunk of every pass exactly once, a pass never
returns before its chunks, two callers competing for the crew both finish -- and TSan must not report a race in the hand-over protocol."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_crew_sessions_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "host_crew_test")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", os.path.join(ROOT, "tests", "cpp", "host_crew_test.cpp"), "-o", exe],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS")}      # (tools/run_sanitizers.sh runs the suite under a preloaded ASan runtime: not in a TSan process)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(env, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1"))
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-4000:]
    assert "every chunk once" in run.stdout and "ThreadSanitizer" not in run.stderr
