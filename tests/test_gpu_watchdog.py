"""The GPU tier's watchdog and suite budget (tests/conftest.py): a hung native call inside a gpu-marked test must end the run with that
test's NAME within the limit under the driver's plain `pytest -m gpu -x -q` (no --timeout), and a suite that outlives its budget must skip
what is left by name instead of being killed at the driver's limit.  Runs on the CPU: the child blocks in pause(), not in a kernel."""
import os
import subprocess
import sys
import time

import snk_testlib as T

CHILD = os.path.join(T.ROOT, "tests", "watchdog_child.py")


def run_child(**env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    t = time.monotonic()
    r = subprocess.run([sys.executable, "-m", "pytest", CHILD, "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=120, env=e, cwd=T.ROOT)
    return r, time.monotonic() - t


def test_hung_gpu_test_is_named_within_the_limit():
    r, dt = run_child(SNK_WD_CHILD_HANG=1, SNK_GPU_TEST_TIMEOUT_S=3)
    out = r.stdout + r.stderr
    assert r.returncode == 3, out[-2000:]
    assert dt < 60
    assert "SNK GPU WATCHDOG: tests/watchdog_child.py::test_native_call_that_never_returns did not finish within 3 s" in out, out[-2000:]
    assert "before it: 1 passed, 0 failed" in out
    assert "FAILED tests/watchdog_child.py::test_native_call_that_never_returns - watchdog" in out
    assert "watchdog_child.py\", line" in r.stderr and "in test_native_call_that_never_returns" in r.stderr   # the stack dump names the blocked frame


def test_watchdog_is_silent_on_a_green_run_and_durations_are_printed():
    r, _ = run_child(SNK_GPU_TEST_TIMEOUT_S=3)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "3 passed" in r.stdout and "WATCHDOG" not in r.stdout + r.stderr
    assert "slowest 15 durations" in r.stdout


def test_suite_budget_skips_the_rest_by_name():
    r, _ = run_child(SNK_GPU_SUITE_BUDGET_S=0.2, SNK_WD_CHILD_SLEEP=0.5)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "1 passed, 2 skipped" in r.stdout, r.stdout[-2000:]
