import faulthandler
import os
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "first_contact: GPU test of code that has not run on hardware yet -- runs by default, sorted last")
    _wd["config"] = config
    if "gpu" in (config.getoption("markexpr", "") or "") and "not gpu" not in config.getoption("markexpr", "") \
            and getattr(config.option, "durations", None) is None:
        config.option.durations = 15                     # a GPU run always says where its time went


# ---- the GPU tier's watchdog (VERDICT r5 "Next round" 2a).  The driver runs plain `pytest tests -m gpu -x -q` under ONE wall-clock limit
# and passes no --timeout: a kernel that never returns used to end as "killed at the limit" with no test named.  Every gpu-marked test now runs
# under a timer of its own (SNK_GPU_TEST_TIMEOUT_S, default 180; 0 = off): when it fires the run says WHICH test hung, dumps every thread's
# Python stack, prints the tally so far and leaves with exit code 3 (a hung hipDeviceSynchronize cannot be unwound, so the process goes).
# And the suite as a whole keeps to SNK_GPU_SUITE_BUDGET_S (default 1050 s; the driver's limit was 1200 s in round 3): what is left when the
# budget is spent is SKIPPED with that reason -- first_contact tests sort last, so the tests that have been green before are the ones that ran.
_WD_TEST_S = float(os.environ.get("SNK_GPU_TEST_TIMEOUT_S", "180"))
_WD_SUITE_S = float(os.environ.get("SNK_GPU_SUITE_BUDGET_S", "1050"))
_wd = {"t0": None, "passed": 0, "failed": 0, "skipped": 0, "slow": [], "config": None}


def _wd_fire(nodeid, limit):
    try:                                                  # the capture manager owns fds 1/2 while a test runs: hand them back first
        capman = _wd["config"].pluginmanager.getplugin("capturemanager")
        if capman is not None:
            capman.suspend_global_capture(in_=True)
    except Exception:
        pass
    slow = ", ".join("%s %.0fs" % (n, s) for s, n in sorted(_wd["slow"], reverse=True)[:8])
    msg = ("\n\nSNK GPU WATCHDOG: %s did not finish within %.0f s (SNK_GPU_TEST_TIMEOUT_S) -- a kernel or a child process of this test "
           "hangs.\nbefore it: %d passed, %d failed, %d skipped; slowest so far: %s\nFAILED %s - watchdog\n"
           % (nodeid, limit, _wd["passed"], _wd["failed"], _wd["skipped"], slow or "-", nodeid))
    for fd in (2, 1):
        try:
            os.write(fd, msg.encode())
        except OSError:
            pass
    try:
        faulthandler.dump_traceback(file=2, all_threads=True)
    except Exception:
        pass
    os._exit(3)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_protocol(item, nextitem):
    if item.get_closest_marker("gpu") is None:
        yield
        return
    if _wd["t0"] is None:
        _wd["t0"] = time.monotonic()
    timer = None
    if _WD_TEST_S > 0:
        mark = item.get_closest_marker("timeout")           # a test that asks for longer (pytest-timeout's own marker) gets it
        limit = max(_WD_TEST_S, float(mark.args[0])) if mark and mark.args else _WD_TEST_S
        timer = threading.Timer(limit, _wd_fire, (item.nodeid, limit))
        timer.daemon = True
        timer.start()
    t = time.monotonic()
    try:
        yield
    finally:
        if timer is not None:
            timer.cancel()
        _wd["slow"].append((time.monotonic() - t, item.nodeid.split("/")[-1]))


def pytest_runtest_setup(item):
    if (item.get_closest_marker("gpu") is not None and _WD_SUITE_S > 0 and _wd["t0"] is not None
            and time.monotonic() - _wd["t0"] > _WD_SUITE_S):
        pytest.skip("GPU suite budget of %.0f s spent (SNK_GPU_SUITE_BUDGET_S): not run, not green" % _WD_SUITE_S)


def pytest_runtest_logreport(report):
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        key = report.outcome if report.outcome in ("passed", "failed", "skipped") else "failed"
        _wd[key] += 1


@pytest.fixture(scope="session")
def snk_lib():
    """The HIP extension through its C ABI.  No fallback: missing library == failure."""
    from soapnuke_amd import abi
    return abi.load_library()


def pytest_collection_modifyitems(config, items):
    """The emulated tier (tests/test_simt_*.py) re-runs the GPU tier's test functions on the CPU; all of it takes about twelve minutes,
    so an ordinary run takes each module's CORE selection (a few minutes) and SNK_SIMT_FULL=1 the rest as well
    (profiles/r04_simt_full.txt holds such a run)."""
    # the emulated / instruction tiers re-use the GPU tier's test functions (star imports): none of them may carry the GPU tier's marker
    # along (a module-level `pytestmark` travels with `import *` -- it took tests/test_simt_cli.py out of the CPU suite once)
    stray = [it.nodeid for it in items if os.path.basename(str(it.fspath)).startswith("test_simt_") and it.get_closest_marker("gpu")]
    if stray:
        raise pytest.UsageError("CPU-tier tests carry the gpu marker: %s ..." % stray[:3])
    # first-contact tests go behind everything else (stable order otherwise): the driver runs `pytest -m gpu -x`
    items.sort(key=lambda it: it.get_closest_marker("first_contact") is not None)
    if os.environ.get("SNK_SIMT_FULL") == "1":
        return
    skip = pytest.mark.skip(reason="emulated tier beyond its core selection: SNK_SIMT_FULL=1 runs it")
    for item in items:
        core = getattr(item.module, "CORE", None)
        if core is not None and not any(c in item.name for c in core):
            item.add_marker(skip)
