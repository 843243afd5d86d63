"""Child of tests/test_gpu_watchdog.py (not collected on its own: no test_ prefix).  Three gpu-marked tests; the middle one blocks in native
code with the GIL released -- what a hipDeviceSynchronize behind a kernel that never returns looks like from Python."""
import ctypes
import os
import time

import pytest

pytestmark = pytest.mark.gpu


def test_before():
    time.sleep(float(os.environ.get("SNK_WD_CHILD_SLEEP", "0")))


def test_native_call_that_never_returns():
    if os.environ.get("SNK_WD_CHILD_HANG") == "1":
        ctypes.CDLL(None).pause()


def test_after():
    pass
