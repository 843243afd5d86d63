import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "first_contact: GPU test of code that has not run on hardware yet -- runs by default, sorted last")


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
