"""Random shapes x random parameter contexts through the instruction tier (tools/isa_fuzz.py): a batch through the emulated library with
its launches captured (the capture compares the emulated result with the oracle), every launch replayed from the kept gfx950 assembly
with noise in the registers, memory compared byte for byte.  An ordinary run takes four contexts; SNK_SIMT_FULL=1 forty
(profiles/r06_isa_fuzz.txt holds two sweeps of 120 and 240 contexts: no difference).  No GPU."""
import os
import sys

import pytest

import snk_testlib as T

sys.path.insert(0, os.path.join(T.ROOT, "tools"))
import isa_fuzz as F               # noqa: E402
import test_simt_isa_interp as TI  # noqa: E402

CORE = ["test_random_contexts_from_the_assembly[5000]", "test_random_contexts_from_the_assembly[5001]", "test_random_contexts_from_the_assembly[5002]",
        "test_random_contexts_from_the_assembly[5003]"]
pytestmark = pytest.mark.skipif(not os.path.exists(TI.ASM), reason="the build's kept assembly is not there (python __graft_entry__.py)")


@pytest.mark.parametrize("seed", range(5000, 5040))
def test_random_contexts_from_the_assembly(seed):
    TI.simt_lib_path()
    _, spec, out = F.one(seed)
    assert out and all(o[2] == "ok" for o in out), (spec, out)
