"""No GPU: the kernels that round 4 freed of scratch-resident objects and generic-pointer accesses stay that way
(tools/isa_allocas.py: device bitcode through `opt -O3`, counts from the gfx950 assembly).  A private object whose address is
compared or that is indexed with a run-time value is not promoted to registers -- every access becomes a scratch load, pointers
stored in it become generic (flat accesses); nothing but the assembly shows it."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel -> (largest scratch size in bytes, most flat instructions); spilled VGPRs: none, except where SPILLS says so
BUDGET = {
    "snk_long_decide_kernel": (16, 0),
    "snk_contam_kernel<5>": (0, 0),
    "snk_contam_kernel<8>": (0, 0),
    "snk_long_contam_kernel": (32, 0),      # capped at 256 VGPRs for two waves per SIMD: six spilled registers
    "inf_decode_coop_kernel": (16, 0),
    "inf_decode_kernel": (16, 0),
}
SPILLS = {"snk_long_contam_kernel": 8}


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_no_scratch_resident_objects_in_the_freed_kernels():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_allocas.py"), "snk_long.hip", "snk_contam.hip", "snk_inflate.hip"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    seen = {}
    for line in r.stdout.split("\n"):
        m = re.match(r"\s+(?:void )?([\w<>]+): vgprs (\d+), spilled (\d+), scratch bytes (\d+), flat (\d+), scratch instr (\d+)", line)
        if m:
            seen[m.group(1)] = (int(m.group(3)), int(m.group(4)), int(m.group(5)))
    for k, (max_scratch, max_flat) in BUDGET.items():
        assert k in seen, (k, sorted(seen))
        spilled, scratch, flat = seen[k]
        assert spilled <= SPILLS.get(k, 0) and scratch <= max_scratch and flat <= max_flat, (k, seen[k], r.stdout)
