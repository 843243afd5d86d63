"""What the HIP runtime or the hardware would refuse at LAUNCH time, read from the metadata of the kept gfx950 assembly -- neither CPU tier
meets these limits (the emulator takes any kernarg size and any register count; the instruction tier runs whatever is there):
the kernarg segment (4 KB is HIP's limit for by-value arguments), static LDS against the CU's 160 KB, registers against the waves
per SIMD the kernel's largest workgroup needs (512 unified VGPRs per SIMD lane on gfx950: 1024 work-items = 4 waves per SIMD = 128
registers, arch + acc), wave size 64, no dynamic stack (an indirect call or recursion would need run-time scratch sizing), scratch
per lane within reason.  No GPU."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "soapnuke_amd", "csrc", "build")
FILES = sorted(glob.glob(os.path.join(BUILD, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
pytestmark = pytest.mark.skipif(not FILES, reason="the build's kept assembly is not there (python __graft_entry__.py)")

KEYS = ("kernarg_segment_size", "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size", "wavefront_size",
        "vgpr_count", "agpr_count", "sgpr_count")


def kernels_of(path):
    """[{name, <KEYS>, uses_dynamic_stack}] from the amdhsa.kernels metadata at the end of the file"""
    text = open(path).read()
    meta = text[text.rfind("amdhsa.kernels:"):]
    out = []
    for block in re.split(r"\n  - \.", meta)[1:]:
        k = {"file": os.path.basename(path)}
        m = re.search(r"\.name:\s+(\S+)", block)
        if not m:
            continue
        k["name"] = m.group(1)
        for key in KEYS:
            m = re.search(r"\.%s:\s+(\d+)" % key, block)
            k[key] = int(m.group(1)) if m else 0
        k["uses_dynamic_stack"] = bool(re.search(r"\.uses_dynamic_stack:\s+true", block))
        out.append(k)
    return out


def test_every_kernel_fits_the_launch_limits():
    ks = [k for f in FILES for k in kernels_of(f)]
    assert len(ks) >= 60, len(ks)                 # (66 kernels in round 5's build)
    bad = []
    for k in ks:
        waves_per_simd = max(1, (k["max_flat_workgroup_size"] + 255) // 256)          # a workgroup's waves spread over the CU's 4 SIMDs
        regs = k["vgpr_count"] + k["agpr_count"]
        why = []
        if k["kernarg_segment_size"] > 4096:
            why.append("kernarg segment of %d bytes" % k["kernarg_segment_size"])
        if k["group_segment_fixed_size"] > 160 * 1024:
            why.append("static LDS of %d bytes" % k["group_segment_fixed_size"])
        if regs > 512 // waves_per_simd:
            why.append("%d registers with %d waves per SIMD in one workgroup" % (regs, waves_per_simd))
        if k["wavefront_size"] != 64:
            why.append("wave size %d" % k["wavefront_size"])
        if k["uses_dynamic_stack"]:
            why.append("dynamic stack")
        if k["private_segment_fixed_size"] > 2048:
            why.append("%d bytes of scratch per lane" % k["private_segment_fixed_size"])
        if why:
            bad.append((k["file"], k["name"], why))
    assert not bad, bad
