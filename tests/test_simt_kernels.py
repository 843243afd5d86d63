"""The kernel parity tests of the GPU tier (tests/test_*_gpu.py) run on the CPU: the same test functions, with the HIP sources
built for the host by the SIMT emulator of tests/simt (wavefronts on fibers, see tests/simt/hip/hip_runtime.h) in place of the
gfx950 library, and smaller batches.  What this tier can and cannot show: the kernels' LOGIC -- every phase of the tiled kernel,
the bit-sliced adapter search, the contaminant screens, the long-read path, the flush / reduce chain -- is executed instruction
for instruction as written (cross-lane operations included) and compared with the oracle bit for bit; timing, register
allocation, and what the LDS unit does behind hand-placed waits (tools/isa_lint.py covers that) are not."""
import numpy as np
import pytest

import simt_lib as S
import snk_testlib as T
from soapnuke_amd import synth

import test_adapter_fuzz_gpu as AF
import test_gpu_parity as GP

CAP = 5000          # pairs per batch under the emulator (about 10 k pairs a second here); sizes up to it stay as they are


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
    S.torch_on_host(monkeypatch)
    real_make = synth.make_batch

    def make_batch(n, *a, **k):
        return real_make(int(n) if int(n) <= CAP else CAP + int(n) % 61, *a, **k)

    monkeypatch.setattr(synth, "make_batch", make_batch)
    monkeypatch.setenv("SNK_RUN_UNVERIFIED", "1")          # (tests guarded until they have run on hardware: this tier is what they wait for)
    for mod in (GP, AF):
        monkeypatch.setattr(mod, "run_hip_device", S.run_device, raising=False)


from test_gpu_parity import (test_pe150_cases, test_se100_cases, test_pe250_full, test_ragged_and_chunked, test_tiny_batches,      # noqa: E402,F401
                             test_rmdup_flags, test_host_verdict_bits, test_error_reporting, test_lowercase_and_N_reads,
                             test_polyx_runs_of_every_kind, test_pitch_not_multiple_of_16, test_generic_anchor, test_capacity_boundaries,
                             test_contaminant_screening, test_host_pointer_entry, test_empty_batch, test_contaminant_list_size_mismatch_is_refused)


@pytest.mark.parametrize("case,var_len", [("C2_adatrim_lowq", 0), ("C3_full", 1)])
@pytest.mark.parametrize("wgs,every", [(2, 3), (5, 1)])
def test_multi_flush_launches_emulated(case, var_len, wgs, every):
    """test_gpu_parity.py::test_multi_flush_launches (the read-modify-write branch of the tiled kernel's histogram flush, the zero
    invariant of the per-workgroup partials across launches) in a child process with the library's test hooks set"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, SNK_TEST_MAX_WGS=str(wgs), SNK_TEST_FLUSH_EVERY=str(every), PYTHONPATH=os.pathsep.join([here, T.ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(here, "flush_hook_child.py"), case, str(var_len), "simt"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "multi-flush OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.fixture
def snk_lib():
    return S.lib()
