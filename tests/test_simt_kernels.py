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
import test_contam_fuzz_gpu as CF
import test_gpu_parity as GP
import test_long_reads_gpu as LR
import test_phred_gpu as PH
import test_rmdup_gpu as RD
import test_fastq_gpu as FQ
import test_gunzip_gpu as GZ
import test_bittr_gpu as BT

# what an ordinary run takes (substrings of the test ids); SNK_SIMT_FULL=1: everything (tests/conftest.py)
CORE = ["test_pe150_cases[C3_full", "test_pe150_cases[C2_adatrim_lowq", "test_pe150_cases[hard_lq_trim-0", "test_pe150_cases[meanq_polyx-0", "test_se100_cases[C3_full-0",
        "test_pe250_full[0]", "test_ragged_and_chunked[0]", "test_tiny_batches[0]", "test_error_reporting[0]", "test_lowercase_and_N_reads[0]",
        "test_pitch_not_multiple_of_16[152-True-C3_full]", "test_capacity_boundaries", "test_contaminant_screening",
        "test_multi_flush_launches_emulated[5-1-C3_full", "test_adapter_fuzz[0]", "test_adapter_fuzz[27]", "test_adapter_budgets_above_three[mis0]",
        "test_adapters_of_any_length_on_the_tiled_kernel[0]", "test_adapters_of_any_length_on_the_tiled_kernel[5]", "test_adapters_of_any_length_on_the_tiled_kernel[10]",
        "test_long_adapter_lists_and_lower_case_on_the_fast_paths[6]", "test_contam_fuzz[8]", "test_contam_fuzz_long_reads[4]", "test_long_reads[600", "test_long_reads[1000-True",
        "test_long_reads_plane_store", "test_long_reads_adapters_of_any_length[600-200", "test_long_reads_adapters_of_any_length[600-3-", "test_random_parameter_contexts_on_the_device[20-", "test_hash_vs_oracle[150-160-True", "test_hash_odd", "test_mark_vs_oracle",
        "test_one_pass_table_single_end_shift", "test_exchange_helpers_against_the_oracle[3]", "test_parse_and_format", "test_device_gzip_members_round_trip[5000", "test_device_inflate_kernels_produce_zlibs_bytes[65536]",
        "test_bit_transpose[random]"]
CAP = 12000          # pairs per batch under the emulator (about 10 k pairs a second here); sizes up to it stay as they are


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
    S.torch_on_host(monkeypatch)
    real_make = synth.make_batch

    def make_batch(n, *a, **k):
        return real_make(int(n) if int(n) <= CAP else CAP + int(n) % 61, *a, **k)

    monkeypatch.setattr(synth, "make_batch", make_batch)
    for mod in (GP, AF, CF, LR, PH, RD, FQ, GZ, BT):
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


# ---- the fuzz suites: every k-th context of the GPU tier's lists (the functions themselves are the GPU tier's)
@pytest.mark.parametrize("i", range(0, AF.N_CONTEXTS, 3))
def test_adapter_fuzz(i):
    AF.test_adapter_fuzz(i)


def test_phase_a_dimers_reach_the_tiled_kernel(monkeypatch):
    import sys
    monkeypatch.setattr(sys.modules[__name__], "CAP", 40000)          # (its count of phase-A reads wants the full batch)
    AF.test_phase_a_dimers_reach_the_tiled_kernel()


@pytest.mark.parametrize("mis", [(4, 5), (6, 4), (9, 3)])
def test_adapter_budgets_above_three(mis):
    AF.test_adapter_budgets_above_three(mis)


@pytest.mark.parametrize("i", range(0, 16, 2))
def test_long_adapter_lists_and_lower_case_on_the_fast_paths(i):
    AF.test_long_adapter_lists_and_lower_case_on_the_fast_paths(i)


@pytest.mark.parametrize("i", range(12))
def test_adapters_of_any_length_on_the_tiled_kernel(i):
    """(written while the GPU was closed: this tier is the first place it runs)"""
    AF.test_adapters_of_any_length_on_the_tiled_kernel(i)


@pytest.mark.parametrize("i", range(0, CF.N_CONTEXTS, 4))
def test_contam_fuzz(i):
    CF.test_contam_fuzz(i)


@pytest.mark.parametrize("i", range(0, 20, 4))
def test_contam_fuzz_long_reads(i):
    CF.test_contam_fuzz_long_reads(i)


@pytest.mark.parametrize("i", range(0, 10, 3))
def test_global_contaminants_outside_the_event_walk_range(i):
    CF.test_global_contaminants_outside_the_event_walk_range(i)


@pytest.mark.parametrize("L,paired,var,name", LR.CASES)
def test_long_reads(L, paired, var, name):
    LR.test_long_reads(L, paired, var, name)


@pytest.mark.parametrize("L,alen,edge,mr", [(600, 100, 6, 0.5), (600, 3, 2, 0.7), (600, 200, 6, 0.5), (1000, 255, 10, 0.3), (600, 40, 60, 0.5), (640, 5, 6, 1.0),
                                            (1000, 130, 3, 0.5)])
def test_long_reads_adapters_of_any_length(L, alen, edge, mr):
    LR.test_long_reads_adapters_of_any_length(L, alen, edge, mr)


from test_long_reads_gpu import (test_long_reads_several_adapters_and_budgets, test_long_reads_contaminants_and_duplicates,      # noqa: E402,F401
                                 test_long_reads_quality_range_error, test_long_reads_plane_store_group_edges_and_regrowth)
from test_phred_gpu import test_se100_phred_and_max_quality, test_quality_range_errors_follow_the_offset      # noqa: E402,F401


@pytest.mark.parametrize("kernel", [1, 0])
@pytest.mark.parametrize("phred,mbq", [(33, 40), (64, 42), (33, 50)])
@pytest.mark.parametrize("name", ["C3_full", "hard_lq_trim"])
def test_pe150_phred_and_max_quality(name, phred, mbq, kernel):
    PH.test_pe150_phred_and_max_quality(name, phred, mbq, kernel)


@pytest.mark.parametrize("L", [250, 1000])
@pytest.mark.parametrize("phred,mbq", [(64, 42), (33, 45)])
def test_long_reads_phred_and_max_quality(L, phred, mbq):
    PH.test_long_reads_phred_and_max_quality(L, phred, mbq)


@pytest.mark.parametrize("kernel", [1, 0])
@pytest.mark.parametrize("i", range(0, 48, 4))
def test_random_parameter_contexts_on_the_device(i, kernel):
    PH.test_random_parameter_contexts_on_the_device(i, kernel)


# ---- duplicate marking (snk_rmdup.hip), device-side FASTQ text (snk_fastq.hip), gzip members (snk_gzip.hip), inflate (snk_inflate.hip),
# the 64 x 64 bit transpose (snk_bittr.hip.h)
from test_rmdup_gpu import (test_hash_golden, test_hash_vs_oracle, test_hash_odd_tile_counts, test_mark_golden, test_mark_vs_oracle_random,      # noqa: E402,F401
                            test_mark_with_explicit_indices, test_too_many_reads_is_refused, test_one_pass_table_vs_oracle,
                            test_one_pass_table_refuses_too_many_reads)


def test_one_pass_table_single_end_shift():
    RD.test_one_pass_table_single_end_shift()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_exchange_helpers_against_the_oracle(world):
    RD.test_exchange_helpers_against_the_oracle(world)


from test_fastq_gpu import (test_parse_and_format_match_the_restatement, test_parse_reports_bad_input, test_device_gzip_members_round_trip,      # noqa: E402,F401
                            test_format_selects_other_verdicts_whole)
from test_gunzip_gpu import test_device_inflate_kernels_produce_zlibs_bytes, test_device_inflate_refuses_what_does_not_fit      # noqa: E402,F401
from test_bittr_gpu import test_bit_transpose      # noqa: E402,F401


@pytest.mark.parametrize("pair", ["0", "2"])
def test_the_other_strip_pairing_builds_emulated(pair):
    """SNK_PAIR: the shipped build pairs the last quality strips of two reads on the static-shape loop (1); 0 (no pairing) and 2 (also
    on the run-time-shape loop) are the A/B builds of tools/ab.sh -- each as its own emulated library, on the cases that walk phase 1's
    whole-tile loops and the flush's remapped slots"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, SIMT_EXTRA_CXXFLAGS="-DSNK_PAIR=" + pair, SIMT_TAG="pair" + pair, SNK_SIMT_FULL="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "test_pe150_cases and (C2_adatrim or C3_full) or test_se100_cases and C3_full or test_multi_flush_launches_emulated or test_pitch_not_multiple"],
                       capture_output=True, text=True, env=env, cwd=T.ROOT)
    assert r.returncode == 0, r.stdout[-1500:]
