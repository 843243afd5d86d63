"""HIP path vs oracle with quality_phred = 64, non-default max_base_quality and random parameter sets
(VERDICT r2 task 1 i): the tiled kernel's clamp [phred-1, phred+nq], the byte-parallel thresholds, the long-read
path's byte tricks and the histogram row count all depend on both."""
import numpy as np
import pytest

import snk_testlib as T
from cases import PE_CASES, rebase_quality, se_kwargs
from soapnuke_amd import abi, synth
from test_gpu_parity import assert_same, run_hip_device
from test_oracle_vs_ref import _fuzz_case

pytestmark = pytest.mark.gpu

KERNELS = [1, 0]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mbq", [40, 42, 45, 50])
@pytest.mark.parametrize("phred", [33, 64])
@pytest.mark.parametrize("name", ["C2_adatrim_lowq", "C3_full", "hard_lq_trim"])
def test_pe150_phred_and_max_quality(name, phred, mbq, kernel):
    d = synth.make_batch(12000, 150, paired=True, var_len=(name == "hard_lq_trim"), seed=61)
    rebase_quality(d, phred, mbq, seed=mbq)            # values up to maxBaseQuality itself: our rows are 0..maxBaseQuality
    p = abi.default_params(paired=True, max_read_len=150, quality_phred=phred, output_quality_phred=phred,
                           max_base_quality=mbq, **PE_CASES[name])
    assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), True)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("phred,mbq", [(64, 42), (64, 40), (33, 50), (64, 50)])
def test_se100_phred_and_max_quality(phred, mbq, kernel):
    d = synth.make_batch(9000, 100, paired=False, var_len=True, seed=62)
    rebase_quality(d, phred, mbq, seed=3)
    p = abi.default_params(paired=False, max_read_len=100, quality_phred=phred, max_base_quality=mbq,
                           **se_kwargs(PE_CASES["C3_full"]))
    assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), False)


@pytest.mark.parametrize("L", [250, 400, 1000])
@pytest.mark.parametrize("phred,mbq", [(64, 42), (33, 45), (64, 50), (64, 40)])
def test_long_reads_phred_and_max_quality(L, phred, mbq):
    """PE250 (tiled, NW = 8) and the long-read path (snk_long.hip) -- kernel = 2 must accept them"""
    d = synth.make_batch(3000 if L > 256 else 6000, L, paired=True, var_len=True, seed=63 + L)
    rebase_quality(d, phred, mbq, seed=5)
    p = abi.default_params(paired=True, max_read_len=L, quality_phred=phred, max_base_quality=mbq, **PE_CASES["C3_full"])
    want = T.run_oracle(p, d)
    assert_same(p, run_hip_device(p, d, 2), want, True)
    assert_same(p, run_hip_device(p, d, 1), want, True)


@pytest.mark.parametrize("phred", [33, 64])
def test_quality_range_errors_follow_the_offset(phred):
    """a quality below the offset or above maxBaseQuality is reported (first offender), whatever the offset"""
    p = abi.default_params(paired=True, max_read_len=150, quality_phred=phred, max_base_quality=40)
    for kernel in KERNELS:
        for mate, row, pos, q in ((0, 17, 5, 41), (1, 900, 100, -1), (0, 2999, 149, 60), (1, 7, 64, -30)):
            d = synth.make_batch(3000, 150, paired=True, seed=29)
            rebase_quality(d, phred, 40, seed=1)
            d["qual"][mate][row, pos] = phred + q
            got = run_hip_device(p, d, kernel)
            assert got["err"] == (abi.E_QUAL_RANGE, mate, row), (kernel, mate, row, pos, q, got["err"])
            assert T.run_oracle(p, d)["err"] == got["err"]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("i", range(48))
def test_random_parameter_contexts_on_the_device(i, kernel):
    """the contexts tests/test_oracle_vs_ref.py pins on the compiled reference, HIP vs oracle"""
    p, d, paired = _fuzz_case(i)
    assert_same(p, run_hip_device(p, d, kernel, chunks=2), T.run_oracle(p, d), paired)
