"""Child process of tests/test_gpu_parity.py::test_multi_flush_launches: the tiled kernel with SNK_TEST_MAX_WGS /
SNK_TEST_FLUSH_EVERY set (read once per process by snk_tiled.hip's launch()), so that a small batch takes many iterations per
workgroup and many flushes per launch -- the read-modify-write branch of the two-stage histogram flush and the reduce
kernel's re-zeroing of DevStats::part between back-to-back launches on one stream slot."""
import sys

import numpy as np

import snk_testlib as T
from cases import PE_CASES
from soapnuke_amd import abi, synth
from soapnuke_amd.filter import FilterContext, records_to_numpy


def main():
    case, var_len = sys.argv[1], sys.argv[2] == "1"
    n = 150_000
    if len(sys.argv) > 3 and sys.argv[3] == "simt":          # tests/test_simt_kernels.py: the emulated library, a tenth of the reads
        import pytest
        import simt_lib
        mp = pytest.MonkeyPatch()
        simt_lib.torch_on_host(mp)
        n = 15_000
    d = synth.make_batch(n, 150, paired=True, seed=77, var_len=var_len)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES[case])
    ctx = FilterContext(p, device=0)
    dev = ctx.upload(d)
    rec = ctx.alloc_records(n)
    want = T.new_stats(p)
    o = None
    for rep in range(3):                    # back-to-back launches on the same slot: the slices must be zero again each time
        ctx.filter_batch(ctx.make_batch(dev, first_index=rep * n), rec, kernel=2)
        o = T.run_oracle(p, d, first_index=rep * n, stats=want)
    s, mx, err = ctx.fetch()
    assert err[0] == 0, err
    for m in range(2):
        assert np.array_equal(records_to_numpy(rec[m]), o["rec"][m]), f"records differ (mate {m})"
    assert np.array_equal(s, want[0]), T.describe_stats_diff(p, s, want[0])
    assert np.array_equal(mx, want[1])
    print("multi-flush OK")


if __name__ == "__main__":
    main()
