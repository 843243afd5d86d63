"""The report writer (merge_stat/update_stat/print_stat restatement, soapnuke_amd/host/snk_report.cpp)
must reproduce the reference's 10 (PE) / 6 (SE) report files byte for byte, including the
dependence on the -T thread partition.  Stats come from the oracle here (CPU) and
from the HIP path in the gpu-marked variant; expected bytes are the committed golden files that
tests/golden/make_golden_reports.py took from the compiled reference binary, and -- where that
binary is present -- a live run."""
import filecmp
import os

import pytest

import report_util as R
import snk_testlib as T

GOLD = os.path.join(T.ROOT, "tests", "golden", "reports")
IDS = [c[0] for c in R.REPORT_CASES]


def _oracle_run(p, sub, first_index, stats):
    o = T.run_oracle(p, sub, first_index=first_index, stats=stats)
    assert o["rc"] == 0


def _compare(case, got_dir, want_dir):
    for f in (R.REPORT_FILES_PE if case[1] else R.REPORT_FILES_SE):
        a, b = os.path.join(got_dir, f), os.path.join(want_dir, f)
        if not filecmp.cmp(a, b, shallow=False):
            la, lb = open(a).read().split("\n"), open(b).read().split("\n")
            bad = [i for i, (x, y) in enumerate(zip(la, lb)) if x != y][:3]
            raise AssertionError(f"{f}: first differing lines {bad}: " + " || ".join(f"{la[i]!r} vs {lb[i]!r}" for i in bad))


@pytest.mark.parametrize("case", R.REPORT_CASES, ids=IDS)
def test_reports_match_golden(case, tmp_path):
    d, p = R.case_inputs(case)
    R.write_reports(p, R.vthread_stats(case, d, p, _oracle_run), str(tmp_path / "ours"))
    _compare(case, str(tmp_path / "ours"), os.path.join(GOLD, case[0]))


@pytest.mark.skipif(not os.path.exists(T.REF_BIN), reason="oracle/_ref/SOAPnuke not built")
@pytest.mark.parametrize("case", R.REPORT_CASES[:2], ids=IDS[:2])
def test_reports_match_live_reference(case, tmp_path):
    if case[0] == "pe_adatrim_T4" and os.environ.get("SNK_SIMT_FULL") != "1":
        # (70 s, sixty of them the reference's own wait in remove_tmpDir after a run with several threads; its reports are the
        #  golden files test_reports_match_golden compares with on every run)
        pytest.skip("the reference's -T 4 run takes 70 s: SNK_SIMT_FULL=1 runs it")
    d, p = R.case_inputs(case)
    ref = R.run_reference_cli(case, d, str(tmp_path / "work"))
    R.write_reports(p, R.vthread_stats(case, d, p, _oracle_run), str(tmp_path / "ours"))
    _compare(case, str(tmp_path / "ours"), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", R.REPORT_CASES, ids=IDS)
def test_reports_from_hip_stats(case, tmp_path):
    """Same files from the GPU path's per-virtual-thread accumulators (snk_bind_stats per block)."""
    import numpy as np
    import torch
    from soapnuke_amd.filter import FilterContext
    d, p = R.case_inputs(case)
    ctx = FilterContext(p, device=0)
    dev = ctx.upload(d)

    def run(pp, sub, first_index, stats):
        lo = first_index
        hi = lo + sub["n"]
        s = torch.zeros(ctx.sum_u64, dtype=torch.int64, device="cuda")
        mx = torch.zeros(8, dtype=torch.int64, device="cuda")
        ctx._check(ctx.lib.snk_bind_stats(ctx.ctx, s.data_ptr(), mx.data_ptr()))
        part = {"n": sub["n"], "L": dev["L"], "pitch": dev["pitch"], "seq": [x[lo:hi] for x in dev["seq"]],
                "qual": [x[lo:hi] for x in dev["qual"]], "len": [None if x is None else x[lo:hi] for x in dev["len"]]}
        rec = ctx.alloc_records(sub["n"])
        ctx.filter_batch(ctx.make_batch(part, first_index=lo), rec)
        ctx.finalize()
        torch.cuda.synchronize()
        np.add(stats[0], s.cpu().numpy().view(np.uint64), out=stats[0])
        np.maximum(stats[1], mx.cpu().numpy().view(np.uint64), out=stats[1])

    stats = R.vthread_stats(case, d, p, run)
    # gs a/c/g/t/n/bases/q20/q30 are derived per block by the finalize kernel: sums of sums stay exact
    R.write_reports(p, stats, str(tmp_path / "ours"))
    _compare(case, str(tmp_path / "ours"), os.path.join(GOLD, case[0]))


# ---- cal_quar_from_array keeps its positions in 32-bit ints; data_num*3 wraps past 715 M counts,
# data_num*9 past 238 M -- exactly what BASELINE configs[2]/[3] (628 M reads) feed it.

def _quartiles_ours(data, nq, length):
    import ctypes as C
    import numpy as np
    lib = C.CDLL(os.path.join(T.ROOT, "soapnuke_amd", "libsnk_report.so"))
    lib.snk_report_quartiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    out = np.zeros(6, dtype=np.float32)
    mask = lib.snk_report_quartiles(data.ctypes.data, nq, length, out.ctypes.data)
    return out, np.array([(mask >> k) & 1 for k in range(6)], dtype=bool)


def _quartiles_ref(data, length):
    import ctypes as C
    import numpy as np
    lib = T.ref_lib()
    lib.snkref_cal_quar.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros(6, dtype=np.float32)
    padded = np.concatenate([data, np.zeros(2, dtype=np.uint64)])       # the reference reads data[len] (one past a PE row)
    lib.snkref_cal_quar(padded.ctypes.data, length, out.ctypes.data)
    return out


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref/libsnkref.so not built")
def test_quartiles_int32_wrap_vs_reference():
    import numpy as np
    rng = np.random.default_rng(12)
    nq = 43
    totals = [1000, 2.0e8, 2.38e8, 2.39e8, 4.0e8, 6.28e8, 7.15e8, 7.2e8, 1.0e9, 2.1e9, 2.2e9, 3.5e9, 4.4e9, 7.0e8 * 9]
    seen_wrap = 0
    for total in totals:
        for shape in range(4):
            w = rng.random(nq) ** (1 + 2 * shape)
            if shape == 3:
                w[:] = 0
                w[[2, 14, 36, 41]] = [0.1, 0.2, 0.5, 0.2]                  # Illumina-style binned qualities
            data = np.floor(w / w.sum() * total).astype(np.uint64)
            for length in (nq - 1, nq):                                    # PE passes max_qual, SE max_qual + 1
                (a, defined), b = _quartiles_ours(data, nq, length), _quartiles_ref(data, length)
                # a wrapped, negative position matches no bin: that field is uninitialised memory in the reference
                assert np.array_equal(a[defined], b[defined]), (total, shape, length, a, b, defined)
                assert defined[:3].all() or int(data.sum()) >= 2**31
            n = int(data.sum())
            if (n * 9) % 2**32 != n * 9:
                seen_wrap += 1
    assert seen_wrap > 10
