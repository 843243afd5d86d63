"""The drop-in boundary from the reference's side: oracle/gpu_binding.cpp is a real
`gpuPeProcess : public peProcess`, compiled against the reference's own headers and objects and linked with
libsnk_filter.so.  It overrides the virtual seam -- both `filter_pe_fqs` overloads, src/peprocess.h:61-62 -- and the
reference's own process() (readers, threads, temp files, writers, its non-virtual CPU stat_pe_fqs, merge and
report code) runs around it.  Same FASTQ through `oracle/_ref/SOAPnuke_gpu` and the unmodified `oracle/_ref/SOAPnuke`:
every report file and the clean FASTQ must be byte-identical."""
import filecmp
import gzip
import os
import subprocess

import pytest

import report_util as R
import snk_testlib as T
from soapnuke_amd import synth

BIND_BIN = os.path.join(T.ROOT, "oracle", "_ref", "SOAPnuke_gpu")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(T.REF_BIN) and os.path.exists(BIND_BIN)), reason="oracle/_ref not built")]


def _cat(path):
    return gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()


def _run_binding(case, work):
    name, paired, L, n, threads, patch, skw, pkw, cli, cfg = case
    cmd = [BIND_BIN, "filter", "-1", os.path.join(work, "r1.fq.gz"), "-2", os.path.join(work, "r2.fq.gz"), "-C", "c1.fq", "-D", "c2.fq",
           "-o", os.path.join(work, "bind"), "-T", str(threads)]
    if os.path.exists(os.path.join(work, "cfg")):
        cmd += ["-c", os.path.join(work, "cfg")]
    r = subprocess.run(cmd + cli, capture_output=True)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-500:])
    return os.path.join(work, "bind")


PE = [c for c in R.REPORT_CASES if c[1]]


@pytest.mark.parametrize("case", PE, ids=[c[0] for c in PE])
def test_reference_process_through_the_gpu_binding(case, tmp_path):
    d, p = R.case_inputs(case)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    got = _run_binding(case, work)
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(got, f), os.path.join(ref, f), shallow=False), f
    for c in ("c1.fq", "c2.fq"):
        assert _cat(os.path.join(got, c)) == _cat(os.path.join(ref, c)), c


def test_binding_rmdup_overload_and_trim_outputs(tmp_path):
    """filter_pe_fqs(opt, index): the duplicate flags of the reference's own pre-pass enter the GPU cascade as
    snk_batch.dup; trimFq1/trimFq2 take the trim_result vectors the binding fills (double pe_info suffix, quirk Q7)."""
    n, L, threads, patch = 20000, 150, 3, 250
    d = synth.make_batch(n, L, paired=True, seed=61)
    for m in range(2):
        d["seq"][m][10000:12000] = d["seq"][m][0:2000]
    cli = ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J"]
    case = ("bind_rmdup", True, L, n, threads, patch, {}, {}, cli, ["rmdup", "trimFq1=t1.fq.gz", "trimFq2=t2.fq.gz", "pe_info"])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    got = _run_binding(case, work)
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(got, f), os.path.join(ref, f), shallow=False), f
    for c in ("c1.fq", "c2.fq", "t1.fq.gz", "t2.fq.gz"):
        assert _cat(os.path.join(got, c)) == _cat(os.path.join(ref, c)), c
    for t in range(threads):
        for m in (1, 2):
            f = f"dupReads.{t}.{m}.gz"
            assert _cat(os.path.join(got, f)) == _cat(os.path.join(ref, f)), f
