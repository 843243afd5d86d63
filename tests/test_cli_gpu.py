"""End-to-end drop-in check on the GPU box: this repo's `SOAPnuke filter` (C++ host + HIP hot path)
against the compiled reference binary on the same FASTQ files -- all report files byte-identical,
decompressed clean FASTQ byte-identical.  The reference is fed < 1 cycle of reads (or .gz),
SURVEY quirk Q10."""
import filecmp
import gzip
import os
import subprocess

import numpy as np
import pytest

import report_util as R
import snk_testlib as T
from soapnuke_amd import synth

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(T.REF_BIN), reason="oracle/_ref/SOAPnuke not built")]
CLI = os.path.join(T.ROOT, "soapnuke_amd", "SOAPnuke")


def _cat(path):
    return gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()


def _run_ours(case, work, gz, env=None):
    name, paired, L, n, threads, patch, skw, pkw, cli, cfg = case
    ext = ".fq.gz" if gz else ".fq"
    cmd = [CLI, "filter", "-1", os.path.join(work, "r1" + ext), "-C", "c1" + ext, "-o", os.path.join(work, "ours"), "-T", str(threads)]
    if paired:
        cmd += ["-2", os.path.join(work, "r2" + ext), "-D", "c2" + ext]
    if os.path.exists(os.path.join(work, "cfg")):
        cmd += ["-c", os.path.join(work, "cfg")]
    r = subprocess.run(cmd + cli, capture_output=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-500:])
    return os.path.join(work, "ours")


@pytest.mark.parametrize("case", R.REPORT_CASES, ids=[c[0] for c in R.REPORT_CASES])
def test_cli_matches_reference_binary(case, tmp_path):
    d, p = R.case_inputs(case)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)   # writes r1.fq(.gz) / r2.fq(.gz) / cfg, runs the reference
    ours = _run_ours(case, work, gz=False)
    for f in (R.REPORT_FILES_PE if case[1] else R.REPORT_FILES_SE):
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in (["c1.fq", "c2.fq"] if case[1] else ["c1.fq"]):
        assert _cat(os.path.join(ours, c)) == _cat(os.path.join(ref, c)), c


@pytest.mark.parametrize("L,paired", [(600, True), (900, False)])      # (the reference binary itself rejects 1000-character lines)
def test_cli_long_reads(L, paired, tmp_path):
    """reads of 257..1000 positions end to end: the long-read kernels (snk_long.hip) behind the CLI, reports (600 / 1000 rows
    per position table) and clean FASTQ byte-identical to the reference binary"""
    kw = dict(adapters1=[synth.ADAPTER1], ada_trim=1, low_qual=10, low_qual_ratio=0.1, n_ratio=0.01)
    cli, cfg = ["-f", synth.ADAPTER1, "-J", "-l", "10", "-q", "0.1", "-n", "0.01"], []
    if paired:
        kw.update(adapters2=[synth.ADAPTER2], trim_bad_tail=(20, 30))
        cli += ["-r", synth.ADAPTER2]
        cfg = ["trimBadTail=20,30"]
    case = ("long%d" % L, paired, L, 6000, 2, 300, dict(seed=70 + L // 100), kw, cli, cfg)
    d, p = R.case_inputs(case)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False)
    for f in (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE):
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in (["c1.fq", "c2.fq"] if paired else ["c1.fq"]):
        assert _cat(os.path.join(ours, c)) == _cat(os.path.join(ref, c)), c


def test_cli_gz_in_gz_out(tmp_path):
    case = R.REPORT_CASES[0]
    d, p = R.case_inputs(case)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=True)
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in ("c1.fq", "c2.fq"):
        assert _cat(os.path.join(ours, c + ".gz")) == _cat(os.path.join(ref, c)), c


@pytest.mark.parametrize("paired,n", [(True, 20000), (False, 20000), (False, 20100), (True, 19999)])
def test_cli_rmdup_matches_reference_binary(paired, n, tmp_path):
    """config key `rmdup`: GPU hash + marking pre-pass, flags into the cascade, the
    dupReads.<thread>.<mate>.gz side files -- all against the reference binary."""
    L, threads, patch = 150, 3, 250       # n = 20100: a partial last patch (SE: the only one whose flags are aligned)
    d = synth.make_batch(n, L, paired=paired, seed=61)
    for m in range(2 if paired else 1):
        d["seq"][m][10000:12000] = d["seq"][m][0:2000]           # duplicates by sequence, qualities differ
        d["seq"][m][15000:15500] = d["seq"][m][0:500]            # third copies, in another virtual thread's block
        d["seq"][m][n - 40:n - 20] = d["seq"][m][700:720]        # duplicates inside the last patch
    cli = ["-f", synth.ADAPTER1, "-J"] + (["-r", synth.ADAPTER2] if paired else [])
    case = ("rmdup", paired, L, n, threads, patch, {}, {}, cli, ["rmdup"])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False)
    for f in (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE):
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in (["c1.fq", "c2.fq"] if paired else ["c1.fq"]):
        assert _cat(os.path.join(ours, c)) == _cat(os.path.join(ref, c)), c
    ndup = 0
    for t in range(threads):
        for m in range(2 if paired else 1):
            f = f"dupReads.{t}.{m + 1}.gz"
            a, b = _cat(os.path.join(ours, f)), _cat(os.path.join(ref, f))
            assert a == b, f
            ndup += a.count(b"\n") // 4
    assert ndup == 2520 * (2 if paired else 1)
    assert b"dup number:\t2520" in open(os.path.join(ours, "log"), "rb").read()


@pytest.mark.parametrize("mode", ["one_pass_gz", "two_pass", "sentinel_restart", "small_batches",
                                  pytest.param("table_does_not_fit", marks=T.first_contact)])
def test_cli_rmdup_one_pass_variants(mode, tmp_path):
    """Paired rmdup is one pass in device-text mode (a hash table resident in HBM, include/snk_rmdup.h snk_rmdup_stream_*):
    .gz output, the retained two-pass path (SNK_RMDUP_TWO_PASS=1), the restart a sentinel hash forces, and batches much
    smaller than the duplicates' distance (table growth, duplicates across batches) -- same bytes as the reference binary."""
    L, threads, patch, n = 150, 3, 250, 20000
    d = synth.make_batch(n, L, paired=True, seed=62)
    for m in range(2):
        d["seq"][m][10000:12000] = d["seq"][m][0:2000]
        d["seq"][m][15000:15500] = d["seq"][m][0:500]
        d["seq"][m][n - 40:n - 20] = d["seq"][m][700:720]
    cli = ["-f", synth.ADAPTER1, "-J", "-r", synth.ADAPTER2]
    env = {}
    if mode == "two_pass":
        env["SNK_RMDUP_TWO_PASS"] = "1"
    if mode == "sentinel_restart":
        env["SNK_RMDUP_SENTINEL_TEST"] = "1"
    if mode == "small_batches":
        env["SNK_BATCH_PAIRS"] = "1536"
    if mode == "table_does_not_fit":                         # the one-pass table's memory is not there: the two passes are chosen up front
        env["SNK_RMDUP_FREE_MB_TEST"] = "1"
    case = ("rmdup1", True, L, n, threads, patch, {}, {}, cli, ["rmdup"])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    gz = mode == "one_pass_gz"
    ours = _run_ours(case, work, gz=gz, env=env)
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in ("c1.fq", "c2.fq"):
        assert _cat(os.path.join(ours, c + (".gz" if gz else ""))) == _cat(os.path.join(ref, c)), c
    for t in range(threads):
        for m in range(2):
            f = f"dupReads.{t}.{m + 1}.gz"
            assert _cat(os.path.join(ours, f)) == _cat(os.path.join(ref, f)), f
    log = open(os.path.join(ours, "log"), "rb").read()
    assert b"dup number:\t2520" in log and b"duplicate reads number:\t2520" in log
    assert (b"restarted" in log) == (mode == "sentinel_restart")
    assert (b"the one-pass table would need" in log) == (mode == "table_does_not_fit")


@T.first_contact
@pytest.mark.parametrize("n,batch,mode", [(20000, "4096", "one"), (20100, "4096", "one"), (20100, "700", "one"), (19999, "4096", "gz"), (20100, "4096", "two_pass"),
                                           (20100, "4096", "sentinel_restart"), (20100, "2048", "two_devices")])
def test_cli_rmdup_single_end_one_pass(n, batch, mode, tmp_path):
    """Single-end rmdup in one pass (snk_rmdup_stream_mark_se_device): the reference filters read i of a full patch with the duplicate
    flag of read i - 1 (src/seprocess.cpp:1086,1159) and only the partial patch at the end of the file aligned -- batches cut at
    patch borders (4096 -> 4000, 700 -> 500 reads with patch = 250), the flag in front of a batch carried on the device, duplicates
    next to patch and batch borders and inside the last patch.  Same bytes as the reference binary, and as the two passes."""
    L, threads, patch = 150, 3, 250
    d = synth.make_batch(n, L, paired=False, seed=63)
    d["seq"][0][10000:12000] = d["seq"][0][0:2000]              # duplicates whose shifted flags cross batch borders (12000 = 3 x 4000)
    d["seq"][0][15999:16001] = d["seq"][0][3:5]                 # ... and sit on both sides of one
    d["seq"][0][4249:4251] = d["seq"][0][7:9]                   # both sides of a patch border
    d["seq"][0][n - 40:n - 20] = d["seq"][0][700:720]           # inside the last patch (partial for n = 20100 / 19999: aligned flags)
    d["seq"][0][n - 1] = d["seq"][0][1]                         # the file's last read
    cli = ["-f", synth.ADAPTER1, "-J"]
    env = {"SNK_BATCH_PAIRS": batch}
    if mode == "two_pass":
        env["SNK_RMDUP_TWO_PASS"] = "1"
    if mode == "sentinel_restart":
        env["SNK_RMDUP_SENTINEL_TEST"] = "1"
    case = ("rmdup_se", False, L, n, threads, patch, {}, {}, cli, ["rmdup"])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    gz = mode == "gz"
    ours_case = case
    if mode == "two_devices":                                   # the table on the first device, batches alternating between the two
        import torch
        ours_case = case[:8] + (cli + ["--devices", "0,1" if torch.cuda.device_count() >= 2 else "0,0"],) + case[9:]
    ours = _run_ours(ours_case, work, gz=gz, env=env)
    for f in R.REPORT_FILES_SE:
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    assert _cat(os.path.join(ours, "c1.fq" + (".gz" if gz else ""))) == _cat(os.path.join(ref, "c1.fq"))
    nd = 0
    for t in range(threads):
        f = f"dupReads.{t}.1.gz"
        a = _cat(os.path.join(ours, f))
        assert a == _cat(os.path.join(ref, f)), f
        nd += a.count(b"\n") // 4
    log = open(os.path.join(ours, "log"), "rb").read()
    assert nd > 2000 and (b"dup number:\t%d" % nd) in log
    assert (b"restarted" in log) == (mode == "sentinel_restart")
    assert (b"rmdup: one pass" in log) == (mode in ("one", "gz", "two_devices"))      # (the restarted run writes a new log)
    if mode in ("one", "gz", "two_devices"):
        assert (b"batches of %d reads" % (int(batch) // patch * patch)) in log


def test_cli_pe_info_outqual_and_crlf(tmp_path):
    """config keys pe_info + outQualSys (output-side transforms) and CRLF input (the first-line white-space rule)."""
    n, L = 3000, 100
    d = synth.make_batch(n, L, paired=True, seed=71)
    d["qual"][0][:] = np.minimum(d["qual"][0], 33 + 30)            # stay printable after +31
    d["qual"][1][:] = np.minimum(d["qual"][1], 33 + 30)
    case = ("peinfo", True, L, n, 2, 300, {}, {}, ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J"], ["pe_info", "outQualSys=1"])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False)
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in ("c1.fq", "c2.fq"):
        a = _cat(os.path.join(ours, c))
        assert a == _cat(os.path.join(ref, c)), c
        assert b"/1/1" in a or b"/2/2" in a                        # ids already end in /1 /2: pe_info appends again
    # the same reads with CRLF line ends: both tools strip 2 characters per line
    for m in (1, 2):
        txt = open(os.path.join(work, f"r{m}.fq"), "rb").read().replace(b"\n", b"\r\n")
        open(os.path.join(work, f"r{m}.fq"), "wb").write(txt)
        subprocess.check_call(["gzip", "-1", "-f", "-k", os.path.join(work, f"r{m}.fq")])
    cmd_tail = ["-C", "c1.fq", "-D", "c2.fq", "-T", "2", "-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-c", os.path.join(work, "cfg")]
    r = subprocess.run([T.REF_BIN, "filter", "-1", os.path.join(work, "r1.fq.gz"), "-2", os.path.join(work, "r2.fq.gz"), "-o", os.path.join(work, "ref2")] + cmd_tail, capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    r = subprocess.run([CLI, "filter", "-1", os.path.join(work, "r1.fq"), "-2", os.path.join(work, "r2.fq"), "-o", os.path.join(work, "ours2")] + cmd_tail, capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    for c in ("c1.fq", "c2.fq"):
        assert _cat(os.path.join(work, "ours2", c)) == _cat(os.path.join(work, "ref2", c)), c
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(work, "ours2", f), os.path.join(work, "ref2", f), shallow=False), f


@pytest.mark.parametrize("seq_type", ["0", "1"])
def test_cli_index_removal(seq_type, tmp_path):
    """config keys index + seqType: the index is cut out of the read names of the clean output (src/read_filter.cpp:357-382)."""
    n, L = 2000, 100
    d = synth.make_batch(n, L, paired=True, seed=72)
    work = str(tmp_path)
    for m in (1, 2):
        with open(os.path.join(work, f"r{m}.fq"), "wb") as f:
            for i in range(n):
                rid = (b"@FCD1PB1ACXX:4:1101:%d:2201#GAAGCACG/%d" % (i, m)) if seq_type == "0" else (b"@HISEQ:310:C5MH9ANXX:1:1101:%d:2043 %d:N:0:TCGGTCAC" % (i, m))
                f.write(rid + b"\n" + d["seq"][m - 1][i, :L].tobytes() + b"\n+\n" + d["qual"][m - 1][i, :L].tobytes() + b"\n")
        subprocess.check_call(["gzip", "-1", "-f", "-k", os.path.join(work, f"r{m}.fq")])
    open(os.path.join(work, "cfg"), "w").write(f"index\nseqType={seq_type}\npatch=300\n")
    tail = ["-C", "c1.fq", "-D", "c2.fq", "-T", "2", "-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-c", os.path.join(work, "cfg")]
    r = subprocess.run([T.REF_BIN, "filter", "-1", os.path.join(work, "r1.fq.gz"), "-2", os.path.join(work, "r2.fq.gz"), "-o", os.path.join(work, "ref")] + tail, capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    r = subprocess.run([CLI, "filter", "-1", os.path.join(work, "r1.fq"), "-2", os.path.join(work, "r2.fq"), "-o", os.path.join(work, "ours")] + tail, capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    for c in ("c1.fq", "c2.fq"):
        a = _cat(os.path.join(work, "ours", c))
        assert a == _cat(os.path.join(work, "ref", c)), c
        assert b"#GAAGCACG" not in a and b"N:0:TCGGTCAC" not in a
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(work, "ours", f), os.path.join(work, "ref", f), shallow=False), f


@pytest.mark.parametrize("paired,L,n", [(True, 150, 12000), (False, 150, 12000), (True, 700, 4000), (False, 400, 4000)])
def test_cli_contaminants_match_reference_binary(paired, L, n, tmp_path):
    """config keys contam1/contam2/ctMatchR + global_contams/glob_cotm_mR/glob_cotm_mM; reads of 400 / 700
    positions: the block-wise bit paths on the long-read plane store (snk_long_contam_kernel)"""
    from cases import CT1, CT2, GC1, plant_contams
    kw = dict(contam1=CT1 + ",GGGGGGGGGGGGGGGGGGGGGGGG", contam2=CT2 + ",CCCCCCCCCCCCCCCCCCCCCC", global_contams=GC1)
    d = synth.make_batch(n, L, paired=paired, seed=93)
    plant_contams(d, kw)
    cfg = ["contam1=" + kw["contam1"], "ctMatchR=0.6,0.7", "global_contams=" + GC1, "glob_cotm_mR=0.4", "glob_cotm_mM=1"]
    if paired:
        cfg.append("contam2=" + kw["contam2"])
    cli = ["-f", synth.ADAPTER1, "-J"] + (["-r", synth.ADAPTER2] if paired else [])
    case = ("contam", paired, L, n, 3, 250, {}, {}, cli, cfg)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False)
    for f in (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE):
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in (["c1.fq", "c2.fq"] if paired else ["c1.fq"]):
        assert _cat(os.path.join(ours, c)) == _cat(os.path.join(ref, c)), c
    txt = open(os.path.join(ours, "Statistics_of_Filtered_Reads.txt")).read()
    assert "Reads with contam sequence" in txt and "Reads with global contam sequence" in txt


@pytest.mark.parametrize("paired", [True, False])
def test_cli_trim_outputs(paired, tmp_path):
    """config keys trimFq1/trimFq2: every read after trimming, before the cascade; with pe_info the clean ids
    carry the suffix twice (preOutput ran twice on the object)."""
    n, L = 8000, 150
    d = synth.make_batch(n, L, paired=paired, seed=94)
    cfg = ["trimFq1=t1.fq.gz", "pe_info"] + (["trimFq2=t2.fq.gz", "trimBadTail=20,30"] if paired else [])
    cli = ["-f", synth.ADAPTER1, "-J", "-t", "2,1,0,3" if paired else "2,1"] + (["-r", synth.ADAPTER2] if paired else [])
    case = ("trimout", paired, L, n, 2, 300, {}, {}, cli, cfg)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False)
    for f in (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE):
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in (["c1.fq", "c2.fq", "t1.fq.gz", "t2.fq.gz"] if paired else ["c1.fq", "t1.fq.gz"]):
        a = _cat(os.path.join(ours, c))
        assert a == _cat(os.path.join(ref, c)), c
        if c.startswith("t"):
            assert a.count(b"\n") == 4 * n


@pytest.mark.parametrize("kind,cfgline", [("tile_old", "tile=1102"), ("tile_old", "tile=1101,1104"),
                                          ("tile_new", "tile=1103,1101"), ("fov", "fov=C002R003"), ("fov", "fov=C001R002,C004R001")])
def test_cli_tile_and_fov_filters(kind, cfgline, tmp_path):
    """config keys tile / fov: reads are dropped by name (src/read_filter.cpp:14-150, src/sequence.cpp:213-231),
    (range elements "a-b" make the reference binary crash: not compared)."""
    n, L = 4000, 100
    d = synth.make_batch(n, L, paired=True, seed=73)
    work = str(tmp_path)
    for m in (1, 2):
        with open(os.path.join(work, f"r{m}.fq"), "wb") as f:
            for i in range(n):
                t = 1101 + i % 5
                if kind == "tile_old":
                    rid = b"@FCD1PB1ACXX:4:%d:%d:2201#GAAGCACG/%d" % (t, i, m)
                elif kind == "tile_new":
                    rid = b"@HISEQ:310:C5MH9ANXX:1:%d:%d:2043 %d:N:0:TCGGTCAC" % (t, i, m)
                else:
                    rid = b"@CL100012345L1C%03dR%03d_%d/%d" % (1 + i % 4, 1 + i % 3, i, m)
                f.write(rid + b"\n" + d["seq"][m - 1][i, :L].tobytes() + b"\n+\n" + d["qual"][m - 1][i, :L].tobytes() + b"\n")
        subprocess.check_call(["gzip", "-1", "-f", "-k", os.path.join(work, f"r{m}.fq")])
    open(os.path.join(work, "cfg"), "w").write(cfgline + ("\nseqType=1" if kind == "tile_new" else "") + "\npatch=300\n")
    tail = ["-C", "c1.fq", "-D", "c2.fq", "-T", "2", "-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-c", os.path.join(work, "cfg")]
    r = subprocess.run([T.REF_BIN, "filter", "-1", os.path.join(work, "r1.fq.gz"), "-2", os.path.join(work, "r2.fq.gz"), "-o", os.path.join(work, "ref")] + tail, capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    r = subprocess.run([CLI, "filter", "-1", os.path.join(work, "r1.fq"), "-2", os.path.join(work, "r2.fq"), "-o", os.path.join(work, "ours")] + tail, capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    for c in ("c1.fq", "c2.fq"):
        assert _cat(os.path.join(work, "ours", c)) == _cat(os.path.join(work, "ref", c)), c
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(work, "ours", f), os.path.join(work, "ref", f), shallow=False), f


def test_cli_error_surface(tmp_path):
    r = subprocess.run([CLI, "filter", "-1", "/nonexistent.fq", "-C", "c.fq", "-o", str(tmp_path)], capture_output=True)
    assert r.returncode == 1 and r.stderr.startswith(b"Error:")
    r = subprocess.run([CLI, "filter", "-v"], capture_output=True)
    assert r.returncode == 1 and b"2.1.9" in r.stderr
    d = synth.make_batch(100, 50, paired=False, seed=1)
    d["seq"][0][7, 3] = ord("#")
    synth.write_fastq(str(tmp_path / "bad.fq"), d["seq"][0], d["qual"][0], 50, 1)
    r = subprocess.run([CLI, "filter", "-1", str(tmp_path / "bad.fq"), "-C", "c.fq", "-o", str(tmp_path / "o")], capture_output=True)
    assert r.returncode == 1 and b"Error:unrecognized sequence" in r.stderr


def _limited_run(tmp_path, paired, n, cfg_lines, extra_cli, seed):
    """Both binaries on the same .gz input with .gz clean output names (the reference insists on .gz there,
    src/process_argv.cpp:614-622); returns (ours_dir, ref_dir, report file list)."""
    L = 100
    d = synth.make_batch(n, L, paired=paired, seed=seed)
    work = str(tmp_path)
    mates = 2 if paired else 1
    for m in range(mates):
        synth.write_fastq(os.path.join(work, f"r{m + 1}.fq"), d["seq"][m], d["qual"][m], L, m + 1)
        subprocess.check_call(["gzip", "-1", "-f", "-k", os.path.join(work, f"r{m + 1}.fq")])
    open(os.path.join(work, "cfg"), "w").write("\n".join(cfg_lines) + "\n")
    tail = ["-C", "c1.fq.gz", "-T", "2", "-f", synth.ADAPTER1, "-J", "-c", os.path.join(work, "cfg")] + extra_cli
    inp = ["-1", os.path.join(work, "r1.fq.gz")]
    if paired:
        tail += ["-D", "c2.fq.gz", "-r", synth.ADAPTER2]
        inp += ["-2", os.path.join(work, "r2.fq.gz")]
    r = subprocess.run([T.REF_BIN, "filter"] + inp + ["-o", os.path.join(work, "ref")] + tail, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr[-300:]
    r = subprocess.run([CLI, "filter"] + inp + ["-o", os.path.join(work, "ours")] + tail, capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    return os.path.join(work, "ours"), os.path.join(work, "ref"), (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE)


@pytest.mark.parametrize("paired,split,how", [(True, 700, "-w"), (False, 900, "cfg"), (True, 1000, "cfg")])
def test_cli_clean_out_split(paired, split, how, tmp_path):
    """-w / cleanOutSplit: split.<k>.<cleanFq> files of that many reads, in input order (src/peprocess.cpp:2474-2560,
    2772-2870).  The block of a reference thread (patch * 160 / T reads) stays below the split size here: with
    bigger blocks the reference binary spins forever in extractReadsToFile (negative head count)."""
    cfg = ["patch=10"] + ([f"cleanOutSplit={split}"] if how == "cfg" else [])
    ours, ref, reports = _limited_run(tmp_path, paired, 3000, cfg, ["-w", str(split)] if how == "-w" else [], seed=61)
    names = sorted(f for f in os.listdir(ref) if f.startswith("split."))
    assert names and names == sorted(f for f in os.listdir(ours) if f.startswith("split."))
    for f in names:
        assert _cat(os.path.join(ours, f)) == _cat(os.path.join(ref, f)), f
    assert not os.path.exists(os.path.join(ours, "c1.fq.gz"))
    for f in reports:
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f


@pytest.mark.parametrize("paired,value", [(True, "1000head"), (False, "700head"), (True, "500"), (True, "0.25"), (False, "0.4"),
                                          (True, "100000head"), (True, "2600")])
def test_cli_total_reads_num(paired, value, tmp_path):
    """config key totalReadsNum: "<N>head" keeps the first N clean reads (src/peprocess.cpp:2960-2985); a number or a
    ratio keeps every k-th clean read in a second pass and leaves the complete file as total.<cleanFq> (:3198-3320;
    nothing happens when fewer than 1.1 x the wanted reads are there).  Statistics cover the whole input."""
    ours, ref, reports = _limited_run(tmp_path, paired, 3000, ["patch=10", f"totalReadsNum={value}"], [], seed=62)
    names = sorted(f for f in os.listdir(ref) if f.endswith(".fq.gz"))
    assert names == sorted(f for f in os.listdir(ours) if f.endswith(".fq.gz"))
    for f in names:
        assert _cat(os.path.join(ours, f)) == _cat(os.path.join(ref, f)), f
    for f in reports:
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f


def test_cli_limited_output_errors(tmp_path):
    base = [CLI, "filter", "-1", "/nonexistent.fq", "-2", "/nonexistent2.fq", "-o", str(tmp_path)]
    r = subprocess.run(base + ["-C", "c1.fq", "-D", "c2.fq", "-w", "100000"], capture_output=True)
    assert r.returncode == 1 and b"non-gz format when clean output reads are limited" in r.stderr
    r = subprocess.run(base + ["-C", "c1.fq.gz", "-D", "c2.fq.gz", "-w", "10"], capture_output=True)
    assert r.returncode == 1 and b"should be more than patch size" in r.stderr
    r = subprocess.run(base + ["-C", "c1.fq.gz", "-D", "c2.fq.gz", "-w", "abc"], capture_output=True)
    assert r.returncode == 1 and b"-w value should be a positive integer" in r.stderr


@pytest.mark.parametrize("value", ["T2C", "GTOa", "ATOA"])
def test_cli_base_convert(value, tmp_path):
    """config key baseConvert (PE): preOutput rewrites one letter in every clean (and trimmed) read before the clean
    statistics count it (src/peprocess.cpp:1617-1647); single-end mode aborts the reference (src/seprocess.cpp:923)."""
    n, L = 6000, 100
    d = synth.make_batch(n, L, paired=True, seed=97)
    cfg = [f"baseConvert={value}", "trimFq1=t1.fq.gz", "trimFq2=t2.fq.gz"]
    cli = ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J"]
    case = ("bconv", True, L, n, 2, 300, {}, {}, cli, cfg)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False)
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in ["c1.fq", "c2.fq", "t1.fq.gz", "t2.fq.gz"]:
        assert _cat(os.path.join(ours, c)) == _cat(os.path.join(ref, c)), c
    r = subprocess.run([CLI, "filter", "-1", os.path.join(work, "r1.fq"), "-C", "c1.fq", "-o", os.path.join(work, "se"), "-c", os.path.join(work, "cfg")],
                       capture_output=True)
    assert r.returncode == 1 and b"single-end" in r.stderr


def test_cli_keys_without_effect_and_module_errors(tmp_path):
    """`overlap` / `mis` (dead in the reference: nothing sets reads_result.over_lapped), the stLFR keys, inputAsList and -E are
    accepted without effect; the sRNA adapter keys are an error of this module (src/process_argv.cpp:763-771)."""
    n, L = 3000, 100
    d = synth.make_batch(n, L, paired=True, seed=98)
    cfg = ["overlap=10", "mis=0.1", "tenX", "notCutNoLFR", "inputAsList"]
    cli = ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-E", "nothing.fa"]
    case = ("noeffect", True, L, n, 2, 300, {}, {}, cli, cfg)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False)
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in ["c1.fq", "c2.fq"]:
        assert _cat(os.path.join(ours, c)) == _cat(os.path.join(ref, c)), c
    open(os.path.join(work, "cfg2"), "w").write("adaRCtg=3\nadaRMm=2\n")
    for exe in (T.REF_BIN, CLI):
        r = subprocess.run([exe, "filter", "-1", os.path.join(work, "r1.fq"), "-2", os.path.join(work, "r2.fq"), "-C", "c1.fq", "-D", "c2.fq",
                            "-o", os.path.join(work, "bad"), "-c", os.path.join(work, "cfg2")], capture_output=True)
        assert r.returncode == 1 and b"Error:these parameters should not appear in the module,-S|--adaRCtg,-b|--adaRMm" in r.stderr, (exe, r.stderr[-200:])


@pytest.mark.parametrize("paired,threads,patch,extra_cfg", [(True, 1, 500, []), (False, 1, 400, []), (True, 2, 50, ["pe_info", "outQualSys=1"]),
                                                            (True, 1, 700, ["trimBadTail=20,30"])])
def test_cli_streaming(paired, threads, patch, extra_cfg, tmp_path):
    """-j / --streaming: every patch's clean reads (">+<TAB>id<TAB>mate<TAB>seq<TAB>qual") and the cumulative statistics of
    its thread go to stdout (src/peprocess.cpp:1952-1976,3398-3418,3485-3594; seprocess.cpp:2405-2462), the clean files
    stay empty, the report files are written as usual.  One reference thread, or input inside one thread block: the
    reference's stdout is then deterministic and must match byte for byte."""
    n, L = 3000, 100
    d = synth.make_batch(n, L, paired=paired, seed=99)
    work = str(tmp_path)
    mates = 2 if paired else 1
    for m in range(mates):
        synth.write_fastq(os.path.join(work, f"r{m + 1}.fq"), d["seq"][m], d["qual"][m], L, m + 1)
        subprocess.check_call(["gzip", "-1", "-f", "-k", os.path.join(work, f"r{m + 1}.fq")])
    open(os.path.join(work, "cfg"), "w").write("\n".join([f"patch={patch}"] + extra_cfg) + "\n")
    tail = ["-C", "c1.fq.gz", "-T", str(threads), "-f", synth.ADAPTER1, "-J", "-j", "-c", os.path.join(work, "cfg")]
    inp = ["-1", os.path.join(work, "r1.fq.gz")]
    if paired:
        tail += ["-D", "c2.fq.gz", "-r", synth.ADAPTER2]
        inp += ["-2", os.path.join(work, "r2.fq.gz")]
    ref = subprocess.run([T.REF_BIN, "filter"] + inp + ["-o", os.path.join(work, "ref")] + tail, capture_output=True, timeout=120)
    assert ref.returncode == 0, ref.stderr[-300:]
    ours = subprocess.run([CLI, "filter"] + inp + ["-o", os.path.join(work, "ours")] + tail, capture_output=True)
    assert ours.returncode == 0, ours.stderr[-300:]
    assert len(ref.stdout) > 100000
    if ours.stdout != ref.stdout:
        a, b = ours.stdout.split(b"\n"), ref.stdout.split(b"\n")
        bad = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
        raise AssertionError((len(a), len(b), bad, a[bad][:200] if bad < len(a) else None, b[bad][:200] if bad < len(b) else None))
    for f in (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE):
        assert filecmp.cmp(os.path.join(work, "ours", f), os.path.join(work, "ref", f), shallow=False), f
    for c in (["c1.fq.gz", "c2.fq.gz"] if paired else ["c1.fq.gz"]):
        assert _cat(os.path.join(work, "ours", c)) == b"" == _cat(os.path.join(work, "ref", c))


def test_cli_streaming_many_threads(tmp_path):
    """-j with several reference threads and several blocks per thread: the reference interleaves its threads' patches
    by timing, so only the set of read lines, the number of statistics dumps and the report files are comparable."""
    n, L = 4000, 100
    d = synth.make_batch(n, L, paired=True, seed=101)
    work = str(tmp_path)
    for m in range(2):
        synth.write_fastq(os.path.join(work, f"r{m + 1}.fq"), d["seq"][m], d["qual"][m], L, m + 1)
        subprocess.check_call(["gzip", "-1", "-f", "-k", os.path.join(work, f"r{m + 1}.fq")])
    open(os.path.join(work, "cfg"), "w").write("patch=8\n")            # T=4: blocks of 8 * 40 = 320 pairs
    tail = ["-C", "c1.fq.gz", "-D", "c2.fq.gz", "-T", "4", "-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-j", "-c", os.path.join(work, "cfg")]
    inp = ["-1", os.path.join(work, "r1.fq.gz"), "-2", os.path.join(work, "r2.fq.gz")]
    ours = subprocess.run([CLI, "filter"] + inp + ["-o", os.path.join(work, "ours")] + tail, capture_output=True)
    assert ours.returncode == 0, ours.stderr[-200:]
    a = ours.stdout.split(b"\n")
    mine = sorted(x for x in a if x.startswith(b">+"))
    # The reference's four threads write to stdout without a lock: in about four runs of ten on the build container lines come out TORN
    # (a read line cut by another thread's statistics row -- found in round 6, tools output in DESIGN 6.1: this CLI's lines are the same
    # in every run, the reference's are not).  So the reference runs until one of its outputs is whole (three tries), a torn
    # run still has to be a subset of ours line by line, and when all three are torn its ONE-thread run -- deterministic -- decides the set.
    import collections
    for attempt in range(3):
        subprocess.call(["rm", "-rf", os.path.join(work, "ref")])
        ref = subprocess.run([T.REF_BIN, "filter"] + inp + ["-o", os.path.join(work, "ref")] + tail, capture_output=True, timeout=120)
        assert ref.returncode == 0, ref.stderr[-200:]
        b = ref.stdout.split(b"\n")
        theirs = sorted(x for x in b if x.startswith(b">+"))
        if theirs == mine:
            break
        whole = [x for x in theirs if x.count(b"\t") == 4 and len(x.split(b"\t")[3]) == len(x.split(b"\t")[4])]
        assert not (collections.Counter(whole) - collections.Counter(mine)), "a whole line of the reference that this CLI did not print"
    if theirs != mine:
        # torn three times out of three (a loaded box tears more): the SET of read lines does not depend on the reference's thread count, and
        # with one thread its stdout is deterministic -- that run decides; the four-thread run above still supplies the reports and the
        # number of statistics dumps
        one = subprocess.run([T.REF_BIN, "filter"] + inp + ["-o", os.path.join(work, "ref1")] + [("1" if x == "4" and tail[i - 1] == "-T" else x) for i, x in enumerate(tail)],
                             capture_output=True, timeout=120)
        assert one.returncode == 0, one.stderr[-200:]
        assert sorted(x for x in one.stdout.split(b"\n") if x.startswith(b">+")) == mine
    # (counted in the reference's raw bytes: a torn line still holds the marker, which one `<<` prints whole)
    assert a.count(b"#Total_statistical_information") == ref.stdout.count(b"#Total_statistical_information") == n // 8
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(work, "ours", f), os.path.join(work, "ref", f), shallow=False), f


# ---- round 2: several devices, capacity regrowth, the quality-system sanity check

def _write_fastq(path, seqs, quals, mate):
    with open(path, "wb") as f:
        for i, (s, q) in enumerate(zip(seqs, quals)):
            f.write(b"@SNK:1:%d:%d:%d/%d\n" % (1101 + i % 7, i, i * 3, mate) + s + b"\n+\n" + q + b"\n")
    subprocess.check_call(["gzip", "-1", "-f", "-k", path])


def _compare_dirs(ours, ref, paired, gz_ours=False):
    for f in (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE):
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in (["c1.fq", "c2.fq"] if paired else ["c1.fq"]):
        assert _cat(os.path.join(ours, c + (".gz" if gz_ours else ""))) == _cat(os.path.join(ref, c)), c


@pytest.mark.parametrize("rmdup", [False, True])
def test_cli_two_device_slots(rmdup, tmp_path):
    """--devices a,b: batches go round the devices, every device keeps one accumulator per virtual thread, the blocks
    are merged at the end (RCCL all-reduce between distinct GPUs; the same GPU listed twice -- all a 1-GPU box can
    offer -- merges on the host), the writer keeps input order.  Output identical to the reference binary."""
    import torch
    devs = "0,1" if torch.cuda.device_count() >= 2 else "0,0"
    n, L, threads, patch = 30000, 150, 3, 250
    d = synth.make_batch(n, L, paired=True, seed=63)
    if rmdup:
        for m in range(2):
            d["seq"][m][20000:21000] = d["seq"][m][0:1000]
    cli = ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J", "-l", "10", "-q", "0.1"]
    case = ("devs", True, L, n, threads, patch, {}, {}, cli, ["rmdup"] if rmdup else [])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    name, paired, L, n, threads, patch, skw, pkw, cli, cfg = case
    cmd = [CLI, "filter", "-1", os.path.join(work, "r1.fq"), "-2", os.path.join(work, "r2.fq"), "-C", "c1.fq", "-D", "c2.fq",
           "-o", os.path.join(work, "ours"), "-T", str(threads), "--devices", devs]
    if os.path.exists(os.path.join(work, "cfg")):
        cmd += ["-c", os.path.join(work, "cfg")]
    env = dict(os.environ, SNK_BATCH_PAIRS="4096")                       # 8 batches: 4 per device slot set
    r = subprocess.run(cmd + cli, capture_output=True, env=env)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-500:])
    _compare_dirs(os.path.join(work, "ours"), ref, True)


@T.first_contact
@pytest.mark.parametrize("paired,gz_out,trim", [(True, False, False), (True, True, True), (False, False, True)])
def test_cli_sharded_ingest(paired, gz_out, trim, tmp_path):
    """SURVEY 8e / VERDICT r3 #5: plain input, several devices, SNK_SHARDED=1 -- one child process per device takes a contiguous
    range of records (cut at record boundaries by the parent's newline count), writes its own part files and dumps its statistics
    blocks; parts concatenated in rank order, blocks added up per virtual thread.  Same bytes as the single-device run and as the
    reference binary (--devices 0,0: two shards on the one GPU of this box)."""
    import torch
    devs = "0,1" if torch.cuda.device_count() >= 2 else "0,0"
    n, L, threads, patch = 40000, 150, 3, 250
    d = synth.make_batch(n, L, paired=paired, seed=64)
    cli = ["-f", synth.ADAPTER1, "-J", "-l", "10", "-q", "0.1"] + (["-r", synth.ADAPTER2] if paired else []) + (["-t", "3,4"] if trim and not paired else [])
    case = ("shard", paired, L, n, threads, patch, {}, {}, cli, ["trimBadTail=20,30"] if trim and paired else [])     # (trimming: clean lengths vary, the reports depend on the virtual threads)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ext = ".fq.gz" if gz_out else ".fq"
    cmd = [CLI, "filter", "-1", os.path.join(work, "r1.fq"), "-C", "c1" + ext, "-o", os.path.join(work, "ours"), "-T", str(threads), "--devices", devs]
    if paired:
        cmd += ["-2", os.path.join(work, "r2.fq"), "-D", "c2" + ext]
    if os.path.exists(os.path.join(work, "cfg")):
        cmd += ["-c", os.path.join(work, "cfg")]
    r = subprocess.run(cmd + cli, capture_output=True, env=dict(os.environ, SNK_SHARDED="1", SNK_BATCH_PAIRS="4096"))
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-800:])
    ours = os.path.join(work, "ours")
    assert b"sharded run: 2 shards" in open(os.path.join(ours, "log"), "rb").read()
    for f in (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE):
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in (["c1", "c2"] if paired else ["c1"]):
        assert _cat(os.path.join(ours, c + ext)) == _cat(os.path.join(ref, c + ".fq")), c
    assert not [x for x in os.listdir(ours) if ".part" in x or x.startswith("shard.")]       # nothing of the shards is left behind


@T.first_contact
@pytest.mark.parametrize("paired,n,gz_in", [(True, 40000, False), (False, 40100, False), (True, 40000, True), (False, 40100, True)])
def test_cli_sharded_rmdup_and_wire(paired, n, gz_in, tmp_path):
    """the sharded run with the collectives `north_star` names.  The shards form one communicator (RCCL between
    different GPUs; a host wire when the device list names one GPU twice -- all this box has -- or SNK_SHARD_WIRE=host says so),
    all-reduce their per-thread statistics blocks over it (rank 0 writes the reports, the parent checks them against the sum of
    the dumped blocks) and, with `rmdup`, send every hash to its owner (hash % G) with its global index and get the flag back:
    duplicates ACROSS the shard border, third copies, the SE flag shift across the border, dupReads side files -- all as the
    reference binary has them.  gz_in: `.gz` input -- the parent's scout pass decodes both files once, counts the records and notes a
    block header + window per border; every shard's decoder starts there (the two files of a pair at different places, cut at
    the same record)."""
    import torch
    devs = "0,1" if torch.cuda.device_count() >= 2 else "0,0"
    L, threads, patch = 150, 3, 250
    d = synth.make_batch(n, L, paired=paired, seed=65)
    h = n // 2
    for m in range(2 if paired else 1):
        d["seq"][m][h + 1000:h + 3000] = d["seq"][m][0:2000]          # second copies in the other shard
        d["seq"][m][h - 600:h - 100] = d["seq"][m][0:500]             # third copies, back in the first
        d["seq"][m][h - 3:h + 3] = d["seq"][m][100:106]               # duplicates on both sides of the border
        d["seq"][m][n - 40:n - 20] = d["seq"][m][700:720]             # ... and inside the last (partial) patch
    cli = ["-f", synth.ADAPTER1, "-J"] + (["-r", synth.ADAPTER2] if paired else [])
    case = ("shardrm", paired, L, n, threads, patch, {}, {}, cli, ["rmdup"])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ext = ".fq.gz" if gz_in else ".fq"
    ours = os.path.join(work, "ours with blanks" if gz_in else "ours")      # (.gz: the shards' window files live there, their paths travel through the environment)
    cmd = [CLI, "filter", "-1", os.path.join(work, "r1" + ext), "-C", "c1.fq", "-o", ours, "-T", str(threads), "--devices", devs,
           "-c", os.path.join(work, "cfg")]
    if paired:
        cmd += ["-2", os.path.join(work, "r2" + ext), "-D", "c2.fq"]
    r = subprocess.run(cmd + cli, capture_output=True, env=dict(os.environ, SNK_SHARDED="1", SNK_BATCH_PAIRS="4096", SNK_GZ_CHUNK="262144"))
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-800:])
    log = open(os.path.join(ours, "log"), "rb").read()
    assert b"sharded run: 2 shards" in log and b"statistics merged over" in log and b"Warning" not in r.stderr, (log[-600:], r.stderr[-400:])
    _compare_dirs(ours, ref, paired)
    ndup = 0
    for t in range(threads):
        for m in range(2 if paired else 1):
            f = f"dupReads.{t}.{m + 1}.gz"
            a, b = _cat(os.path.join(ours, f)), _cat(os.path.join(ref, f))
            assert a == b, f
            ndup += a.count(b"\n") // 4
    assert ndup > 2500 and (b"dup number:\t%d" % (ndup // (2 if paired else 1))) in log
    assert (b"totalReadsNum:\t%d" % n) in r.stdout
    assert not [x for x in os.listdir(ours) if ".part" in x or x.startswith("shard.")]


SHARDED_VARIANTS = ["trim_pe", "trim_se", "tile", "fov", "index", "base_convert", "pe_info_crlf", "contam", "long", "reports_0", "reports_3", "gz_gz"]


@T.first_contact
@pytest.mark.parametrize("which", SHARDED_VARIANTS)
def test_cli_variants_as_sharded_runs(which, monkeypatch, tmp_path):
    """round 5: what keeps a run from being sharded is only what counts reads across the whole input (-j, -w, totalReadsNum); the host
    formatter's variants -- trimFq1/2 (their parts joined like the clean files'), index removal, tile / fov, baseConvert, pe_info +
    outQualSys, contaminants, long reads, .gz in and out -- run in the shards.  This module's own tests as sharded runs: two shards
    on device 0 (listed twice: the host wire carries the collectives), whatever the size of the input."""
    for k, v in (("SNK_DEVICES", "0,0"), ("SNK_SHARDED", "1"), ("SNK_SHARD_MIN_RECORDS", "300"), ("SNK_GZ_CHUNK", "65536")):
        monkeypatch.setenv(k, v)
    if which == "trim_pe":
        test_cli_trim_outputs(True, tmp_path)
    elif which == "trim_se":
        test_cli_trim_outputs(False, tmp_path)
    elif which == "tile":
        test_cli_tile_and_fov_filters("tile_new", "tile=1103,1101", tmp_path)
    elif which == "fov":
        test_cli_tile_and_fov_filters("fov", "fov=C001R002,C004R001", tmp_path)
    elif which == "index":
        test_cli_index_removal("1", tmp_path)
    elif which == "base_convert":
        test_cli_base_convert("T2C", tmp_path)
    elif which == "pe_info_crlf":
        test_cli_pe_info_outqual_and_crlf(tmp_path)
    elif which == "contam":
        test_cli_contaminants_match_reference_binary(True, 150, 6000, tmp_path)
    elif which == "long":
        test_cli_long_reads(600, True, tmp_path)
    elif which.startswith("reports_"):
        test_cli_matches_reference_binary(R.REPORT_CASES[int(which[-1])], tmp_path)
    else:
        test_cli_gz_in_gz_out(tmp_path)
    logs = [os.path.join(dp, f) for dp, _, fs in os.walk(str(tmp_path)) for f in fs if f == "log"]
    assert any(b"sharded run: 2 shards" in open(x, "rb").read() for x in logs), "the run was not sharded"


@pytest.mark.parametrize("paired", [True, False])
def test_cli_longer_read_after_first_batch(paired, tmp_path):
    """The reference takes any read up to 1000 nt at any position; here the capacity comes from the first batch and is
    regrown (pipeline drained, statistics folded into the wider geometry) when a longer read shows up later."""
    rng = np.random.default_rng(9)
    n = 9000
    lens = np.concatenate([rng.integers(60, 101, 5000), rng.integers(60, 181, 2000), rng.integers(100, 301, 2000)])
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    work = str(tmp_path)
    for m in range(2 if paired else 1):
        seqs = [bytes(B[rng.integers(0, 4, l)]) for l in lens]
        quals = [bytes((33 + np.clip(rng.normal(34, 6, l), 2, 41)).astype(np.uint8)) for l in lens]
        _write_fastq(os.path.join(work, f"r{m + 1}.fq"), seqs, quals, m + 1)
    open(os.path.join(work, "cfg"), "w").write("patch=250\n")
    tail = ["-C", "c1.fq"] + (["-D", "c2.fq"] if paired else []) + ["-T", "2", "-l", "10", "-q", "0.3", "-c", os.path.join(work, "cfg")]
    ins = lambda ext: ["-1", os.path.join(work, "r1.fq" + ext)] + (["-2", os.path.join(work, "r2.fq" + ext)] if paired else [])
    r = subprocess.run([T.REF_BIN, "filter"] + ins(".gz") + ["-o", os.path.join(work, "ref")] + tail, capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    env = dict(os.environ, SNK_BATCH_PAIRS="2048")
    r = subprocess.run([CLI, "filter"] + ins("") + ["-o", os.path.join(work, "ours")] + tail, capture_output=True, env=env)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-500:])
    _compare_dirs(os.path.join(work, "ours"), os.path.join(work, "ref"), paired)


@pytest.mark.parametrize("paired", [True, False])
@pytest.mark.parametrize("shift,qualsys,expect", [(31, None, "Error"), (0, "1", "Error"), (0, None, None)])
def test_cli_quality_system_sanity(paired, shift, qualsys, expect, tmp_path):
    """stat_pe_fqs / stat_se_fqs check the quality system once, on the first patch (src/peprocess.cpp:1207-1319):
    Phred-64 data under qualSys=2, or Phred-33 data under qualSys=1 -> the same message and exit code as the reference."""
    n, L = 3000, 100
    d = synth.make_batch(n, L, paired=paired, seed=73)
    work = str(tmp_path)
    for m in range(2 if paired else 1):
        _write_fastq(os.path.join(work, f"r{m + 1}.fq"), [bytes(x) for x in d["seq"][m][:, :L]],
                     [bytes(x + shift) for x in d["qual"][m][:, :L]], m + 1)
    open(os.path.join(work, "cfg"), "w").write("patch=300\n" + (f"qualSys={qualsys}\n" if qualsys else ""))
    tail = ["-C", "c1.fq"] + (["-D", "c2.fq"] if paired else []) + ["-T", "2", "-c", os.path.join(work, "cfg")]
    ins = lambda ext: ["-1", os.path.join(work, "r1.fq" + ext)] + (["-2", os.path.join(work, "r2.fq" + ext)] if paired else [])
    a = subprocess.run([T.REF_BIN, "filter"] + ins(".gz") + ["-o", os.path.join(work, "ref")] + tail, capture_output=True)
    b = subprocess.run([CLI, "filter"] + ins("") + ["-o", os.path.join(work, "ours")] + tail, capture_output=True)
    msg = b"base quality seems abnormal,please check the quality system parameter or fastq file"
    if expect:
        assert a.returncode == 1 and b.returncode == 1, (a.returncode, b.returncode, a.stderr[-200:], b.stderr[-200:])
        assert (expect.encode() + b":" + msg) in a.stderr and (expect.encode() + b":" + msg) in b.stderr
    else:
        assert a.returncode == 0 and b.returncode == 0
        assert msg not in a.stderr and msg not in b.stderr


# ---- round 3: Phred-64 input, non-default maxBaseQuality (VERDICT r2 task 1 i)

@pytest.mark.parametrize("paired,cfg,qmax", [(True, ["qualSys=1"], 41), (True, ["qualSys=1", "outQualSys=1"], 41),
                                             (False, ["qualSys=1", "maxBaseQuality=44"], 43), (True, ["maxBaseQuality=40"], 39),
                                             (True, ["maxBaseQuality=50", "outQualSys=1"], 49)])
def test_cli_phred64_and_max_base_quality(paired, cfg, qmax, tmp_path):
    """`qualSys=1` on Phred-64 files (+- `outQualSys=1`) and `maxBaseQuality` != 42: HIP kernels behind the CLI, every report
    file and the clean FASTQ (qualities re-based by the writer, src/peprocess.cpp:3398-3405) against the reference binary.
    (SE takes an even maxBaseQuality: seProcess's column scan reads position_qual[i][maxBaseQuality], one word past the row
    -- src/seprocess.cpp:275 vs global_variable.cpp:43 -- which for odd values is the next heap chunk's size field: quirk Q12)"""
    from cases import rebase_quality
    phred = 64 if "qualSys=1" in cfg else 33
    mbq = [int(c.split("=")[1]) for c in cfg if c.startswith("maxBaseQuality")]
    kw = dict(adapters1=[synth.ADAPTER1], ada_trim=1, low_qual=10, low_qual_ratio=0.1, n_ratio=0.01, mean_quality=20,
              trim_bad_tail=(20, 30), quality_phred=phred, max_base_quality=mbq[0] if mbq else 42)
    cli = ["-f", synth.ADAPTER1, "-J", "-l", "10", "-q", "0.1", "-n", "0.01", "-m", "20"]
    if paired:
        kw.update(adapters2=[synth.ADAPTER2])
        cli += ["-r", synth.ADAPTER2]
    else:
        kw.pop("trim_bad_tail")                        # (SE: the reference's option check wants one field, Appendix A)
    case = ("phred", paired, 150, 30000, 3, 200, dict(seed=81), kw, cli, cfg + (["trimBadTail=20,30"] if paired else []))
    d, p = R.case_inputs(case)
    rebase_quality(d, phred, qmax, seed=qmax)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False)
    _compare_dirs(ours, ref, paired)


# ---- round 3: device-side FASTQ ingest / egress (include/snk_fastq.h) behind the CLI

@pytest.mark.parametrize("paired,gz,ragged", [(True, False, False), (False, True, False), (True, True, False), (True, False, True), (False, True, True)])
def test_cli_device_text_growing_names_and_ragged_end(paired, gz, ragged, tmp_path):
    """device-text mode on awkward text: read names that get much longer after the first batch (the slots' text buffers
    grow), read lengths that grow (capacity regrowth through the same path) -- against the reference binary, and the host
    formatter (SNK_HOST_TEXT=1) must write the very same bytes.  ragged: no newline behind the last line; the reference then
    drops the last quality character of the file with the line terminator it did not find (quality shorter than the
    sequence: it indexes past the string), both readers here keep it -- compared with each other only."""
    rng = np.random.default_rng(19)
    lens = np.concatenate([rng.integers(50, 91, 3000), rng.integers(50, 141, 3000), rng.integers(80, 201, 3000)])
    n = len(lens)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    work = str(tmp_path)
    for m in range(2 if paired else 1):
        with open(os.path.join(work, f"r{m + 1}.fq"), "wb") as f:
            for i, l in enumerate(lens):
                name = b"@SNK:%d:%d" % (i % 9, i) + (b":" + b"x" * (i // 40) if i >= 4000 else b"") + b"/%d" % (m + 1)
                q = bytes((33 + np.clip(rng.normal(34, 6, l), 2, 41)).astype(np.uint8))
                rec = name + b"\n" + bytes(B[rng.integers(0, 4, l)]) + b"\n+\n" + q + b"\n"
                f.write(rec[:-1] if (ragged and i == n - 1) else rec)
        subprocess.check_call(["gzip", "-1", "-f", "-k", os.path.join(work, f"r{m + 1}.fq")])
    open(os.path.join(work, "cfg"), "w").write("patch=250\n" + ("pe_info\n" if paired else "") + "outQualSys=1\n")
    ext = ".gz" if gz else ""
    tail = ["-C", "c1.fq" + ext] + (["-D", "c2.fq" + ext] if paired else []) + ["-T", "2", "-l", "10", "-q", "0.3", "-c", os.path.join(work, "cfg")]
    ins = lambda e: ["-1", os.path.join(work, "r1.fq" + e)] + (["-2", os.path.join(work, "r2.fq" + e)] if paired else [])   # noqa: E731
    r = subprocess.run([T.REF_BIN, "filter"] + ins(".gz") + ["-o", os.path.join(work, "ref")] + tail, capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    for out, extra in (("ours", {}), ("host", {"SNK_HOST_TEXT": "1"})):
        env = dict(os.environ, SNK_BATCH_PAIRS="1024", **extra)
        r = subprocess.run([CLI, "filter"] + ins(ext) + ["-o", os.path.join(work, out)] + tail, capture_output=True, env=env)
        assert r.returncode == 0, (out, r.stdout[-300:], r.stderr[-500:])
        other = "ours" if ragged else "ref"
        if out == other:
            continue
        for f in (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE):
            assert filecmp.cmp(os.path.join(work, out, f), os.path.join(work, other, f), shallow=False), (out, f)
        for c in (["c1.fq", "c2.fq"] if paired else ["c1.fq"]):
            assert _cat(os.path.join(work, out, c + ext)) == _cat(os.path.join(work, other, c + ext)), (out, c)


def test_cli_device_text_input_errors(tmp_path):
    """malformed input in device-text mode: the same messages as the host reader's"""
    work = str(tmp_path)
    good = b"".join(b"@r%d\nACGTACGTACGTACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" % i for i in range(3000))
    cases = {"mismatch": good.replace(b"@r1500\nACGT", b"@r1500\nACGTAC", 1), "truncated": good[:len(good) - 40] }
    for name, blob in cases.items():
        open(os.path.join(work, name + ".fq"), "wb").write(blob)
        for extra in ({}, {"SNK_HOST_TEXT": "1"}):
            r = subprocess.run([CLI, "filter", "-1", os.path.join(work, name + ".fq"), "-C", "c.fq", "-o", os.path.join(work, "o_" + name)],
                               capture_output=True, env=dict(os.environ, SNK_BATCH_PAIRS="512", **extra))
            assert r.returncode == 1, (name, extra, r.stderr[-200:])
            assert (b"lengths differ" in r.stderr) if name == "mismatch" else (b"truncated" in r.stderr), (name, extra, r.stderr[-200:])


def test_cli_streaming_across_a_capacity_regrowth_and_devices(tmp_path):
    """-j prints cumulative per-thread statistics; after a capacity regrowth (a read longer than the 256 positions -j
    starts with) the counts of the closed epoch must still be in them -- stdout byte-identical to the reference with one
    thread.  And -j with more than one device is refused (each device would only know its own patches)."""
    rng = np.random.default_rng(29)
    lens = np.concatenate([rng.integers(60, 101, 1500), rng.integers(200, 301, 1500)])
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    work = str(tmp_path)
    seqs = [bytes(B[rng.integers(0, 4, l)]) for l in lens]
    quals = [bytes((33 + np.clip(rng.normal(34, 6, l), 2, 41)).astype(np.uint8)) for l in lens]
    _write_fastq(os.path.join(work, "r1.fq"), seqs, quals, 1)
    open(os.path.join(work, "cfg"), "w").write("patch=300\n")
    tail = ["-C", "c1.fq.gz", "-T", "1", "-l", "10", "-q", "0.3", "-j", "-c", os.path.join(work, "cfg")]
    inp = ["-1", os.path.join(work, "r1.fq.gz")]
    ref = subprocess.run([T.REF_BIN, "filter"] + inp + ["-o", os.path.join(work, "ref")] + tail, capture_output=True, timeout=120)
    ours = subprocess.run([CLI, "filter"] + inp + ["-o", os.path.join(work, "ours")] + tail, capture_output=True)
    assert ref.returncode == 0 and ours.returncode == 0, (ref.stderr[-200:], ours.stderr[-300:])
    if ours.stdout != ref.stdout:
        a, b = ours.stdout.split(b"\n"), ref.stdout.split(b"\n")
        bad = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
        raise AssertionError((len(a), len(b), bad, a[bad][:200] if bad < len(a) else None, b[bad][:200] if bad < len(b) else None))
    for f in R.REPORT_FILES_SE:
        assert filecmp.cmp(os.path.join(work, "ours", f), os.path.join(work, "ref", f), shallow=False), f
    r = subprocess.run([CLI, "filter"] + inp + ["-o", os.path.join(work, "two"), "--devices", "0,0"] + tail, capture_output=True)
    assert r.returncode == 1 and b"-j/--streaming runs on one device" in r.stderr, r.stderr[-200:]


def test_cli_adapter_list_files_of_any_length(tmp_path):
    """-f / -r given as list files (src/process_argv.cpp:242-304) with more adapters than fit snk_params.adapters[]: the
    lists travel through snk_params.adapter_list to the tiled kernel; reports and clean FASTQ against the reference binary"""
    rng = np.random.default_rng(77)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    lists = [[bytes(B[rng.integers(0, 4, int(rng.integers(12, 50)))]).decode() for _ in range(k)] for k in (23, 37)]
    lists[0][5] = synth.ADAPTER1
    lists[1][30] = synth.ADAPTER2
    n, L = 20000, 150
    d = synth.make_batch(n, L, paired=True, seed=123)
    for m in range(2):                                         # some reads carry adapters from the middle of the lists
        rows = rng.choice(n, 3000, replace=False)
        for r in rows:
            a = np.frombuffer(lists[m][int(rng.integers(0, len(lists[m])))].encode(), dtype=np.uint8)
            p = int(rng.integers(20, L - len(a)))
            d["seq"][m][r, p:p + len(a)] = a
    work = str(tmp_path)
    for m in range(2):
        open(os.path.join(work, f"ada{m + 1}.list"), "w").write("\n".join(lists[m]) + "\n")
    case = ("adalist", True, L, n, 3, 250, {}, {}, ["-f", os.path.join(work, "ada1.list"), "-r", os.path.join(work, "ada2.list"), "-J", "-l", "10", "-q", "0.2"], [])
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False)
    _compare_dirs(ours, ref, True)


def test_cli_device_gzip_and_its_host_fallback(tmp_path):
    """.gz output of device-text mode: gzip members made on the device (default), by the host encoder (SNK_HOST_DEFLATE=1), and the
    fallback when the members do not fit their buffer (forced with SNK_GZ_CAP_TEST=1) -- `gzip -t` accepts all three (CRC-32 and
    ISIZE of every member), the decompressed bytes are the reference's."""
    case = R.REPORT_CASES[1]
    d, p = R.case_inputs(case)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    name, paired, L, n, threads, patch, skw, pkw, cli, cfg = case
    sizes = {}
    for tag, extra in (("dev", {}), ("host", {"SNK_HOST_DEFLATE": "1"}), ("fallback", {"SNK_GZ_CAP_TEST": "1"})):
        out = os.path.join(work, tag)
        cmd = [CLI, "filter", "-1", os.path.join(work, "r1.fq.gz"), "-2", os.path.join(work, "r2.fq.gz"), "-C", "c1.fq.gz", "-D", "c2.fq.gz", "-o", out,
               "-T", str(threads), "-c", os.path.join(work, "cfg")]
        r = subprocess.run(cmd + cli, capture_output=True, env=dict(os.environ, SNK_BATCH_PAIRS="8192", **extra))
        assert r.returncode == 0, (tag, r.stderr[-400:])
        for c in ("c1.fq", "c2.fq"):
            assert subprocess.run(["gzip", "-t", os.path.join(out, c + ".gz")]).returncode == 0, (tag, c)
            assert _cat(os.path.join(out, c + ".gz")) == _cat(os.path.join(ref, c)), (tag, c)
        sizes[tag] = os.path.getsize(os.path.join(out, "c1.fq.gz"))
    assert sizes["dev"] < 1.10 * sizes["host"], sizes


@pytest.mark.first_contact
@pytest.mark.parametrize("case", [R.REPORT_CASES[1], R.REPORT_CASES[4]], ids=lambda c: c[0])
def test_cli_proven_only_switch(case, tmp_path):
    """SNK_PROVEN_ONLY=1 (csrc/snk_tables.h, csrc/snk_filter.cpp): the run's automatic dispatch takes the generic kernel + the LDS
    histogram kernel -- the device sources of the last hardware-green record -- instead of the tiled kernel; same bytes as the
    reference binary either way.  With a red first contact of the rewritten kernels this is the A/B inside one library."""
    d, p = R.case_inputs(case)
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = _run_ours(case, work, gz=False, env={"SNK_PROVEN_ONLY": "1"})
    for f in (R.REPORT_FILES_PE if case[1] else R.REPORT_FILES_SE):
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in (["c1.fq", "c2.fq"] if case[1] else ["c1.fq"]):
        assert _cat(os.path.join(ours, c)) == _cat(os.path.join(ref, c)), c


@pytest.mark.first_contact
def test_cli_sharded_gz_index(tmp_path):
    """SNK_GZ_INDEX_DIR: the scout pass of a sharded run over `.gz` input -- one full decode in front of the run -- leaves its
    borders (block header, window, record number per shard border) in a small index named after the input's size and modification time;
    the next run over the same files reads them instead ("no decode in front of the run" in the log), a changed file is scouted again.
    The reference binary's bytes each time."""
    import torch
    devs = "0,1" if torch.cuda.device_count() >= 2 else "0,0"
    n, L = 24000, 150
    d = synth.make_batch(n, L, paired=True, seed=68)
    for m in range(2):
        d["seq"][m][n // 2 + 100:n // 2 + 900] = d["seq"][m][0:800]
    cli = ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J"]
    case = ("gzindex", True, L, n, 2, 250, {}, {}, cli, ["rmdup"])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    idx = os.path.join(work, "index dir")
    os.makedirs(idx)
    env = dict(os.environ, SNK_SHARDED="1", SNK_BATCH_PAIRS="4096", SNK_GZ_CHUNK="262144", SNK_GZ_INDEX_DIR=idx, SNK_SHARD_MIN_RECORDS="1000")

    def run(tag):
        ours = os.path.join(work, tag)
        cmd = [CLI, "filter", "-1", os.path.join(work, "r1.fq.gz"), "-2", os.path.join(work, "r2.fq.gz"), "-C", "c1.fq", "-D", "c2.fq", "-o", ours, "-T", "2",
               "--devices", devs, "-c", os.path.join(work, "cfg")] + cli
        r = subprocess.run(cmd, capture_output=True, env=env, timeout=170)
        assert r.returncode == 0, (r.stdout[-300:], r.stderr[-800:])
        _compare_dirs(ours, ref, True)
        return open(os.path.join(ours, "log"), "rb").read()
    log = run("first")
    assert b"scout pass over the .gz input" in log and b"sharded run: 2 shards" in log
    made = sorted(os.listdir(idx))
    assert len(made) == 2 and all(f.endswith(".snkidx") for f in made), made
    log = run("second")
    assert b"from the index in SNK_GZ_INDEX_DIR (no decode in front of the run)" in log and b"scout pass over" not in log and b"sharded run: 2 shards" in log
    # a file that changed (here: only its modification time) is scouted again, and a cut-off index is not trusted
    st = os.stat(os.path.join(work, "r1.fq.gz"))
    os.utime(os.path.join(work, "r1.fq.gz"), ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000_000))
    with open(os.path.join(idx, [f for f in made if f.startswith("r2")][0]), "r+b") as f:
        f.truncate(40000)
    log = run("third")
    assert b"scout pass over the .gz input" in log
    assert len(os.listdir(idx)) == 3                       # (r1's new index next to its old one, r2's rewritten)


@pytest.mark.first_contact
@pytest.mark.parametrize("G,paired,gz_in", [(3, True, True), (4, True, False), (5, False, True)])
def test_cli_more_than_two_shards(G, paired, gz_in, tmp_path):
    """Three, four and five shards (one device listed G times: the host wire carries the collectives -- since round 6 piece by piece
    through rank 0, a peer sending from a thread of its own while it receives, host/snk_wire.h) with `rmdup`: duplicates across EVERY
    shard border, owners = hash % G, the single-end flag shift across the borders; plain and `.gz` input (G - 1 scout borders).
    Reports, clean FASTQ and the duplicate side files are the reference binary's."""
    n, L, threads = 40000, 150, 3
    d = synth.make_batch(n, L, paired=paired, seed=65)
    for m in range(2 if paired else 1):
        for g in range(1, G):
            h = n * g // G
            d["seq"][m][h + 200:h + 1200] = d["seq"][m][0:1000]
            d["seq"][m][h - 3:h + 3] = d["seq"][m][100:106]
    cli = ["-f", synth.ADAPTER1, "-J"] + (["-r", synth.ADAPTER2] if paired else [])
    case = ("shards%d" % G, paired, L, n, threads, 250, {}, {}, cli, ["rmdup"])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ext = ".fq.gz" if gz_in else ".fq"
    ours = os.path.join(work, "ours")
    cmd = [CLI, "filter", "-1", os.path.join(work, "r1" + ext), "-C", "c1.fq", "-o", ours, "-T", str(threads), "--devices", ",".join(["0"] * G),
           "-c", os.path.join(work, "cfg")]
    if paired:
        cmd += ["-2", os.path.join(work, "r2" + ext), "-D", "c2.fq"]
    r = subprocess.run(cmd + cli, capture_output=True, timeout=170,
                       env=dict(os.environ, SNK_SHARDED="1", SNK_BATCH_PAIRS="4096", SNK_GZ_CHUNK="131072", SNK_SHARD_MIN_RECORDS="1000"))
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-800:])
    log = open(os.path.join(ours, "log"), "rb").read()
    assert (b"sharded run: %d shards" % G) in log and (b"statistics merged over the host wire (%d shards)" % G) in log, log[-600:]
    _compare_dirs(ours, ref, paired)
    for t in range(threads):
        for m in range(2 if paired else 1):
            f = f"dupReads.{t}.{m + 1}.gz"
            assert _cat(os.path.join(ours, f)) == _cat(os.path.join(ref, f)), f
