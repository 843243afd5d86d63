"""Random `SOAPnuke filter` command lines, this repo's CLI (on the emulated library, tests/simt) against the compiled reference
binary on the same files: random combinations of the options the tests of tests/test_cli_gpu.py take one at a time -- adapters
(README / random, trim or discard, adaMis / adaMR / adaEdge), quality and content thresholds, hard trims, low-quality-end trims,
minimum length, rmdup, contaminants, pe_info, PE / SE, fixed and variable read lengths, thread and patch counts -- crossed with
this CLI's own modes: plain / .gz input and output, inflate on the device, small batches, the host-text pipeline, two slot sets,
sharded ingest.  All report files, the clean FASTQ and the duplicate side files must be the reference's bytes."""
import filecmp
import gzip
import os
import subprocess

import numpy as np
import pytest

import report_util as R
import simt_lib as S
import snk_testlib as T
from soapnuke_amd import synth

N_CONTEXTS = int(os.environ.get("SNK_FUZZ_CONTEXTS", "40"))      # (a longer one-off run: SNK_FUZZ_CONTEXTS=200 SNK_SIMT_FULL=1)
CORE = ["test_random_command_lines[3]", "test_random_command_lines[11]", "test_random_command_lines[24]"]
pytestmark = pytest.mark.skipif(not os.path.exists(T.REF_BIN), reason="oracle/_ref/SOAPnuke not built")
B4 = np.frombuffer(b"ACGT", dtype=np.uint8)


def _cat(path):
    return gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()


def _context(i):
    rng = np.random.default_rng(77000 + i)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]       # noqa: E731
    paired = bool(rng.random() < 0.7)
    L = pick([100, 150, 150, 250, 600] if i % 8 else [150])
    n = int(rng.integers(3000, 8000)) if L <= 250 else int(rng.integers(1500, 3000))
    var_len = bool(rng.random() < 0.5)
    if rng.random() < 0.7:
        ada = [synth.ADAPTER1, synth.ADAPTER2]
    else:
        ada = ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(10, 41)))) for _ in range(2)]
    d = synth.make_batch(n, L, paired=paired, var_len=var_len, seed=5000 + i, adapters=(ada[0], ada[1]), dimer_frac=pick([0.0, 0.0, 0.03]))
    for m in range(2 if paired else 1):
        if d["len"][m] is not None:                           # the reference's SE report takes the raw row count from the LAST read's length and prints
            j = int(np.argmax(d["len"][m]))                   # uninitialised floats behind it (Distribution_of_Q20_Q30...: src/seprocess.cpp): the last
            d["len"][m][-1] = d["len"][m][j]                  # read is one of the longest
            d["seq"][m][-1] = d["seq"][m][j]
            d["qual"][m][-1] = d["qual"][m][j]
    cli, cfg = ["-f", ada[0]] + (["-r", ada[1]] if paired else []), []
    if rng.random() < 0.6:
        cli.append("-J")
    cli += ["-l", str(pick([5, 10, 15, 20])), "-q", str(pick([0.1, 0.2, 0.3, 0.5]))]
    if rng.random() < 0.6:
        cli += ["-n", str(pick([0.01, 0.05, 0.1]))]
    if rng.random() < 0.4:
        cli += ["-m", str(pick([15, 20, 25]))]
    if rng.random() < 0.4:
        cli += ["-g", str(pick([5, 10, 15]))]
    if rng.random() < 0.4:
        cli += ["-X", str(pick([10, 30, 50]))]
    if rng.random() < 0.3:
        cli += ["-p", str(pick([0.5, 0.8]))]
    if rng.random() < 0.4:
        cli += ["-t", ",".join(str(int(x)) for x in rng.integers(0, 6, 4 if paired else 2))]
    if rng.random() < 0.4:
        cli += ["-4", str(pick([30, 60, 100]))]
    if paired and rng.random() < 0.4:
        cfg.append(pick(["trimBadTail=20,30", "trimBadHead=10,25", "trimBadTail=15,20"]))
    if rng.random() < 0.4:
        cfg += [f"adaMis={int(rng.integers(0, 4))},{int(rng.integers(0, 4))}", f"adaMR={pick([0.3, 0.5, 0.7])},{pick([0.3, 0.5, 0.7])}",
                f"adaEdge={int(rng.integers(1, 9))},{int(rng.integers(1, 9))}"]
    if paired and rng.random() < 0.25:
        cfg.append("pe_info")
    rmdup = bool(rng.random() < 0.25)
    if rmdup:
        cfg.append("rmdup")
        k = n // 8
        for m in range(2 if paired else 1):
            d["seq"][m][n // 2:n // 2 + k] = d["seq"][m][0:k]
            if d["len"][m] is not None:                       # (variable lengths: the copy takes the length and the qualities along)
                d["len"][m][n // 2:n // 2 + k] = d["len"][m][0:k]
                d["qual"][m][n // 2:n // 2 + k] = d["qual"][m][0:k]
    if rng.random() < 0.15:
        ct = ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(20, 33)))) for _ in range(2)]
        cfg += ["contam1=" + ct[0]] + (["contam2=" + ct[1]] if paired else []) + ["ctMatchR=0.5"]
        for m in range(2 if paired else 1):
            a = np.frombuffer(ct[m].encode(), dtype=np.uint8)
            for r in rng.choice(n, n // 20, replace=False):
                rl = int(d["len"][m][r]) if d["len"][m] is not None else L
                if rl > len(a) + 5:
                    p0 = int(rng.integers(0, rl - len(a)))
                    d["seq"][m][r, p0:p0 + len(a)] = a
    threads, patch = pick([1, 2, 3, 4, 6]), pick([0, 100, 250, 777])
    # this CLI's own modes
    env, ours_cli = {}, []
    gz_in, gz_out = bool(rng.random() < 0.5), bool(rng.random() < 0.4)
    if rng.random() < 0.4:
        env["SNK_BATCH_PAIRS"] = pick(["1536", "4096"])
    if gz_in and rng.random() < 0.5:
        env["SNK_DEVICE_INFLATE"] = "1"
        env["SNK_DGZ_WINDOW_MB"] = "1"
    if rng.random() < 0.1:
        env["SNK_HOST_TEXT"] = "1"
    mode = rng.random()
    if mode < 0.15:
        ours_cli = ["--devices", "0,0"]
    elif mode < 0.4:                                          # sharded run: two shards (a host wire between them: the device is listed twice),
        ours_cli = ["--devices", "0,0"]                      # plain or .gz input (scout pass), with or without rmdup (hash exchange)
        env["SNK_SHARDED"] = "1"
        env["SNK_SHARD_MIN_RECORDS"] = "700"                 # (the parent shards inputs of 4096 records per shard and more: these are smaller)
        env["SNK_GZ_CHUNK"] = "65536"
        env.setdefault("SNK_BATCH_PAIRS", "1536")
    if rng.random() < 0.1:                                    # (drawn last: the contexts of earlier rounds stay what they were)
        env["SNK_PROVEN_ONLY"] = "1"                          # round 6: every automatic dispatch takes the generic + LDS-histogram kernels
    return dict(paired=paired, L=L, n=n, d=d, cli=cli, cfg=cfg, threads=threads, patch=patch, rmdup=rmdup, env=env, ours_cli=ours_cli, gz_in=gz_in, gz_out=gz_out)


@pytest.mark.parametrize("i", range(N_CONTEXTS))
def test_random_command_lines(i, tmp_path):
    c = _context(i)
    work, d, paired, L = str(tmp_path), c["d"], c["paired"], c["L"]
    mates = 2 if paired else 1
    for m in range(mates):
        path = os.path.join(work, f"r{m + 1}.fq")
        synth.write_fastq(path, d["seq"][m], d["qual"][m], L, m + 1, lens=d["len"][m])
        R._gzip_copy(path)
    lines = list(c["cfg"]) + ([f"patch={c['patch']}"] if c["patch"] else [])
    cfg_args = []
    if lines:
        open(os.path.join(work, "cfg"), "w").write("\n".join(lines) + "\n")
        cfg_args = ["-c", os.path.join(work, "cfg")]

    def command(exe, out, ext_in, ext_out):
        cmd = [exe, "filter", "-1", os.path.join(work, "r1" + ext_in), "-C", "c1" + ext_out, "-o", os.path.join(work, out), "-T", str(c["threads"])]
        if paired:
            cmd += ["-2", os.path.join(work, "r2" + ext_in), "-D", "c2" + ext_out]
        return cmd + cfg_args + c["cli"]

    what = (i, c["cli"], c["cfg"], c["env"], c["ours_cli"], dict(paired=paired, L=L, n=c["n"], T=c["threads"], patch=c["patch"], gz_in=c["gz_in"], gz_out=c["gz_out"]))
    r = subprocess.run(command(T.REF_BIN, "ref", ".fq.gz", ".fq"), capture_output=True)        # (.gz input: the reference's clean FASTQ is complete and in order)
    assert r.returncode == 0, (what, r.stderr[-400:])
    cli = S.build_module().build_cli()
    ext_out = ".fq.gz" if c["gz_out"] else ".fq"
    r = subprocess.run(command(cli, "ours", ".fq.gz" if c["gz_in"] else ".fq", ext_out) + c["ours_cli"], capture_output=True, env=dict(os.environ, **c["env"]))
    assert r.returncode == 0, (what, r.stdout[-300:], r.stderr[-600:])
    ours, ref = os.path.join(work, "ours"), os.path.join(work, "ref")
    if os.path.getsize(os.path.join(ref, "c1.fq")) == 0:
        pytest.skip("no read survives: the reference prints uninitialised bytes into Basic_Statistics_of_Sequencing_Quality.txt")
    for f in (R.REPORT_FILES_PE if paired else R.REPORT_FILES_SE):
        if not paired and f.startswith("Distribution_of_Q20_Q30"):
            # single end: the reference's raw rows end at the `read_length` its thread merge is left with and it prints uninitialised
            # floats behind them (src/seprocess.cpp; this writer prints zeros there): those rows are compared on their clean columns
            a, b = open(os.path.join(ours, f)).read().splitlines(), open(os.path.join(ref, f)).read().splitlines()
            assert len(a) == len(b), (f, what)
            for x, y in zip(a, b):
                xs, ys = x.split("\t"), y.split("\t")
                if len(xs) == 5 and xs[1:3] == ["0.0000", "0.0000"]:
                    xs, ys = xs[:1] + xs[3:], ys[:1] + ys[3:]
                assert xs == ys, (f, x, y, what)
            continue
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), (f, what)
    for k in range(mates):
        assert _cat(os.path.join(ours, f"c{k + 1}" + ext_out)) == _cat(os.path.join(ref, f"c{k + 1}.fq")), (k, what)
    if c["rmdup"]:
        for t in range(c["threads"]):
            for k in range(mates):
                f = f"dupReads.{t}.{k + 1}.gz"
                assert _cat(os.path.join(ours, f)) == _cat(os.path.join(ref, f)), (f, what)
