"""The parent side of the CLI's sharded ingest / egress (soapnuke_amd/host/snk_main.cpp, SNK_SHARDED=1) without a GPU: with the
SNK_SHARD_FAKE test hook every shard copies its record range to its part files and reports empty statistics, so what runs is the
parent's own work -- newline counting, record-aligned byte ranges that agree between the mates, the children's environment, the
concatenation in rank order, the merge of the statistics blocks, the clean-up."""
import os
import subprocess

import numpy as np
import pytest

import snk_testlib as T
from soapnuke_amd import synth

CLI = os.path.join(T.ROOT, "soapnuke_amd", "SOAPnuke")
pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="soapnuke_amd/SOAPnuke not built")


@pytest.mark.parametrize("paired,shards,last_newline", [(True, 2, True), (True, 3, False), (False, 4, True)])
def test_record_ranges_concatenation_and_merge(paired, shards, last_newline, tmp_path):
    n, L = 30011, 100                                          # (not a multiple of the shard count)
    d = synth.make_batch(n, L, paired=paired, seed=71, var_len=True)
    files = []
    for m in range(2 if paired else 1):
        p = str(tmp_path / f"r{m + 1}.fq")
        synth.write_fastq(p, d["seq"][m], d["qual"][m], L, m + 1, lens=d["len"][m])     # reads of different lengths: records of different sizes
        if not last_newline:
            raw = open(p, "rb").read()
            open(p, "wb").write(raw[:-1])
        files.append(p)
    out = str(tmp_path / "out")
    cmd = [CLI, "filter", "-1", files[0], "-C", "c1.fq", "-o", out, "-T", "3", "--devices", ",".join(["0"] * shards)]
    if paired:
        cmd += ["-2", files[1], "-D", "c2.fq"]
    r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, SNK_SHARDED="1", SNK_SHARD_FAKE="1"))
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-600:])
    for m, f in enumerate(files):
        assert open(os.path.join(out, f"c{m + 1}.fq"), "rb").read() == open(f, "rb").read()      # the ranges tile the file, in order
    left = sorted(os.listdir(out))
    assert not [x for x in left if ".part" in x or x.startswith("shard.") or ".shard" in x], left
    log = open(os.path.join(out, "log")).read()
    assert f"sharded run: {shards} shards" in log and "Analysis accomplished" in log
    # the fake shards report (global index of their first record + 1) raw reads: the merged report shows the sum
    firsts = [n * g // shards for g in range(shards)]
    rep = open(os.path.join(out, "Basic_Statistics_of_Sequencing_Quality.txt")).read()
    assert str(sum(f + 1 for f in firsts)) in rep, rep[:400]


def test_mates_of_different_length_are_refused(tmp_path):
    d = synth.make_batch(20000, 100, paired=True, seed=72)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    synth.write_fastq(f1, d["seq"][0], d["qual"][0], 100, 1)
    synth.write_fastq(f2, d["seq"][1][:19999], d["qual"][1][:19999], 100, 2)
    r = subprocess.run([CLI, "filter", "-1", f1, "-2", f2, "-C", "c1.fq", "-D", "c2.fq", "-o", str(tmp_path / "o"), "--devices", "0,0"],
                       capture_output=True, env=dict(os.environ, SNK_SHARDED="1", SNK_SHARD_FAKE="1"))
    assert r.returncode == 1 and b"reads number in fq1 and fq2 are different" in r.stderr
