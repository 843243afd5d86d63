"""The end-to-end tests of the GPU tier (tests/test_cli_gpu.py: this repo's `SOAPnuke filter` against the compiled reference binary on
the same FASTQ files -- reports, clean FASTQ and side files byte-identical) run on the CPU: the CLI of soapnuke_amd/host built
against the SIMT-emulated library (tests/simt).  Readers, parallel gunzip, device-side FASTQ parse / format / gzip, the filter
kernels, duplicate marking, shards, report writers: the same code as on the GPU box, every kernel executed by the emulator."""
import os

import pytest

import simt_lib as S
import snk_testlib as T
import test_cli_gpu as CG
import test_gunzip_gpu as GZ

# what an ordinary run takes (substrings of the test ids); SNK_SIMT_FULL=1: everything (tests/conftest.py)
CORE = ["test_cli_matches_reference_binary[se_trim_T2]", "test_cli_gz_in_gz_out", "test_cli_rmdup_one_pass_variants[small_batches]",
        "test_cli_sharded_ingest_emulated[True-True-True]", "test_cli_sharded_rmdup_and_wire_emulated[True-40000-True]",
        "test_cli_proven_only_launches_no_rewritten_kernel", "test_cli_sharded_gz_index",
        "test_cli_more_than_two_shards[4-True-False]"]


@pytest.fixture(autouse=True)
def _emulated_cli(monkeypatch):
    cli = S.build_module().build_cli()
    monkeypatch.setattr(CG, "CLI", cli)
    monkeypatch.setattr(GZ, "CLI", cli, raising=False)


from test_cli_gpu import *      # noqa: E402,F401,F403  (every test function of the GPU tier's module)
from test_gunzip_gpu import test_cli_with_device_inflate_matches_the_reference_binary      # noqa: E402,F401

# BEHIND the star import: it brings test_cli_gpu's own `pytestmark` (gpu) along, which would take this whole module out of the CPU
# suite and into `pytest -m gpu` on the GPU box (it did, for part of round 5)
pytestmark = pytest.mark.skipif(not os.path.exists(T.REF_BIN), reason="oracle/_ref/SOAPnuke not built")


# ---- written while the GPU was closed (guarded in the GPU tier until they have run on hardware): this tier is where they run first
@pytest.mark.parametrize("paired,gz_out,trim", [(True, False, False), (True, True, True), (False, False, True)])
def test_cli_sharded_ingest_emulated(paired, gz_out, trim, tmp_path):
    CG.test_cli_sharded_ingest(paired, gz_out, trim, tmp_path)


@pytest.mark.parametrize("paired,n,gz_in", [(True, 40000, False), (False, 40100, False), (True, 40000, True), (False, 40100, True)])
def test_cli_sharded_rmdup_and_wire_emulated(paired, n, gz_in, two_emulated_devices, monkeypatch, tmp_path):
    """two shards on two emulated devices; the emulator has no RCCL: the host wire carries the collectives"""
    monkeypatch.setenv("SNK_SHARD_WIRE", "host")
    CG.test_cli_sharded_rmdup_and_wire(paired, n, gz_in, tmp_path)


def test_cli_shards_fall_back_to_the_host_wire_together_emulated(two_emulated_devices, tmp_path):
    """two different (emulated) devices: the parent asks for the RCCL wire; no shard can join a communicator here (no GPU: rank 0 gets
    no id from ncclGetUniqueId) -- they learn that from each other over the host wire, which then carries the collectives: a warning,
    not a hang, and the reference's bytes"""
    import filecmp
    import subprocess
    n, L = 20000, 150
    d = CG.synth.make_batch(n, L, paired=True, seed=66)
    for m in range(2):
        d["seq"][m][n // 2 + 100:n // 2 + 600] = d["seq"][m][0:500]
    cli = ["-f", CG.synth.ADAPTER1, "-r", CG.synth.ADAPTER2, "-J"]
    case = ("fallback", True, L, n, 2, 250, {}, {}, cli, ["rmdup"])
    work = str(tmp_path)
    ref = CG.R.run_reference_cli(case, d, work, gz_input=True)
    cmd = [CG.CLI, "filter", "-1", os.path.join(work, "r1.fq"), "-2", os.path.join(work, "r2.fq"), "-C", "c1.fq", "-D", "c2.fq", "-o", os.path.join(work, "ours"),
           "-T", "2", "--devices", "0,1", "-c", os.path.join(work, "cfg")] + cli
    env = dict(os.environ, SNK_SHARDED="1", SNK_SHARD_MIN_RECORDS="1000")
    env.pop("SNK_SHARD_WIRE", None)
    r = subprocess.run(cmd, capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-800:])
    assert b"the host wire carries the collectives" in r.stderr
    log = open(os.path.join(work, "ours", "log"), "rb").read()
    assert b"shards talk over: RCCL" in log and b"statistics merged over the host wire (2 shards)" in log
    CG._compare_dirs(os.path.join(work, "ours"), ref, True)
    for f in ("dupReads.0.1.gz", "dupReads.1.2.gz"):
        assert CG._cat(os.path.join(work, "ours", f)) == CG._cat(os.path.join(ref, f)), f


def test_cli_rmdup_table_that_does_not_fit_emulated(tmp_path):
    CG.test_cli_rmdup_one_pass_variants("table_does_not_fit", tmp_path)


@pytest.mark.parametrize("n,batch,mode", [(20000, "4096", "one"), (20100, "4096", "one"), (20100, "700", "one"), (19999, "4096", "gz"), (20100, "4096", "two_pass"),
                                           (20100, "4096", "sentinel_restart"), (20100, "2048", "two_devices")])
def test_cli_rmdup_single_end_one_pass_emulated(n, batch, mode, tmp_path):
    CG.test_cli_rmdup_single_end_one_pass(n, batch, mode, tmp_path)


@pytest.fixture
def two_emulated_devices(monkeypatch):
    """--devices 0,1 with the emulator posing as two devices (one address space, so the copies between them are plain copies:
    what this shows is the orchestration -- which buffers, which stream, which order -- not peer access)"""
    import torch
    monkeypatch.setenv("SIMT_DEVICES", "2")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)


@pytest.mark.parametrize("paired", [True, False])
def test_cli_rmdup_one_pass_across_two_devices_emulated(paired, two_emulated_devices, tmp_path):
    """the duplicate table on the first device, the second device's batches marked there (hashes over, flags back)"""
    if paired:
        CG.test_cli_two_device_slots(True, tmp_path)
        assert b"rmdup: one pass" in open(os.path.join(str(tmp_path), "ours", "log"), "rb").read()
    else:
        CG.test_cli_rmdup_single_end_one_pass(20100, "2048", "two_devices", tmp_path)


def test_cli_proven_only_launches_no_rewritten_kernel(tmp_path):
    """SNK_PROVEN_ONLY=1 on the emulated CLI with every launch recorded (SIMT_DUMP_DIR): the filter step is the generic kernel + the
    LDS histogram kernel and nothing of the tiled / long-read / contaminant families; without the switch the tiled kernel runs"""
    import json
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(T.ROOT, "tools"))
    import gfx950_interp as G
    n, L = 600, 150
    d = CG.synth.make_batch(n, L, paired=True, seed=91)
    for m in range(2):
        CG.synth.write_fastq(os.path.join(str(tmp_path), "r%d.fq" % (m + 1)), d["seq"][m], d["qual"][m], L, m + 1)
    seen = {}
    for tag, env in (("proven", {"SNK_PROVEN_ONLY": "1"}), ("default", {})):
        dump = os.path.join(str(tmp_path), "dump_" + tag)
        os.makedirs(dump)
        cmd = [CG.CLI, "filter", "-1", os.path.join(str(tmp_path), "r1.fq"), "-2", os.path.join(str(tmp_path), "r2.fq"), "-C", "c1.fq", "-D", "c2.fq",
               "-o", os.path.join(str(tmp_path), tag), "-T", "1", "-f", CG.synth.ADAPTER1, "-r", CG.synth.ADAPTER2, "-J", "-l", "10", "-q", "0.1"]
        r = subprocess.run(cmd, capture_output=True, timeout=600,
                           env=dict(os.environ, SIMT_DUMP_DIR=dump, SIMT_DUMP_PER_KERNEL="1", SIMT_DUMP_MAX_ALLOC="1", SIMT_CUS="2", **env))
        assert r.returncode == 0, r.stderr[-800:]
        metas = [json.load(open(os.path.join(dump, f))) for f in os.listdir(dump) if f.endswith(".json")]
        seen[tag] = {G.symbol_at(m["lib"], m["offset"]) for m in metas}
    rewritten = ("snk_tiled_kernel", "snk_long_decide", "snk_long_prep", "snk_contam_kernel")
    assert any("snk_generic_kernel" in k for k in seen["proven"]) and any("snk_long_hist_kernel" in k for k in seen["proven"]), seen["proven"]
    assert not [k for k in seen["proven"] if any(x in k for x in rewritten)], seen["proven"]
    assert any("snk_tiled_kernel" in k for k in seen["default"]) and not any("snk_generic_kernel" in k for k in seen["default"])
    for f in ("c1.fq", "c2.fq", "Basic_Statistics_of_Sequencing_Quality.txt", "Statistics_of_Filtered_Reads.txt"):
        assert open(os.path.join(str(tmp_path), "proven", f), "rb").read() == open(os.path.join(str(tmp_path), "default", f), "rb").read(), f
