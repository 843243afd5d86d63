"""AddressSanitizer over the kernels and the host side at once: the emulated library (tests/simt: the HIP sources compiled for the
host) and the CLI of soapnuke_amd/host built with -fsanitize=address, end-to-end cases of tests/test_cli_gpu.py run with them
against the reference binary.  Device memory is the sanitizer's malloc here, so a kernel that reads or writes past a device buffer
(or past the 160 KB of LDS), a host buffer overrun in the readers / writers, a use after free of a slot -- abort the run.
(VERDICT r3 #9 asked for sanitizer builds of the CLI; the emulator's fiber switches are announced to the sanitizer,
tests/simt/simt_runtime.cpp.)"""
import os

import pytest

import simt_lib as S
import snk_testlib as T
import test_cli_gpu as CG
import test_gunzip_gpu as GZ

# the sanitizer builds take minutes to make: only SNK_SIMT_FULL=1 runs these (tests/conftest.py; profiles/r04_simt_full.txt holds such a run)
CORE = []
pytestmark = pytest.mark.skipif(not os.path.exists(T.REF_BIN), reason="oracle/_ref/SOAPnuke not built")
ASAN_FLAGS = ("-fsanitize=address", "-fno-sanitize-recover=all", "-shared-libasan")


@pytest.fixture
def tsan_cli(monkeypatch):
    """ThreadSanitizer over the CLI's own threads (the emulated library stays uninstrumented: its fibers are not threads)"""
    cli = S.build_module().build_cli_tsan()
    monkeypatch.setattr(CG, "CLI", cli)
    monkeypatch.setenv("TSAN_OPTIONS", "halt_on_error=1 second_deadlock_stack=1 exitcode=66")


@pytest.fixture(autouse=True)
def _asan_cli(monkeypatch):
    import subprocess
    mod = S.build_module()
    cli = mod.build_cli(extra=ASAN_FLAGS, tag="_asan")
    rt = os.path.dirname(subprocess.check_output([mod.CXX, "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip())
    monkeypatch.setattr(CG, "CLI", cli)
    monkeypatch.setattr(GZ, "CLI", cli, raising=False)
    monkeypatch.setenv("LD_LIBRARY_PATH", rt + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    monkeypatch.setenv("ASAN_OPTIONS", "detect_leaks=0:abort_on_error=0:detect_stack_use_after_return=0")


@pytest.mark.parametrize("case", CG.R.REPORT_CASES, ids=[c[0] for c in CG.R.REPORT_CASES])
def test_asan_cli_matches_reference_binary(case, tmp_path):
    CG.test_cli_matches_reference_binary(case, tmp_path)


def test_asan_gz_in_gz_out(tmp_path):
    CG.test_cli_gz_in_gz_out(tmp_path)


def test_asan_long_reads(tmp_path):
    CG.test_cli_long_reads(600, True, tmp_path)


def test_asan_contaminants(tmp_path):
    CG.test_cli_contaminants_match_reference_binary(True, 150, 12000, tmp_path)


@pytest.mark.parametrize("mode", ["small_batches", "two_pass"])
def test_asan_rmdup(mode, tmp_path):
    CG.test_cli_rmdup_one_pass_variants(mode, tmp_path)


def test_asan_rmdup_single_end(tmp_path):
    CG.test_cli_rmdup_single_end_one_pass(20100, "700", "one", tmp_path)


def test_asan_device_inflate(tmp_path):
    GZ.test_cli_with_device_inflate_matches_the_reference_binary(tmp_path)


def test_asan_sharded_ingest(tmp_path):
    CG.test_cli_sharded_ingest(True, True, True, tmp_path)


@pytest.mark.parametrize("paired,n,gz_in", [(True, 40000, True), (False, 40100, False)])
def test_asan_sharded_rmdup_and_wire(paired, n, gz_in, monkeypatch, tmp_path):
    """round 5: the shards' wire (host wire here), the statistics all-reduce, the hash exchange with its partition / flags-home
    kernels, the scout pass and the mid-stream start of the decoders"""
    monkeypatch.setenv("SIMT_DEVICES", "2")
    monkeypatch.setenv("SNK_SHARD_WIRE", "host")
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    CG.test_cli_sharded_rmdup_and_wire(paired, n, gz_in, tmp_path)


def test_asan_streaming(tmp_path):
    CG.test_cli_streaming(True, 2, 50, ["pe_info", "outQualSys=1"], tmp_path)


@pytest.mark.parametrize("which", ["pe_full", "gz", "rmdup_small_batches", "streaming", "sharded", "sharded_rmdup_gz", "four_shards_rmdup"])
def test_tsan_cli_host_threads(which, tsan_cli, tmp_path):
    if which == "pe_full":
        CG.test_cli_matches_reference_binary(CG.R.REPORT_CASES[1], tmp_path)
    elif which == "gz":
        CG.test_cli_gz_in_gz_out(tmp_path)
    elif which == "rmdup_small_batches":
        CG.test_cli_rmdup_one_pass_variants("small_batches", tmp_path)
    elif which == "streaming":
        CG.test_cli_streaming(True, 2, 50, ["pe_info", "outQualSys=1"], tmp_path)
    elif which == "four_shards_rmdup":         # round 6: the host wire's exchange piece by piece, a peer's sender thread next to its receiving main thread
        CG.test_cli_more_than_two_shards(4, True, False, tmp_path)
    elif which == "sharded_rmdup_gz":          # the scout's two decoder threads, the shards' readers started mid-stream, the wire
        import torch
        os.environ["SIMT_DEVICES"] = "2"
        os.environ["SNK_SHARD_WIRE"] = "host"
        try:
            real = torch.cuda.device_count
            torch.cuda.device_count = lambda: 2
            CG.test_cli_sharded_rmdup_and_wire(True, 40000, True, tmp_path)
        finally:
            torch.cuda.device_count = real
            os.environ.pop("SIMT_DEVICES", None)
            os.environ.pop("SNK_SHARD_WIRE", None)
    else:
        CG.test_cli_sharded_ingest(True, False, False, tmp_path)
