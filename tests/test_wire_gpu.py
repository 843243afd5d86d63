"""The wires between the shards of a sharded run (soapnuke_amd/host/snk_wire.h) where a ONE-GPU box can reach them:

* `snk_wire_selftest` forms a world-1 communicator through RcclWire::make_id / connect -- the dlopen'ed ABI the shards use: ncclUniqueId
  by value, the enum values of ncclUint64 / ncclUint8 / ncclSum / ncclMax / ncclMin, grouped ncclSend + ncclRecv -- and checks
  allreduce_u64 (sum, max, min), alltoallv (with guard bytes behind the receive buffer) and the count exchange against host
  arithmetic, the host wire likewise;
* `SOAPnuke filter` with SNK_SHARDED=1, SNK_SHARD_WIRE=rccl and `--devices 0,0`: two shards ask for ONE communicator on the same
  device, RCCL refuses, the shards learn that from each other over the host wire (the bootstrap), warn, and the host wire carries
  the collectives -- the reference binary's bytes all the same.  With two GPUs the same test takes `0,1` and the RCCL wire itself.

On the CPU (no marker): the emulated build of the self-test runs the host wire's half and says why the RCCL half cannot run."""
import os
import subprocess

import pytest

import report_util as R
import snk_testlib as T
from soapnuke_amd import synth

EXE = os.path.join(T.ROOT, "soapnuke_amd", "snk_wire_selftest")
CLI = os.path.join(T.ROOT, "soapnuke_amd", "SOAPnuke")


@pytest.mark.gpu
@pytest.mark.first_contact
def test_rccl_wire_world1_on_the_device():
    assert os.path.exists(EXE), "soapnuke_amd/snk_wire_selftest is missing: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([EXE, "0"], capture_output=True, text=True, timeout=150)
    if r.returncode == 77:
        pytest.skip("snk_wire_selftest: " + r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-800:])
    assert "host wire, world 1: ok" in r.stdout and "RCCL wire, world 1: ok" in r.stdout and "wire self-test passed" in r.stdout, r.stdout
    assert "RCCL communicator: 1 rank(s), this is rank 0" in r.stdout


@pytest.mark.gpu
@pytest.mark.first_contact
@pytest.mark.skipif(not os.path.exists(T.REF_BIN), reason="oracle/_ref/SOAPnuke not built")
def test_cli_shards_asked_for_rccl(tmp_path):
    import filecmp
    import torch
    two = torch.cuda.device_count() >= 2
    n, L = 20000, 150
    d = synth.make_batch(n, L, paired=True, seed=67)
    for m in range(2):
        d["seq"][m][n // 2 + 100:n // 2 + 600] = d["seq"][m][0:500]          # duplicates across the shard border
    cli = ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2, "-J"]
    case = ("rcclwire", True, L, n, 2, 250, {}, {}, cli, ["rmdup"])
    work = str(tmp_path)
    ref = R.run_reference_cli(case, d, work, gz_input=True)
    ours = os.path.join(work, "ours")
    cmd = [CLI, "filter", "-1", os.path.join(work, "r1.fq"), "-2", os.path.join(work, "r2.fq"), "-C", "c1.fq", "-D", "c2.fq", "-o", ours,
           "-T", "2", "--devices", "0,1" if two else "0,0", "-c", os.path.join(work, "cfg")] + cli
    r = subprocess.run(cmd, capture_output=True, timeout=170, env=dict(os.environ, SNK_SHARDED="1", SNK_SHARD_WIRE="rccl", SNK_SHARD_MIN_RECORDS="1000"))
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-1200:])
    log = open(os.path.join(ours, "log"), "rb").read()
    assert b"shards talk over: RCCL" in log, log[-600:]
    if two:
        assert b"statistics merged over RCCL (2 shards)" in log and b"Warning" not in r.stderr, (log[-600:], r.stderr[-600:])
    else:      # one device twice: no communicator -- agreed on over the bootstrap wire, said out loud, and carried by the host wire
        assert b"the host wire carries the collectives" in r.stderr, r.stderr[-800:]
        assert b"statistics merged over the host wire (2 shards)" in log, log[-600:]
    for f in R.REPORT_FILES_PE:
        assert filecmp.cmp(os.path.join(ours, f), os.path.join(ref, f), shallow=False), f
    for c in ("c1.fq", "c2.fq"):
        assert open(os.path.join(ours, c), "rb").read() == open(os.path.join(ref, c), "rb").read(), c


def test_wire_selftest_on_the_emulated_device():
    import simt_lib as S
    exe = S.build_module().build_wire_selftest()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, (r.returncode, r.stdout, r.stderr[-400:])
    assert "host wire, world 1: ok" in r.stdout and "SKIP: RCCL needs a HIP device" in r.stdout


def test_wire_selftest_without_a_device_says_so():
    import torch
    if torch.cuda.is_available() or not os.path.exists(EXE):
        pytest.skip("needs the gfx950 build of the self-test on a box without a HIP device")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert r.returncode == 77 and "SKIP: no HIP device" in r.stdout
