"""bench.py from its first line to its last on the CPU emulator (tests/simt), at a thousandth of its sizes: what this checks is
that the ONE JSON line the round-end driver parses is produced with every key of the contract -- the headline block with
`roofline`, `cpu_baseline`, the `end_to_end` legs (this CLI and the reference binary on the same files, `report_identical`),
the `pe250_rmdup` leg, every `other_workloads` row without an error -- by the code paths written while the GPU was closed.
The numbers mean nothing here."""
import importlib.util
import json
import os
import sys

import pytest

import simt_lib as S
import snk_testlib as T

CORE = ["test_bench_line", "test_bounded_memory", "test_smoke_entry_point", "test_bench_with_two_ranks"]
pytestmark = pytest.mark.skipif(not os.path.exists(T.REF_BIN), reason="oracle/_ref/SOAPnuke not built")


def test_bench_line_has_every_leg(monkeypatch, capsys):
    S.torch_on_host(monkeypatch)
    monkeypatch.setenv("SNK_BENCH_TEST_DIVISOR", "1000")
    monkeypatch.setenv("SNK_BENCH_INPROCESS", "1")       # (on the hardware the other_workloads rows run in a child process: a fault there leaves the headline)
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(T.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys.path.insert(0, os.path.join(T.ROOT, "tools"))
    import bench_e2e
    monkeypatch.setattr(bench_e2e, "OURS", S.build_module().build_cli())
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1", "--pairs", "10000", "--e2e-pairs", "16000"])
    bench.main()
    line = [x for x in capsys.readouterr().out.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "end_to_end", "other_workloads"):
        assert k in out, k
    assert out["steps"] == 2 and out["warmup"] == 1 and out["n_gpus"] == 1 and out["config"]["pairs_per_gpu_per_step"] == 10000
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and out["roofline"]["bound"] == "hbm"
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and out["cpu_baseline"]["kind"] == "reference"
    e2e = out["end_to_end"]
    assert "error" not in e2e, e2e
    assert set(e2e["modes"]) == {"plain_ours", "gz", "gz2plain", "gz_c3"}
    for m in ("gz", "gz2plain", "gz_c3"):
        leg = e2e["modes"][m]
        assert leg["ours"]["rc"] == 0 and leg["reference"]["rc"] == 0 and leg["report_identical"] is True, (m, leg)
    assert e2e["modes"]["plain_ours"]["ours"]["rc"] == 0 and "speedup_vs_reference_gz2plain" not in e2e["modes"]["plain_ours"]
    assert set(out["e2e_value"]) >= {"gz", "gz2plain", "plain_ours", "gz_c3", "unit"} and out["e2e_value"]["gz"] == e2e["modes"]["gz"]["ours"]["Mreads_per_s"]
    rm = e2e["pe250_rmdup"]
    assert "error" not in rm and rm["modes"]["gz"]["report_identical"] is True, rm
    rows = out["other_workloads"]
    assert len(rows) == 11 and all(isinstance(r.get("error"), int) and r["error"] == 0 for r in rows), [r for r in rows if r.get("error") != 0]


def test_bounded_memory_big_run_tool(monkeypatch, capsys):
    """tools/bench_e2e_big.py --bounded (the 628 M-pair run of BASELINE configs[2] under the GPU box's 300 GiB memory cgroup: this CLI
    timed with holes punched behind its writer, once more into named pipes read by md5 consumers, the reference with a tail reader
    behind its appends) at 30 000 pairs with the emulated CLI: every stage runs, reports and clean FASTQ compare equal."""
    sys.path.insert(0, os.path.join(T.ROOT, "tools"))
    import bench_e2e
    import bench_e2e_big
    monkeypatch.setattr(bench_e2e, "OURS", S.build_module().build_cli())
    monkeypatch.setenv("SNK_BIG_UNIT", "6000")
    monkeypatch.setattr(sys, "argv", ["bench_e2e_big.py", "30000", "4", "--bounded"])
    bench_e2e_big.main()
    out = json.loads([x for x in capsys.readouterr().out.splitlines() if x.startswith("{")][-1])
    gz = out["gz"]
    assert gz["ours"]["rc"] == 0 and gz["ours_verify"]["rc"] == 0 and gz["reference"]["rc"] == 0
    assert gz["ours_verify"]["reports_same_as_timed_run"] is True
    assert gz["reports_compared"] == 10 and gz["reports_differing"] == [] and gz["clean_fastq_identical"] is True
    assert out["watchdog_tripped"] is False and "bounded" in out["mode"]


def test_smoke_entry_point(monkeypatch, capsys):
    """__graft_entry__.smoke() -- what the driver runs on the GPU box before the bench -- with the emulated library"""
    S.torch_on_host(monkeypatch)
    spec = importlib.util.spec_from_file_location("graft_entry_under_test", os.path.join(T.ROOT, "__graft_entry__.py"))
    entry = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(entry)
    entry.smoke()
    assert "smoke OK" in capsys.readouterr().out


def test_bench_with_two_ranks(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* in the environment), on the
    emulated library with gloo in RCCL's place: the stats all-reduce inside the timed region, the MAX of the elapsed time, rank 0's line"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SIMT_THREADS="3",
                   PYTHONPATH=os.pathsep.join([here, T.ROOT, os.environ.get("PYTHONPATH", "")]))
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "simt_bench_rank.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "6000"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    lines = [x for x in outs[0][0].splitlines() if x.startswith("{")]
    assert len(lines) == 1 and not [x for x in outs[1][0].splitlines() if x.startswith("{")]        # one line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "weak" and out["config"]["parallelism"] == "shard2"
    assert out["config"]["pairs_per_gpu_per_step"] == 6000 and out["steps"] == 3
    # whole-job throughput: both ranks' reads over the slower rank's time
    assert abs(out["value"] - 2.0 * 6000 * 2 * 3 / (out["ms_per_step"] * 3 / 1e3) / 1e6) < 0.02 * out["value"] + 1e-3
    assert out["config"]["clean_pairs_per_step_per_gpu"] > 0 and "cpu_baseline" not in out
