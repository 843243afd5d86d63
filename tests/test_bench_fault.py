"""bench.py keeps its ONE JSON line when the timed region dies: the headline step runs in a child at N == 1, and a child
that faults, aborts or hangs leaves an "error" record that names the kernel instance.  Here the child dies because this container has no
HIP device (same path as a HIP runtime abort: non-zero exit, no JSON), and once more at its time limit."""
import json
import os
import subprocess
import sys

import snk_testlib as T


def run_bench(*extra, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    e.pop("SNK_BENCH_INPROCESS", None)
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--pairs", "1000", "--no-cpu-baseline",
                        "--no-traffic", *extra], capture_output=True, text=True, timeout=300, env=e)
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    return json.loads(lines[0])


def test_a_dead_headline_child_leaves_an_error_line_naming_the_kernel():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a HIP device (the child is meant to die)")
    out = run_bench("--workload", "c3")
    assert out["value"] is None and out["roofline"] is None and out["n_gpus"] == 1 and out["metric"].startswith("Mreads/s PE150")
    assert "headline child ended with rc" in out["error"] and "HIP device" in out["stderr_tail"]
    assert out["kernel_instance"] == "snk_tiled_kernel<5,true,true,16,TileShape<160,768,4>>"


def test_a_hung_headline_child_ends_at_its_limit():
    out = run_bench(SNK_BENCH_HEADLINE_TIMEOUT_S="0.05")
    assert out["value"] is None and "did not finish within" in out["error"] and out["kernel_instance"].startswith("snk_tiled_kernel<5,false")
