"""One rank of `bench.py --gpus N` on the CPU (child process of tests/test_simt_bench.py::test_bench_with_two_ranks): the emulated
library instead of the gfx950 one, "cuda" tensors in host memory (simt_lib.torch_on_host) and torch.distributed over gloo where the
GPU box uses RCCL -- so that the multi-rank path of bench.py (rank-dependent data, the stats all-reduce inside the timed region,
the barrier, the MAX over ranks of the elapsed time, one JSON line from rank 0), which no round could run on more than one GPU,
executes once."""
import importlib.util
import os
import sys

import pytest

import simt_lib as S
import snk_testlib as T


def main():
    mp = pytest.MonkeyPatch()
    S.torch_on_host(mp)
    import torch
    import torch.distributed as dist
    mp.setattr(torch.cuda, "device_count", lambda: 8)
    real_init = dist.init_process_group

    def init(backend=None, **kw):
        kw.pop("device_id", None)
        return real_init("gloo", **kw)

    mp.setattr(dist, "init_process_group", init)
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(T.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys.argv = ["bench.py"] + sys.argv[1:]
    bench.main()


if __name__ == "__main__":
    main()
