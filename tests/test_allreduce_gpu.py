"""The stats collective on a real GPU, world_size 1: (a) through torch.distributed/RCCL as bench.py issues
it, (b) through the C ABI snk_stats_allreduce() with an ncclComm_t created straight from librccl."""
import ctypes as C
import os

import numpy as np
import pytest

import snk_testlib as T
from cases import PE_CASES
from soapnuke_amd import abi, synth

pytestmark = pytest.mark.gpu


def _run(ctx, d):
    dev = ctx.upload(d)
    rec = ctx.alloc_records(d["n"])
    ctx.filter_batch(ctx.make_batch(dev), rec)
    return rec


def test_torch_rccl_allreduce_world1():
    import torch
    import torch.distributed as dist
    from soapnuke_amd.filter import FilterContext
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 500))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        d = synth.make_batch(20000, 150, paired=True, seed=81)
        p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C3_full"])
        ctx = FilterContext(p, device=0)
        _run(ctx, d)
        ctx.allreduce()
        s, mx, err = ctx.fetch()
        o = T.run_oracle(p, d)
        assert err[0] == 0 and np.array_equal(s, o["sum"]) and np.array_equal(mx, o["max"])
    finally:
        dist.destroy_process_group()


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_c_abi_allreduce_world1():
    import torch
    from soapnuke_amd.filter import FilterContext
    rccl = None
    for name in ("librccl.so.1", "librccl.so"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("librccl not loadable by name")
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        d = synth.make_batch(20000, 150, paired=True, seed=82)
        p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C2_adatrim_lowq"])
        ctx = FilterContext(p, device=0)
        _run(ctx, d)
        ctx.lib.snk_stats_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        rc = ctx.lib.snk_stats_allreduce(ctx.ctx, comm, ctx._stream())
        assert rc == 0, ctx.lib.snk_last_error()
        torch.cuda.synchronize()
        s, mx, err = ctx.fetch()
        o = T.run_oracle(p, d)
        assert err[0] == 0 and np.array_equal(s, o["sum"]) and np.array_equal(mx, o["max"])
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
