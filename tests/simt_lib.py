"""Helpers of the SIMT tier (tests/test_simt_*.py): the HIP sources of soapnuke_amd/csrc built for the host with the emulator of
tests/simt (kernels run on fibers with wavefront semantics, see tests/simt/hip/hip_runtime.h), called through the same C ABI as the
gfx950 library.  With the emulator device memory IS host memory: numpy arrays are passed where the ABI wants device pointers."""
import ctypes as C
import importlib.util
import os

import numpy as np

import snk_testlib as T
from soapnuke_amd import abi

SIMT = os.path.join(T.ROOT, "tests", "simt")
_lib = None


def build_module():
    spec = importlib.util.spec_from_file_location("build_simt", os.path.join(SIMT, "build_simt.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def lib():
    """the emulated twin of libsnk_filter.so (built on first use, ~90 s)"""
    global _lib
    if _lib is None:
        _lib = abi.load_library(build_module().build())
    return _lib


def run_device(p, d, kernel=0, first_index=0, dup=None, chunks=1):
    """snk_filter_batch_device() over numpy "device" memory"""
    L = lib()
    ctx = L.snk_create(C.byref(p), 0)
    assert ctx, L.snk_last_error()
    try:
        n = d["n"]
        mates = len(d["seq"])
        rec = [np.zeros(n, dtype=abi.record_dtype()) for _ in range(2)]
        edges = np.linspace(0, n, chunks + 1).astype(int)
        for a, z in zip(edges[:-1], edges[1:]):
            if z == a:
                continue
            sub = {"n": int(z - a), "L": d["L"], "pitch": d["pitch"], "seq": [x[a:z] for x in d["seq"]], "qual": [x[a:z] for x in d["qual"]],
                   "len": [None if x is None else x[a:z] for x in d["len"]]}
            b = T.host_batch(sub, first_index + int(a), None if dup is None else dup[a:z])
            r = [rec[0][a:z], rec[1][a:z]]
            rc = L.snk_filter_batch_device(ctx, C.byref(b), r[0].ctypes.data, r[1].ctypes.data if mates == 2 else None, None, kernel)
            assert rc == 0, L.snk_last_error()
        s, mx = T.new_stats(p)
        err = abi.Error()
        assert L.snk_stats_fetch(ctx, s.ctypes.data, mx.ctypes.data, C.byref(err), None) == 0, L.snk_last_error()
        return dict(rec=rec, sum=s, max=mx, err=(err.code, err.mate, err.index))
    finally:
        L.snk_destroy(ctx)


def torch_on_host(monkeypatch):
    """For test functions written against the GPU tier: with the emulator device memory is host memory, so "cuda" tensors
    are CPU tensors, streams are the one host timeline, and abi.load_library() hands out the emulated library.  Only the
    torch entry points the tests and soapnuke_amd/filter.py use are covered."""
    import contextlib
    import types

    import torch

    L = lib()
    monkeypatch.setattr(abi, "load_library", lambda path=None: L)
    stream = types.SimpleNamespace(cuda_stream=0, synchronize=lambda: None, wait_stream=lambda s: None)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: stream)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: stream)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    class Event:                                        # wall-clock stand-in (timing means nothing on the emulator)
        def __init__(self, enable_timing=False):
            self.t = 0.0

        def record(self, stream=None):
            import time
            self.t = time.perf_counter()

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return max((other.t - self.t) * 1e3, 1e-6)

    monkeypatch.setattr(torch.cuda, "Event", Event)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (16 << 30, 32 << 30))

    def is_cuda(d):
        return d is not None and (str(d).startswith("cuda") or (isinstance(d, int) and not isinstance(d, bool)))

    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    real_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, (torch.device, str)) and is_cuda(x)) else x for x in a)
        if is_cuda(k.get("device")):
            k["device"] = "cpu"
        return real_to(self, *a, **k)

    monkeypatch.setattr(torch.Tensor, "to", to)
    for name in ("zeros", "empty", "ones", "full", "arange", "tensor", "zeros_like", "empty_like", "randint"):
        real = getattr(torch, name)

        def factory(*a, _real=real, **k):
            if is_cuda(k.get("device")):
                k["device"] = "cpu"
            return _real(*a, **k)

        monkeypatch.setattr(torch, name, factory)
    return L
