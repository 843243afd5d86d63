"""The bit-sliced adapter search of the kernels (soapnuke_amd/csrc/snk_adapter_bits.hip.h) compiled for the HOST as one lane of a
wavefront (tests/host_emul/: a stand-in <hip/hip_runtime.h>) and fuzzed against the oracle's adapter_pos() -- no GPU needed.
Covers what round 4 added to the fast paths: adapters of 1..255 characters (the screen sees the first 64; survivors of longer
ones are decided character by character), adapters shorter than 6, adaEdge beyond the adapter's length, budgets of any size."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import snk_testlib as T

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
LIB = os.path.join(HERE, "libsnk_emul.so")


@pytest.fixture(scope="module")
def emul():
    srcs = [os.path.join(HERE, "adapter_emul.cpp"), os.path.join(HERE, "hip", "hip_runtime.h")] + [
        os.path.join(T.ROOT, "soapnuke_amd", "csrc", f) for f in ("snk_adapter_bits.hip.h", "snk_common.hip.h", "snk_tables.h", "snk_device.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-fPIC", "-shared", "-I.", "-I" + os.path.join(T.ROOT, "soapnuke_amd", "csrc"),
                               "-x", "c++", "adapter_emul.cpp", "-o", LIB], cwd=HERE)
    lib = C.CDLL(LIB)
    lib.snk_emul_adapter_pos.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int)]
    return lib


def _case(rng, long_adapters):
    al = int(rng.integers(65, 256)) if long_adapters else int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12, 20, 32, 42, 63, 64]))
    alphabet = b"ACGT" if rng.random() < 0.8 else b"ACGTN"
    ada = bytes(rng.choice(list(alphabet), al).astype(np.uint8))
    if rng.random() < 0.1:
        ada = ada[:al // 2] + ada[al // 2:].lower()
    L = int(rng.integers(max(al, 20) if rng.random() < 0.85 else 8, 257))
    read = bytearray(rng.choice(list(b"ACGT"), L).astype(np.uint8).tobytes())
    kind = rng.integers(0, 6)
    a = bytearray(ada)
    for _ in range(int(rng.integers(0, 5))):                  # a few substitutions in the planted copy
        a[int(rng.integers(0, al))] = int(rng.choice(list(b"ACGT")))
    if kind == 0 and L >= al:                                 # whole adapter somewhere
        p = int(rng.integers(0, L - al + 1))
        read[p:p + al] = a
    elif kind == 1:                                           # adapter running off the read end (phase C)
        keep = int(rng.integers(1, min(al, L) + 1))
        read[L - keep:] = a[:keep]
    elif kind == 2:                                           # adapter's head missing (phase A)
        r1 = int(rng.integers(1, 7))
        tail = a[r1:][:L]
        read[:len(tail)] = tail
    elif kind == 3 and L >= al:                               # two copies
        for p in (int(rng.integers(0, L - al + 1)), int(rng.integers(0, L - al + 1))):
            read[p:p + al] = a
    if rng.random() < 0.1:
        for _ in range(3):
            read[int(rng.integers(0, L))] = int(rng.choice(list(b"Nacgtn")))
    mis = int(rng.integers(0, 6))
    mr = float(rng.choice([0.2, 0.3, 0.5, 0.7, 1.0]))
    edge = int(rng.integers(1, 13)) if rng.random() < 0.8 else int(rng.integers(al, al + 5))
    return bytes(read), ada, mis, mr, edge


@pytest.mark.parametrize("long_adapters", [False, True])
def test_bit_sliced_search_matches_the_oracle(emul, long_adapters):
    olib = T.oracle_lib()
    rng = np.random.default_rng(4100 + int(long_adapters))
    n, hits, ok_flag = 25000, 0, C.c_int()
    for _ in range(n):
        read, ada, mis, mr, edge = _case(rng, long_adapters)
        want = olib.snk_oracle_adapter_pos(read, len(read), ada, len(ada), mis, mr, edge)
        got = emul.snk_emul_adapter_pos(read, len(read), ada, mis, mr, edge, C.byref(ok_flag))
        assert ok_flag.value == 1, (ada, mis, mr, edge)
        assert got == want, (read, ada, mis, mr, edge, got, want)
        hits += want >= 0
    assert n // 10 < hits < n                                  # (the generator plants findable and unfindable copies)


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref/libsnkref.so not built")
def test_bit_sliced_search_matches_the_compiled_reference(emul):
    """the same against adapter_pos() of the compiled reference itself, reads at least as long as the adapter"""
    ref = T.ref_lib()
    rng = np.random.default_rng(4200)
    ok_flag = C.c_int()
    for _ in range(8000):
        read, ada, mis, mr, edge = _case(rng, bool(rng.integers(0, 2)))
        if len(read) < len(ada) or edge > len(ada) or ada != ada.upper() or read != read.upper():
            continue
        want = ref.snkref_adapter_pos(read, len(read), ada, len(ada), mis, mr, edge)
        got = emul.snk_emul_adapter_pos(read, len(read), ada, mis, mr, edge, C.byref(ok_flag))
        assert got == want, (read, ada, mis, mr, edge, got, want)
