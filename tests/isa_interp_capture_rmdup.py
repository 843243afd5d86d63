"""Child process of tests/test_simt_isa_coverage.py: the duplicate-marking kernels of the EMULATED library on hashes given directly --
among them the reference's sentinel value 2^64 - 1 (src/rmdup.cpp:100,116), which no read's hash realistically takes and whose
branches the captures of whole CLI runs therefore never enter -- with SIMT_DUMP_DIR set.  Two-pass marking (explicit indices and
positions; sentinel population known / counted) and the one-pass table over three batches; flags compared with a dictionary."""
import ctypes as C
import sys

import numpy as np

import simt_lib as S
from soapnuke_amd import abi

SENT = np.uint64(0xFFFFFFFFFFFFFFFF)


def expect(h, idx=None, sentinel_dup=None):
    idx = np.arange(len(h)) if idx is None else idx
    first = {}
    for x, i in zip(h.tolist(), idx.tolist()):
        first[x] = min(first.get(x, i), i)
    out = np.array([1 if first[x] != i else 0 for x, i in zip(h.tolist(), idx.tolist())], dtype=np.uint8)
    if sentinel_dup is not None:
        out[h == SENT] = sentinel_dup
    return out


def main():
    lib = S.lib()
    lib.simt_dump_register.argtypes = [C.c_void_p, C.c_size_t]
    p = abi.default_params(paired=True, max_read_len=150, rmdup=1)
    ctx = lib.snk_create(C.byref(p), 0)
    assert ctx, lib.snk_last_error()
    rng = np.random.default_rng(41)
    n = 3000
    h = rng.integers(1, 1 << 62, n, dtype=np.uint64)
    h[rng.integers(0, n, 400)] = h[rng.integers(0, n, 400)]
    keep = []

    def reg(a):
        lib.simt_dump_register(a.ctypes.data, a.nbytes)
        keep.append(a)
        return a
    # two passes: positions as indices, no sentinel; explicit (shuffled) indices with the sentinel three times and its population counted on the device
    for variant in range(3):
        hv = reg(h.copy())
        idx = None
        if variant >= 1:
            hv[[5, 1777, 2999]] = SENT
            idx = reg(rng.permutation(n).astype(np.uint32))
        dup = reg(np.zeros(n, dtype=np.uint8))
        total = -1 if variant < 2 else 1                 # variant 2: the caller knows the population of the sentinel's bucket (one: not a duplicate)
        rc = lib.snk_rmdup_mark_device(ctx, hv.ctypes.data, None if idx is None else idx.ctypes.data, n, n, total, dup.ctypes.data, None)
        assert rc == 0, lib.snk_last_error()
        want = expect(hv, idx, None if variant == 0 else (1 if variant == 1 else 0))
        if variant == 1:                                 # population counted: hash % prime of the sentinel's bucket (src/rmdup.cpp:100) -- at least the three
            assert (dup[hv == SENT] == 1).all()
            want[hv == SENT] = 1
        assert np.array_equal(dup, want), (variant, int((dup != want).sum()))
    # one pass: the table lives across three batches, the sentinel in the second
    lib.snk_rmdup_stream_create.restype = C.c_void_p
    t = lib.snk_rmdup_stream_create(ctx, C.c_uint64(n))
    assert t, lib.snk_last_error()
    hs = h.copy()
    hs[1500] = SENT
    got = np.zeros(n, dtype=np.uint8)
    for a, z in ((0, 1000), (1000, 2200), (2200, n)):
        part, d = reg(hs[a:z].copy()), reg(np.zeros(z - a, dtype=np.uint8))
        assert lib.snk_rmdup_stream_mark_device(C.c_void_p(t), part.ctypes.data, C.c_uint64(a), z - a, d.ctypes.data, None) == 0, lib.snk_last_error()
        got[a:z] = d
    seen = C.c_int32(0)
    cnt = C.c_uint64(0)
    lib.snk_rmdup_stream_stats(C.c_void_p(t), C.byref(cnt), C.byref(seen))
    assert seen.value == 1
    ok = hs != SENT
    assert np.array_equal(got[ok], expect(hs)[ok])
    lib.snk_rmdup_stream_destroy(C.c_void_p(t))
    lib.snk_destroy(ctx)
    print("captured")


if __name__ == "__main__":
    main()
