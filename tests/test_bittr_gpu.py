"""The lane = position -> lane = read hand-over primitive of the tiled kernel on its own:
64 x 64 bit-matrix transposes across a wave (csrc/snk_bittr.hip.h) against numpy, through the
shipped library (include/snk_selftest.h)."""
import ctypes as C

import numpy as np
import pytest

from soapnuke_amd import abi

pytestmark = pytest.mark.gpu


def _bits(words):            # (..., 2) uint32 -> (..., 64) bits, bit r of word r // 32
    w = words.astype(np.uint64)
    v = w[..., 0] | (w[..., 1] << np.uint64(32))
    return ((v[..., None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(np.uint8)


@pytest.mark.parametrize("kind", ["random", "identity", "single_bits", "rows", "columns"])
def test_bit_transpose(kind):
    lib = abi.load_library()
    rng = np.random.default_rng(20260928)
    n = 64
    m = np.zeros((n, 64, 64), dtype=np.uint8)          # [matrix][lane p][bit r]
    if kind == "random":
        m = rng.integers(0, 2, size=m.shape, dtype=np.uint8)
    elif kind == "identity":
        m[:, np.arange(64), np.arange(64)] = 1
    elif kind == "single_bits":
        for i in range(n):
            m[i, (7 * i) % 64, (13 * i + 5) % 64] = 1
    elif kind == "rows":
        for i in range(n):
            m[i, i % 64, :] = 1
    else:
        for i in range(n):
            m[i, :, i % 64] = 1
    weights = (np.uint64(1) << np.arange(32, dtype=np.uint64))
    words = np.stack([(m[..., :32] * weights).sum(-1), (m[..., 32:] * weights).sum(-1)], axis=-1).astype(np.uint32)
    words = np.ascontiguousarray(words)
    out = np.zeros_like(words)
    out_lo = np.zeros((n, 64), dtype=np.uint32)
    rc = lib.snk_selftest_bit_transpose(0, words.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p),
                                        out_lo.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.snk_last_error().decode()
    got = _bits(out)                                   # [matrix][lane r][bit p]
    assert np.array_equal(got, m.transpose(0, 2, 1))
    lo = ((out_lo[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.uint8)
    assert np.array_equal(lo, m.transpose(0, 2, 1)[..., :32])
