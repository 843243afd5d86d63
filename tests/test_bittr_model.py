"""CPU model of the 64 x 64 bit-matrix transpose of soapnuke_amd/csrc/snk_bittr.hip.h: the same six
butterfly stages (lane-index bit k <-> bit-index bit k) with the lane exchanges written as numpy
permutations.  It documents the network and pins its masks / rotations without a GPU; the device code
itself is checked by tests/test_bittr_gpu.py."""
import numpy as np

LANE = np.arange(64)
M32 = np.uint64(0xFFFFFFFF)
MK = {8: 0x00FF00FF, 4: 0x0F0F0F0F, 2: 0x33333333, 1: 0x55555555}


def _rotr(x, n):
    n = n.astype(np.uint64)
    return ((x >> n) | (x << (np.uint64(32) - n))) & M32


def _stage16(x):
    # v_permlane16_swap of the word with itself: a = word of the even-row lane of each pair of rows, b = of the odd-row lane
    a, b = x.copy(), x.copy()
    for row in (0, 2):
        a[16 * (row + 1):16 * (row + 2)], b[16 * row:16 * (row + 1)] = b[16 * row:16 * (row + 1)].copy(), a[16 * (row + 1):16 * (row + 2)].copy()
    even = (LANE & 16) == 0
    return np.where(even, (a & np.uint64(0xFFFF)) | ((b << np.uint64(16)) & M32), (a >> np.uint64(16)) | (b & np.uint64(0xFFFF0000)))


def _stage(x, k):
    partner = x[LANE ^ k]                      # DPP: row_ror:8, row_shl/shr:4 with bank masks, quad_perm
    odd = (LANE & k) != 0
    y = _rotr(partner, np.where(odd, k, 32 - k))
    keep = np.where(odd, np.uint64(~MK[k] & 0xFFFFFFFF), np.uint64(MK[k]))
    return (x & keep) | (y & ~keep & M32)


def bit_transpose64(lo, hi):
    lo, hi = lo.copy(), hi.copy()
    lo[32:], hi[:32] = hi[:32].copy(), lo[32:].copy()          # v_permlane32_swap(lo, hi)
    out = []
    for x in (lo, hi):
        x = _stage16(x)
        for k in (8, 4, 2, 1):
            x = _stage(x, k)
        out.append(x)
    return out


def test_model_transposes():
    rng = np.random.default_rng(7)
    for _ in range(20):
        m = rng.integers(0, 2, size=(64, 64)).astype(np.uint64)          # m[lane p][bit r]
        w = (np.uint64(1) << np.arange(32, dtype=np.uint64))
        lo, hi = (m[:, :32] * w).sum(1), (m[:, 32:] * w).sum(1)
        tl, th = bit_transpose64(lo, hi)
        got = np.concatenate([(tl[:, None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1),
                              (th[:, None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1)], axis=1)
        assert np.array_equal(got, m.T)


def test_model_is_an_involution():
    rng = np.random.default_rng(8)
    lo = rng.integers(0, 2 ** 32, size=64, dtype=np.uint64)
    hi = rng.integers(0, 2 ** 32, size=64, dtype=np.uint64)
    a, b = bit_transpose64(*bit_transpose64(lo, hi))
    assert np.array_equal(a, lo) and np.array_equal(b, hi)
