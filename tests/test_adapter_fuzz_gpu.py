"""HIP-side adapter search fuzz: random adapter contexts -- adapter 6..64 nt with N inside,
1..4 adapters per mate, adaMis 0..3, adaMR in {0.3, 0.5, 0.7, 1}, adaEdge 1..8, trim and discard mode -- on reads
with planted whole / tail-truncated / HEAD-truncated (phase A of adapter_pos: the read starts inside the adapter,
src/read_filter.cpp:720-742) / mutated adapters; both kernels against the oracle, bit-exact records and counters.
The bit-sliced screen (screen_planes) and the closed-form early-exit decision (accept_exact) of the tiled kernel
see every phase with every budget here; the oracle's adapter_pos itself is pinned on the compiled reference by
tests/test_oracle_vs_ref.py."""
import numpy as np
import pytest

import snk_testlib as T
from soapnuke_amd import abi, synth
from test_gpu_parity import assert_same, run_hip_device

pytestmark = pytest.mark.gpu

B4 = np.frombuffer(b"ACGT", dtype=np.uint8)
N_CONTEXTS, READS = 56, 20000


def random_adapter(rng, lo=6, hi=64):
    n = int(rng.integers(lo, hi + 1))
    a = B4[rng.integers(0, 4, n)].copy()
    if rng.random() < 0.3:                                   # 'N' inside an adapter: an ordinary character (exact compare)
        a[rng.integers(0, n, int(rng.integers(1, 3)))] = ord("N")
    if rng.random() < 0.15:                                  # low-complexity adapters: long runs, many candidate offsets
        a[:] = B4[rng.integers(0, 4)]
        a[rng.integers(0, n, 2)] = B4[rng.integers(0, 4, 2)]
    return bytes(a).decode()


def plant(rng, seq, lens, L, adapters, frac):
    """writes adapter material into `frac` of the rows of seq (n x pitch uint8)"""
    n = seq.shape[0]
    rows = rng.choice(n, int(n * frac), replace=False)
    for r in rows:
        a = np.frombuffer(adapters[int(rng.integers(0, len(adapters)))].encode(), dtype=np.uint8).copy()
        la, rl = len(a), int(lens[r]) if lens is not None else L
        for k in rng.integers(0, la, int(rng.choice([0, 0, 0, 1, 2, 3, 4]))):      # substitutions (also across the budget)
            a[int(k)] = B4[rng.integers(0, 4)]
        mode = int(rng.integers(0, 4))
        if mode == 0 and rl >= la:                          # whole adapter somewhere (phase B), insert >= 0
            p = int(rng.integers(0, rl - la + 1))
            seq[r, p:p + la] = a
        elif mode == 1:                                     # tail-truncated: adapter prefix at the read end (phase C)
            k = int(rng.integers(1, min(la, rl) + 1))
            seq[r, rl - k:rl] = a[:k]
        elif mode == 2:                                     # head-truncated: read starts r1 characters into the adapter (phase A: 1..5)
            r1 = int(rng.integers(1, 9))
            k = min(la - r1, rl)
            if k > 0:
                seq[r, :k] = a[r1:r1 + k]
        else:                                               # adapter dimer with a short random insert in front
            p = int(rng.integers(0, 12))
            k = min(la, rl - p)
            if k > 0:
                seq[r, p:p + k] = a[:k]


def context(i):
    rng = np.random.default_rng(7000 + i)
    paired = i % 4 != 3
    L = 150 if i % 2 == 0 else 100
    var = i % 3 != 0
    na = [int(rng.integers(1, 5)) for _ in range(2)]
    if i % 7 == 0:
        na = [1, 1]
    ada = [[random_adapter(rng, 6, min(64, L // 2 - 8)) for _ in range(na[m])] for m in range(2)]
    kw = dict(adapters1=ada[0], ada_trim=int(rng.integers(0, 2)),
              ada_mis=(int(rng.integers(0, 4)), int(rng.integers(0, 4))),
              ada_mr=(float(rng.choice([0.3, 0.5, 0.7, 1.0])), float(rng.choice([0.3, 0.5, 0.7, 1.0]))),
              ada_edge=(int(rng.integers(1, 9)), int(rng.integers(1, 9))),
              low_qual=10, low_qual_ratio=0.3, min_read_length=int(rng.choice([30, 15, 50])))
    if paired:
        kw["adapters2"] = ada[1]
    d = synth.make_batch(READS, L, paired=paired, var_len=var, seed=9000 + i, dimer_frac=0.02,
                         adapters=(ada[0][0], ada[1][0]))
    for m in range(2 if paired else 1):
        plant(rng, d["seq"][m], d["len"][m], L, ada[m], 0.25)
    p = abi.default_params(paired=paired, max_read_len=L, **kw)
    return p, d, paired


@pytest.mark.parametrize("i", range(N_CONTEXTS))
def test_adapter_fuzz(i):
    p, d, paired = context(i)
    want = T.run_oracle(p, d)
    assert int((want["rec"][0]["adacut_pos"] >= 0).sum()) > 100            # adapters are found ...
    for kernel in (0, 1):
        assert_same(p, run_hip_device(p, d, kernel, chunks=2), want, paired)


def test_phase_a_dimers_reach_the_tiled_kernel():
    """README adapters, C2 parameters, 5 % adapter dimers with negative inserts: reads whose first base is adapter
    character r1 = 1..5 must come back with adapter position 0 (phase A) from the tiled kernel."""
    d = synth.make_batch(40000, 150, paired=True, seed=31, dimer_frac=0.05)
    p = abi.default_params(paired=True, max_read_len=150, adapters1=[synth.ADAPTER1], adapters2=[synth.ADAPTER2],
                           ada_trim=1, low_qual=10, low_qual_ratio=0.1)
    want = T.run_oracle(p, d)
    got = run_hip_device(p, d, 2)
    assert_same(p, got, want, True)
    a1 = np.frombuffer(synth.ADAPTER1.encode(), dtype=np.uint8)
    phase_a = [r for r in range(d["n"]) if any(np.array_equal(d["seq"][0][r, :20], a1[k:k + 20]) for k in range(1, 6))]
    assert len(phase_a) > 100
    # adapter at read position 0 (adacut_pos = length - 0): nothing is left of mate 1
    assert all(want["rec"][0]["adacut_pos"][r] == 150 and got["rec"][0]["adacut_pos"][r] == 150 for r in phase_a)


@pytest.mark.parametrize("mis", [(4, 5), (6, 4), (9, 3)])
def test_adapter_budgets_above_three(mis):
    """adaMis 4..9: the unary counters of the screen stop at four mismatches, offsets whose own budget is four or more all
    go to the exact decision -- the tiled kernel takes them (kernel = 2), whatever the budget"""
    rng = np.random.default_rng(sum(mis))
    ada = [[random_adapter(rng, 24, 64) for _ in range(2)] for _ in range(2)]
    d = synth.make_batch(READS, 150, paired=True, var_len=True, seed=4242 + mis[0], adapters=(ada[0][0], ada[1][0]))
    for m in range(2):
        plant(rng, d["seq"][m], d["len"][m], 150, ada[m], 0.3)
    p = abi.default_params(paired=True, max_read_len=150, adapters1=ada[0], adapters2=ada[1], ada_trim=1, ada_mis=mis,
                           ada_mr=(0.5, 0.7), ada_edge=(6, 3), low_qual=10, low_qual_ratio=0.3, min_read_length=30)
    want = T.run_oracle(p, d)
    assert_same(p, run_hip_device(p, d, 2, chunks=2), want, True)


# ---- round 3: adapter lists longer than four, lower-case adapters, lists longer than SNK_MAX_ADAPTERS

def _lower_some(rng, a):
    b = bytearray(a.encode())
    for k in rng.integers(0, len(b), int(rng.integers(1, 4))):
        b[int(k)] |= 0x20
    return b.decode()


@pytest.mark.parametrize("i", range(16))
def test_long_adapter_lists_and_lower_case_on_the_fast_paths(i):
    """1..12 adapters per mate (i % 4 == 3: 20..40, through snk_params.adapter_list), a quarter of them with lower-case
    characters, reads with lower-case stretches that such adapters can match: kernel = 2 must take the configuration
    (tiled kernel up to 256 positions, the long-read path beyond), first adapter of the list with a hit wins."""
    rng = np.random.default_rng(8100 + i)
    paired = i % 3 != 2
    L = [150, 100, 250, 400][i % 4] if i % 5 else 150
    na = [int(rng.integers(20, 41)) if i % 4 == 3 else int(rng.integers(5, 13)) for _ in range(2)]
    ada = [[random_adapter(rng, 8, min(64, L // 2 - 8)) for _ in range(na[m])] for m in range(2)]
    for m in range(2):
        for k in range(len(ada[m])):
            if rng.random() < 0.25:
                ada[m][k] = _lower_some(rng, ada[m][k])
    kw = dict(adapters1=ada[0], ada_trim=int(rng.integers(0, 2)), ada_mis=(int(rng.integers(0, 4)), int(rng.integers(0, 4))),
              ada_mr=(float(rng.choice([0.3, 0.5, 0.7])), float(rng.choice([0.3, 0.5, 0.7]))),
              ada_edge=(int(rng.integers(1, 8)), int(rng.integers(1, 8))), low_qual=10, low_qual_ratio=0.3)
    if paired:
        kw["adapters2"] = ada[1]
    n = 12000 if L <= 256 else 4000
    d = synth.make_batch(n, L, paired=paired, var_len=bool(i % 2), seed=9100 + i, adapters=(ada[0][0].upper(), ada[1][0].upper()))
    for m in range(2 if paired else 1):
        plant(rng, d["seq"][m], d["len"][m], L, ada[m], 0.3)
        rows = rng.choice(n, n // 10, replace=False)                       # reads with lower-case stretches (and the planted lower-case adapters)
        blk = d["seq"][m][rows, :L]
        d["seq"][m][rows, :L] = np.where((rng.random(blk.shape) < 0.2) & (blk >= 65) & (blk <= 90), blk | 0x20, blk)
        if d["len"][m] is not None:
            beyond = np.arange(L)[None, :] >= d["len"][m][:, None].astype(np.int32)
            d["seq"][m][:, :L][beyond] = 0xEE
    p = abi.default_params(paired=paired, max_read_len=L, **kw)
    want = T.run_oracle(p, d)
    assert int((want["rec"][0]["adacut_pos"] >= 0).sum()) > 50
    assert_same(p, run_hip_device(p, d, 2, chunks=2), want, paired)
    assert_same(p, run_hip_device(p, d, 1), want, paired)


def any_length_context(i, n_pairs):
    """parameters and batch of context i of test_adapters_of_any_length_on_the_tiled_kernel (tests/isa_interp_capture.py replays
    some of them from the gfx950 assembly at a smaller size)"""
    rng = np.random.default_rng(8100 + i)
    L = 150 if i % 3 else 250
    if i % 4 == 0:
        lens_ = [int(rng.integers(1, 6)), int(rng.integers(2, 6))]          # shorter than 6
    elif i % 4 == 1:
        lens_ = [int(rng.integers(65, 121)), int(rng.integers(65, L))]       # longer than the 64 match bits
    elif i % 4 == 2:
        lens_ = [int(rng.integers(130, 201)), int(rng.integers(6, 65))]      # S - 1 beyond 64 screened cells; mixed with a short one
    else:
        lens_ = [int(rng.integers(6, 20)), int(rng.integers(6, 20))]         # adaEdge beyond the adapter (below)
    ada = [[random_adapter(rng, n, n)] + ([random_adapter(rng, 6, 64)] if rng.random() < 0.5 else []) for n in lens_]
    edge = (int(rng.integers(1, 9)), int(rng.integers(1, 9)))
    if i % 4 == 3:
        edge = (len(ada[0][0]) + int(rng.integers(0, 4)), len(ada[1][0]) + 1)
    d = synth.make_batch(n_pairs, L, paired=True, var_len=bool(i % 2), seed=9500 + i, adapters=(ada[0][0], ada[1][0]))
    for m in range(2):
        plant(rng, d["seq"][m], d["len"][m], L, ada[m], 0.3)
    p = abi.default_params(paired=True, max_read_len=L, adapters1=ada[0], adapters2=ada[1], ada_trim=int(rng.integers(0, 2)),
                           ada_mis=(int(rng.integers(0, 4)), int(rng.integers(0, 4))), ada_mr=(float(rng.choice([0.3, 0.5, 0.7])), 0.5),
                           ada_edge=edge, low_qual=10, low_qual_ratio=0.3, min_read_length=30)
    return p, d


# ---- round 4: adapters of any length on the tiled kernel (65..200 characters, shorter than 6, adaEdge beyond the adapter)
@T.first_contact
@pytest.mark.parametrize("i", range(12))
def test_adapters_of_any_length_on_the_tiled_kernel(i):
    """the reference takes adapters of any length (src/read_filter.cpp:707-790); the bit paths took 6..64
    characters and everything else fell back to the generic kernel (40 x slower).  Now kernel = 2 accepts 1..255 characters:
    the screen looks at an adapter's first 64 characters, survivors of longer adapters are decided character by character, reads
    shorter than the adapter take the sequential matcher in their lane, adaEdge may exceed the adapter (no phase C)."""
    p, d = any_length_context(i, READS // 2)
    want = T.run_oracle(p, d)
    assert_same(p, run_hip_device(p, d, 2, chunks=2), want, True)          # kernel = 2: the fast path must take it
