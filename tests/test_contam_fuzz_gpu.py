"""HIP-side contaminant screening fuzz: the bit-parallel matchers of snk_contam.hip -- the unary
counter screen + exact decision of hasContam() (src/read_filter.cpp:507-603) and the block screen in front of the
window walk of global_contam_pos() (:961-1062) -- against the oracle (pinned on the compiled reference's own
functions by tests/test_oracle_vs_ref.py), through the whole filter: bit-exact records and counters.

Contexts: contaminants of 8..64 nt (plus some the bit paths must hand to the sequential matchers: over 64 nt, lower
case), with 'N', low-complexity ones (long carried windows across the lays of global_contam_pos), ctMatchR 0.2..1,
adaMis 0..7, adaEdge 1..12, global contaminants with match ratios 0.3..1 and 0..4 mismatches; reads of 150 / 250 /
variable length (also shorter than the contaminant), random, low-complexity and with planted whole / head- / tail-
truncated / mutated / N-sprinkled copies."""
import numpy as np
import pytest

import snk_testlib as T
from soapnuke_amd import abi, synth
from test_gpu_parity import assert_same, run_hip_device

pytestmark = pytest.mark.gpu

B4 = np.frombuffer(b"ACGT", dtype=np.uint8)
N_CONTEXTS, READS = 48, 12000


def random_contam(rng, lo=8, hi=64, style=None):
    n = int(rng.integers(lo, hi + 1))
    style = int(rng.integers(0, 6)) if style is None else style
    if style == 0:                                          # homopolymer with a few other letters
        a = np.full(n, B4[rng.integers(0, 4)], dtype=np.uint8)
        a[rng.integers(0, n, 2)] = B4[rng.integers(0, 4, 2)]
    elif style == 1:                                        # short period
        per = B4[rng.integers(0, 4, int(rng.integers(2, 5)))]
        a = np.resize(per, n).copy()
    else:
        a = B4[rng.integers(0, 4, n)].copy()
    if rng.random() < 0.25:
        a[rng.integers(0, n, int(rng.integers(1, 3)))] = ord("N")
    return bytes(a).decode()


def plant(rng, seq, lens, L, contams, frac):
    n = seq.shape[0]
    for r in rng.choice(n, int(n * frac), replace=False):
        a = np.frombuffer(contams[int(rng.integers(0, len(contams)))].upper().encode(), dtype=np.uint8).copy()
        la, rl = len(a), int(lens[r]) if lens is not None else L
        if rl < 4:
            continue
        for k in rng.integers(0, la, int(rng.choice([0, 0, 0, 1, 1, 2, 3, 5]))):
            a[int(k)] = B4[rng.integers(0, 4)]
        mode = int(rng.integers(0, 4))
        if mode == 0 and rl >= la:                          # whole, anywhere (also flush with either end)
            p = int(rng.choice([0, rl - la, int(rng.integers(0, rl - la + 1))]))
            seq[r, p:p + la] = a
        elif mode == 1:                                     # its tail on the read start
            k = int(rng.integers(1, min(la, rl) + 1))
            seq[r, :k] = a[la - k:]
        elif mode == 2:                                     # its head on the read end
            k = int(rng.integers(1, min(la, rl) + 1))
            seq[r, rl - k:rl] = a[:k]
        else:                                               # a middle piece somewhere
            k = int(rng.integers(4, la + 1))
            s = int(rng.integers(0, la - k + 1))
            k = min(k, rl)
            p = int(rng.integers(0, rl - k + 1))
            seq[r, p:p + k] = a[s:s + k]
        if rng.random() < 0.2:                              # 'N' in the read: neither match nor mismatch for hasContam
            seq[r, rng.integers(0, rl, int(rng.integers(1, 4)))] = ord("N")


def low_complexity(rng, seq, lens, L, frac):
    n = seq.shape[0]
    for r in rng.choice(n, int(n * frac), replace=False):
        rl = int(lens[r]) if lens is not None else L
        if rng.random() < 0.5:
            row = np.resize(B4[rng.integers(0, 4, int(rng.integers(1, 5)))], rl).copy()
        else:
            row = B4[rng.integers(0, 2, rl) + int(rng.integers(0, 3))].copy()
        k = int(rng.integers(0, 6))
        if k:
            row[rng.integers(0, rl, k)] = B4[rng.integers(0, 4, k)]
        seq[r, :rl] = row


def context(i, long_L=None, reads=READS):
    rng = np.random.default_rng(5100 + i)
    paired = i % 4 != 3
    L = 250 if i % 5 == 4 else (150 if i % 2 == 0 else 100)
    if long_L:
        L = long_L
    var = i % 3 == 1
    kw = dict(low_qual=10, low_qual_ratio=0.5, min_read_length=15,
              ada_mis=(int(rng.integers(0, 4)), int(rng.integers(0, 4))),
              ada_edge=(int(rng.integers(1, 13)), int(rng.integers(1, 13))))
    if i % 11 == 10:
        kw["ada_mis"] = (4, 1)                              # budgets over 3: the counters stop at four, such offsets are all decided exactly
    if i % 11 == 9:
        kw["ada_mis"] = (7, 5)
    cts = []
    kind = i % 3                                            # 0: contam lists, 1: global only, 2: both
    if kind != 1:
        for m in range(2 if paired else 1):
            k = int(rng.integers(1, 4))
            cs = [random_contam(rng, 8, min(64, L // 2)) for _ in range(k)]
            if i % 13 == 5:
                cs[0] = random_contam(rng, 70, 90, style=3)           # over 64: sequential
            if i % 13 == 7:
                cs[0] = cs[0].lower()                                  # not upper case: sequential (and never matches)
            mrs = [str(rng.choice([0.2, 0.3, 0.5, 0.8, 1.0])) for _ in range(k)]
            kw["contam%d" % (m + 1)] = ",".join(cs)
            if m == 0:
                kw["ct_match_r"] = ",".join(mrs) if k > 1 else mrs[0]
                k0 = k
            elif (k > 1) != (k0 > 1) or k != k0:            # one ctMatchR list serves both mates: same entry count
                cs = (cs * 3)[:k0]
                kw["contam2"] = ",".join(cs)
            cts += cs
    if kind != 0:
        k = int(rng.integers(1, 3))
        gs, mrs, mms = [], [], []
        for _ in range(k):
            g = random_contam(rng, 10, min(64, L // 2))
            mr = float(rng.choice([0.3, 0.5, 0.7, 0.9, 1.0]))
            mml = int(np.float32(len(g)) * np.float32(mr))
            mm = int(rng.integers(0, min(5, mml)))          # snk_create admits 0..4 and < the match length
            gs.append(g), mrs.append(str(mr)), mms.append(str(mm))
        if i % 13 == 9:
            gs[0] = random_contam(rng, 70, 90, style=3)      # over 64: the window walk for every read
        kw.update(global_contams=",".join(gs), g_mrs=",".join(mrs), g_mms=",".join(mms))
        cts += gs
        # both strands are screened
        comp = bytes.maketrans(b"ACGTN", b"TGCAN")
        cts += [g.encode().translate(comp)[::-1].decode() for g in gs]
    if i % 6 == 0:
        kw["contam_trim"] = 1
    d = synth.make_batch(reads, L, paired=paired, var_len=var, seed=6100 + i)
    for m in range(2 if paired else 1):
        low_complexity(rng, d["seq"][m], d["len"][m], L, 0.1)
        plant(rng, d["seq"][m], d["len"][m], L, cts, 0.3)
        if long_L:                                          # copies across the 256-offset blocks of the long-read kernel and their 64-position reach
            S, lens = d["seq"][m], d["len"][m]
            for r in rng.choice(reads, reads // 4, replace=False):
                a = np.frombuffer(cts[int(rng.integers(0, len(cts)))].upper().encode(), dtype=np.uint8).copy()
                rl = int(lens[r]) if lens is not None else L
                for k in rng.integers(0, len(a), int(rng.choice([0, 0, 1, 2]))):
                    a[int(k)] = B4[rng.integers(0, 4)]
                q = int(rng.choice([256, 320, 512, 576, 768, 832])) + int(rng.integers(-len(a) - 2, 3))
                if 0 <= q and q + len(a) <= rl:
                    S[r, q:q + len(a)] = a
    if var:                                                 # some reads shorter than any contaminant
        for m in range(2 if paired else 1):
            rows = rng.choice(reads, reads // 50, replace=False)
            d["len"][m][rows] = rng.integers(1, 12, len(rows))
            if long_L:                                      # ... and some between a contaminant's length and one block
                rows = rng.choice(reads, reads // 20, replace=False)
                d["len"][m][rows] = np.minimum(d["len"][m][rows], rng.integers(12, 331, len(rows)))   # (only ever shorter: the rows end where the reads did)
    p = abi.default_params(paired=paired, max_read_len=L, **kw)
    return p, d, paired, kw


@pytest.mark.parametrize("i", range(N_CONTEXTS))
def test_contam_fuzz(i):
    p, d, paired, kw = context(i)
    want = T.run_oracle(p, d)
    got = run_hip_device(p, d, 2, chunks=2)                 # the tiled path: verdicts from snk_contam.hip
    assert_same(p, got, want, paired)
    if not kw.get("contam_trim"):                           # ... and the screen did see contaminated reads
        assert int(want["sum"][abi.FS_CONTAM]) + int(want["sum"][abi.FS_GCONTAM]) > 0


@pytest.mark.parametrize("i", range(20))
def test_contam_fuzz_long_reads(i):
    """reads of 257..1000 positions: the same bit paths block by block on the plane store of the long-read path
    (snk_long_contam_kernel) -- alignments hanging off the read's start in the first block, off its end in the final one, whole ones
    in the block of their offset -- against the oracle"""
    L = (300, 320, 321, 500, 576, 700, 1000)[i % 7]
    p, d, paired, kw = context(100 + i, long_L=L, reads=2000 if L <= 576 else 1200)
    want = T.run_oracle(p, d)
    got = run_hip_device(p, d, 2, chunks=2)
    assert_same(p, got, want, paired)
    if not kw.get("contam_trim"):
        assert int(want["sum"][abi.FS_CONTAM]) + int(want["sum"][abi.FS_GCONTAM]) > 0


@pytest.mark.parametrize("i", range(10))
def test_global_contaminants_outside_the_event_walk_range(i):
    """glob_cotm_mM above 4, or a match length not above the mismatch number -- settings the reference accepts
    and whose score arithmetic lets "dead" windows pass the hit test: the kernels walk the lays cell by cell there
    (gc_lay_cells); HIP vs oracle on both kernels, and the oracle's matcher itself is pinned on the compiled reference for
    the same settings by tests/test_oracle_vs_ref.py::test_contam_matchers_fuzz."""
    rng = np.random.default_rng(6600 + i)
    L, paired = (150, True) if i % 2 == 0 else (100, False)
    gs, mrs, mms = [], [], []
    for _ in range(int(rng.integers(1, 3))):
        g = random_contam(rng, 10, 48)
        mr = float(rng.choice([0.1, 0.2, 0.3, 0.5]))
        mml = int(np.float32(len(g)) * np.float32(mr))
        mm = int(rng.integers(5, 9)) if rng.random() < 0.5 else int(rng.integers(mml, mml + 3))
        gs.append(g), mrs.append(str(mr)), mms.append(str(mm))
    kw = dict(global_contams=",".join(gs), g_mrs=",".join(mrs), g_mms=",".join(mms))
    d = synth.make_batch(6000, L, paired=paired, var_len=bool(i % 3), seed=6700 + i)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    for m in range(2 if paired else 1):
        plant(rng, d["seq"][m], d["len"][m], L, gs + [g.encode().translate(comp)[::-1].decode() for g in gs], 0.3)
    p = abi.default_params(paired=paired, max_read_len=L, **kw)
    want = T.run_oracle(p, d)
    for kernel in (2, 1):
        assert_same(p, run_hip_device(p, d, kernel, chunks=2), want, paired)
