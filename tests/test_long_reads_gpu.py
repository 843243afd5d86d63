"""Reads of 257..1000 positions on the fast path (reference limit READ_MAX_LEN 1000,
src/global_variable.h:9): snk_long.hip -- byte-parallel stat_read per lane, the bit-sliced adapter search on 320-position
blocks of the read, LDS histograms over 128-position blocks -- against the oracle, bit-exact records and counters, with
`kernel=2` (fast path required: the generic kernel is not an option there)."""
import numpy as np
import pytest

import snk_testlib as T
from cases import PE_CASES, se_kwargs
from soapnuke_amd import abi, synth
from test_gpu_parity import assert_same, run_hip_device

pytestmark = pytest.mark.gpu

B4 = np.frombuffer(b"ACGT", dtype=np.uint8)


def plant_everywhere(d, adapters, seed, L):
    """whole / truncated / mutated adapter copies at block boundaries (255, 256, 319, 320, 512 ...), read ends and starts"""
    rng = np.random.default_rng(seed)
    for m in range(len(d["seq"])):
        S, lens = d["seq"][m], d["len"][m]
        n = S.shape[0]
        a0 = np.frombuffer(adapters[m].encode(), dtype=np.uint8)
        for r in rng.choice(n, n // 3, replace=False):
            rl = int(lens[r]) if lens is not None else L
            a = a0.copy()
            for k in rng.integers(0, len(a), int(rng.choice([0, 0, 1, 2, 4]))):
                a[int(k)] = B4[rng.integers(0, 4)]
            kind = int(rng.integers(0, 5))
            if kind == 0:                                    # around a block boundary
                edge = int(rng.choice([256, 320, 512, 576, 768, 832]))
                p = edge + int(rng.integers(-70, 8))
            elif kind == 1:                                  # tail-truncated at the read end (phase C)
                p = rl - int(rng.integers(1, len(a)))
            elif kind == 2:                                  # head-truncated at the read start (phase A)
                r1 = int(rng.integers(1, 8))
                k = min(len(a) - r1, rl)
                if k > 0:
                    S[r, :k] = a[r1:r1 + k]
                continue
            else:
                p = int(rng.integers(0, max(rl - 8, 1)))
            if p < 0 or p >= rl:
                continue
            k = min(len(a), rl - p)
            S[r, p:p + k] = a[:k]


CASES = [
    # (L, paired, var_len, params)
    (257, True, True, "C3_full"),
    (300, True, False, "C2_adatrim_lowq"),
    (319, True, False, "C2_adatrim_lowq"),
    (320, True, True, "C3_full"),
    (321, False, False, "C2_adatrim_lowq"),
    (500, True, True, "C2_adatrim_lowq"),
    (512, True, False, "C3_full"),
    (640, False, True, "C3_full"),
    (1000, True, True, "C3_full"),
    (1000, True, False, "defaults"),
    (1000, False, False, "C2_adatrim_lowq"),
]


@pytest.mark.parametrize("L,paired,var,name", CASES)
def test_long_reads(L, paired, var, name):
    n = 3000 if L <= 512 else 1500
    d = synth.make_batch(n, L, paired=paired, var_len=var, seed=300 + L)
    kw = PE_CASES[name] if paired else se_kwargs(PE_CASES[name])
    if "adapters1" in kw:
        plant_everywhere(d, (kw["adapters1"][0], kw.get("adapters2", kw["adapters1"])[0]), 17 + L, L)
    rng = np.random.default_rng(L)
    for m in range(len(d["seq"])):                          # lower case, other letters' neighbours, short reads: the per-lane fallback
        rows = rng.choice(n, n // 40, replace=False)
        d["seq"][m][rows, 3] = ord("a")
        if var:
            rows = rng.choice(n, n // 50, replace=False)
            d["len"][m][rows] = rng.integers(1, 64, len(rows))
    p = abi.default_params(paired=paired, max_read_len=L, **kw)
    want = T.run_oracle(p, d)
    if "adapters1" in kw:
        assert int((want["rec"][0]["adacut_pos"] >= 0).sum()) > n // 10
    assert_same(p, run_hip_device(p, d, 2, chunks=2), want, paired)


def test_long_reads_several_adapters_and_budgets():
    """four adapters per mate incl. 'N', adaMis 2, discard mode: the first adapter of the list with a hit decides, whatever
    block its hit is in"""
    L, n = 700, 2000
    rng = np.random.default_rng(5)
    ada = [["".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(12, 60)))) for _ in range(4)] for _ in range(2)]
    ada[0][1] = ada[0][1][:5] + "N" + ada[0][1][6:]
    d = synth.make_batch(n, L, paired=True, var_len=True, seed=77, adapters=(ada[0][0], ada[1][0]))
    for a in range(4):
        plant_everywhere(d, (ada[0][a], ada[1][a]), 100 + a, L)
    p = abi.default_params(paired=True, max_read_len=L, adapters1=ada[0], adapters2=ada[1], ada_trim=0, ada_mis=(2, 1), ada_mr=(0.5, 0.7),
                           ada_edge=(4, 8), low_qual=10, low_qual_ratio=0.4, min_read_length=30)
    assert_same(p, run_hip_device(p, d, 2), T.run_oracle(p, d), True)


def test_long_reads_contaminants_and_duplicates():
    """contaminant verdicts (sequential matchers inside the decide kernel beyond 256 positions) and duplicate / tile flags"""
    from cases import CONTAM_CASES, plant_contams
    L, n = 400, 2500
    kw = dict(CONTAM_CASES["both_discard"])
    d = synth.make_batch(n, L, paired=True, var_len=True, seed=123)
    plant_contams(d, kw)
    p = abi.default_params(paired=True, max_read_len=L, rmdup=1, **kw)
    rng = np.random.default_rng(3)
    dup = (rng.random(n) < 0.1).astype(np.uint8) | ((rng.random(n) < 0.03).astype(np.uint8) << 1)
    want = T.run_oracle(p, d, dup=dup)
    assert_same(p, run_hip_device(p, d, 2, dup=dup, chunks=2), want, True)


def test_long_reads_quality_range_error():
    """a quality below the offset is the reference's heap corruption (src/peprocess.cpp:1196): reported with the read's index"""
    L, n = 600, 1000
    d = synth.make_batch(n, L, paired=True, seed=9)
    d["qual"][1][700, 555] = 20                                # '!' - 13
    p = abi.default_params(paired=True, max_read_len=L, **PE_CASES["defaults"])
    got = run_hip_device(p, d, 2)
    assert got["err"][0] == abi.SNK_E_QUAL_RANGE if hasattr(abi, "SNK_E_QUAL_RANGE") else got["err"][0] != 0
    want = T.run_oracle(p, d)
    assert tuple(got["err"]) == tuple(want["err"])


def test_long_reads_plane_store_group_edges_and_regrowth():
    """the plane store holds groups of 64 reads (snk_long_prep_kernel): batches that end inside a group, one read, and batches
    growing through one context (the scratch is regrown) -- SE and PE"""
    from soapnuke_amd.filter import FilterContext, records_to_numpy
    L = 1000
    for paired in (True, False):
        kw = PE_CASES["C3_full"] if paired else se_kwargs(PE_CASES["C3_full"])
        p = abi.default_params(paired=paired, max_read_len=L, **kw)
        ctx = FilterContext(p, device=0)
        for n in (1, 63, 64, 65, 129, 1000):
            d = synth.make_batch(n, L, paired=paired, var_len=True, seed=500 + n)
            if n >= 63:
                plant_everywhere(d, (kw["adapters1"][0], kw.get("adapters2", kw["adapters1"])[0]), n, L)
            dev = ctx.upload(d)
            rec = ctx.alloc_records(n)
            ctx.clear()
            ctx.filter_batch(ctx.make_batch(dev), rec, kernel=2)
            s, mx, err = ctx.fetch()
            got = dict(rec=[records_to_numpy(rec[0]), records_to_numpy(rec[1])], sum=s, max=mx, err=err)
            assert_same(p, got, T.run_oracle(p, d), paired)
        ctx.close()


def long_any_length_context(L, alen, edge, mr, n=1500):
    """parameters and batch of test_long_reads_adapters_of_any_length (tests/isa_interp_capture.py replays some from the gfx950 assembly)"""
    rng = np.random.default_rng(1000 + alen)
    ada = ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, alen)) for _ in range(2)]
    d = synth.make_batch(n, L, paired=True, var_len=True, seed=900 + alen)
    plant_everywhere(d, ada, 41 + alen, L)
    for m in range(2):                                      # truncated copies at the read's end, at every overlap from edge - 2 up: phase C across the last two blocks
        a0 = np.frombuffer(ada[m].encode(), dtype=np.uint8)
        rows = rng.choice(n, n // 4, replace=False)
        for r in rows:
            rl = int(d["len"][m][r])
            k = int(rng.integers(max(1, min(edge, alen) - 2), alen + 1))
            if k <= rl:
                d["seq"][m][r, rl - k:rl] = a0[:k]
    p = abi.default_params(paired=True, max_read_len=L, adapters1=[ada[0]], adapters2=[ada[1]], ada_trim=1, ada_mis=(2, 1), ada_mr=(mr, mr),
                           ada_edge=(edge, edge), low_qual=10, low_qual_ratio=0.5)
    return p, d


@T.first_contact
@pytest.mark.parametrize("L,alen,edge,mr", [(600, 100, 6, 0.5), (600, 3, 2, 0.7), (600, 200, 6, 0.5), (1000, 255, 10, 0.3), (600, 40, 60, 0.5), (640, 5, 6, 1.0),
                                            (1000, 130, 3, 0.5)])
def test_long_reads_adapters_of_any_length(L, alen, edge, mr):
    """the long-read path takes the adapters the tiled kernel takes -- 1..255 characters, adaEdge beyond the adapter --
    instead of falling back to the generic kernel (`kernel=2`: fast path required).  A block of the read is told how much of the
    read is left and which offsets are its own (csrc/snk_adapter_bits.hip.h): the phase C offsets of an adapter of more than 64
    characters can lie in the read's last TWO blocks, where the later block's hit comes first (descending offsets,
    src/read_filter.cpp:765-788)."""
    n = 1500
    p, d = long_any_length_context(L, alen, edge, mr, n)
    want = T.run_oracle(p, d)
    if alen >= 5:
        assert int((want["rec"][0]["adacut_pos"] >= 0).sum()) > n // 10
    assert_same(p, run_hip_device(p, d, 2, chunks=2), want, True)
