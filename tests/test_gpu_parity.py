"""GPU parity tests proper: the HIP path, called through the C ABI, against the
oracle on the same seeded inputs -- bit-exact on per-read records, every stats
counter and the max block."""
import ctypes as C
import os

import numpy as np
import pytest

import snk_testlib as T
from cases import CONTAM_CASES, PE_CASES, contam_kwargs, plant_contams, se_kwargs
from soapnuke_amd import abi, synth

pytestmark = pytest.mark.gpu

KERNELS = [1, 0]   # generic, auto (tiled where supported)


def run_hip_host(lib, p, d, first_index=0, dup=None):
    """through snk_filter_batch(): host pointers in, host records out"""
    ctx = lib.snk_create(C.byref(p), 0)
    assert ctx, lib.snk_last_error()
    try:
        b = T.host_batch(d, first_index, dup)
        r1 = np.zeros(b.n, dtype=abi.record_dtype())
        r2 = np.zeros(b.n, dtype=abi.record_dtype())
        rc = lib.snk_filter_batch(ctx, C.byref(b), r1.ctypes.data, r2.ctypes.data)
        assert rc == 0, lib.snk_last_error()
        s, mx = T.new_stats(p)
        err = abi.Error()
        assert lib.snk_stats_fetch(ctx, s.ctypes.data, mx.ctypes.data, C.byref(err), None) == 0
        return dict(rec=[r1, r2], sum=s, max=mx, err=(err.code, err.mate, err.index))
    finally:
        lib.snk_destroy(ctx)


def run_hip_device(p, d, kernel, first_index=0, dup=None, chunks=1):
    """through snk_filter_batch_device(): torch-owned device memory, current stream"""
    import torch
    from soapnuke_amd.filter import FilterContext, records_to_numpy
    ctx = FilterContext(p, device=0)
    dev = ctx.upload(d)
    n = d["n"]
    rec = ctx.alloc_records(n)
    dupt = None if dup is None else torch.from_numpy(np.ascontiguousarray(dup, dtype=np.uint8)).cuda()
    edges = np.linspace(0, n, chunks + 1).astype(int)
    for a, z in zip(edges[:-1], edges[1:]):
        if z == a:
            continue
        sub = {"n": int(z - a), "L": dev["L"], "pitch": dev["pitch"],
               "seq": [x[a:z] for x in dev["seq"]], "qual": [x[a:z] for x in dev["qual"]],
               "len": [None if x is None else x[a:z] for x in dev["len"]]}
        b = ctx.make_batch(sub, first_index + int(a), None if dupt is None else dupt[a:z])
        ctx.filter_batch(b, [rec[0][a:z], rec[1][a:z]], kernel=kernel)
    s, mx, err = ctx.fetch()
    out = dict(rec=[records_to_numpy(rec[0]), records_to_numpy(rec[1])], sum=s, max=mx, err=err)
    ctx.close()
    return out


def assert_same(p, got, want, paired):
    assert got["err"][0] == 0, got["err"]
    for m in range(2 if paired else 1):
        bad = np.nonzero(got["rec"][m] != want["rec"][m])[0]
        assert len(bad) == 0, (m, len(bad), bad[:5], got["rec"][m][bad[:3]], want["rec"][m][bad[:3]])
    assert np.array_equal(got["sum"], want["sum"]), T.describe_stats_diff(p, got["sum"], want["sum"])
    assert np.array_equal(got["max"], want["max"]), (got["max"], want["max"])


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", sorted(PE_CASES))
def test_pe150_cases(name, kernel):
    d = synth.make_batch(20000, 150, paired=True, var_len=(name in ("hard_lq_trim", "C3_full")), seed=21)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES[name])
    assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), True)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", ["defaults", "C2_adatrim_lowq", "C3_full", "hard_lq_trim", "short_adapter_edge"])
def test_se100_cases(name, kernel):
    d = synth.make_batch(20000, 100, paired=False, var_len=(name == "C3_full"), seed=22)
    p = abi.default_params(paired=False, max_read_len=100, **se_kwargs(PE_CASES[name]))
    assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), False)


@pytest.mark.parametrize("kernel", KERNELS)
def test_pe250_full(kernel):
    d = synth.make_batch(8000, 250, paired=True, seed=23)
    p = abi.default_params(paired=True, max_read_len=250, **PE_CASES["C3_full"])
    assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), True)


def test_host_pointer_entry(snk_lib):
    d = synth.make_batch(5000, 150, paired=True, seed=24)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C2_adatrim_lowq"])
    assert_same(p, run_hip_host(snk_lib, p, d), T.run_oracle(p, d), True)


@pytest.mark.parametrize("kernel", KERNELS)
def test_ragged_and_chunked(kernel):
    """n not a multiple of anything, several calls accumulating into one stats block,
    non-zero first_index, reads much shorter than the capacity."""
    d = synth.make_batch(7777, 120, paired=True, var_len=True, seed=25)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C3_full"])
    want = T.run_oracle(p, d, first_index=123456789)
    assert_same(p, run_hip_device(p, d, kernel, first_index=123456789, chunks=5), want, True)


@pytest.mark.parametrize("kernel", KERNELS)
def test_tiny_batches(kernel):
    """last tiles of every fill level: the bit collectors of the tiled kernel park after read 32 of a tile and
    are re-aligned for tiles of fewer than 64 (32) pairs"""
    for n in (1, 2, 3, 31, 32, 33, 34, 63, 64, 65, 95, 96, 97, 129):
        d = synth.make_batch(n, 150, paired=True, seed=26 + n)
        p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C2_adatrim_lowq"])
        assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), True)
        d = synth.make_batch(n, 150, paired=True, var_len=True, seed=126 + n)
        p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C3_full"])
        assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), True)
        d = synth.make_batch(n, 100, paired=False, seed=226 + n)
        p = abi.default_params(paired=False, max_read_len=100, **{k: (v[:2] if k == "hard_trim" else v)
                                                                  for k, v in PE_CASES["C3_full"].items() if k != "adapters2"})
        assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), False)


def test_empty_batch(snk_lib):
    p = abi.default_params()
    ctx = snk_lib.snk_create(C.byref(p), 0)
    assert ctx
    b = abi.Batch()
    b.n, b.pitch = 0, 160
    assert snk_lib.snk_filter_batch(ctx, C.byref(b), None, None) == 0
    s, mx = T.new_stats(p)
    err = abi.Error()
    assert snk_lib.snk_stats_fetch(ctx, s.ctypes.data, mx.ctypes.data, C.byref(err), None) == 0
    assert not s.any() and not mx.any() and err.code == 0
    snk_lib.snk_destroy(ctx)


@pytest.mark.parametrize("kernel", KERNELS)
def test_rmdup_flags(kernel):
    d = synth.make_batch(5000, 150, paired=True, seed=27)
    dup = (np.random.default_rng(1).random(5000) < 0.05).astype(np.uint8)
    p = abi.default_params(paired=True, max_read_len=150, rmdup=1, **PE_CASES["C2_adatrim_lowq"])
    assert_same(p, run_hip_device(p, d, kernel, dup=dup), T.run_oracle(p, d, dup=dup), True)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("paired,rmdup", [(True, 1), (True, 0), (False, 1)])
def test_host_verdict_bits(kernel, paired, rmdup):
    """snk_batch.dup as a flag byte: bit 0 duplicate (only with params.rmdup), bit 1 tile, bit 2 fov -- the
    first three steps of the cascade (src/sequence.cpp:200-231)."""
    n = 6000
    d = synth.make_batch(n, 150, paired=paired, seed=28)
    flags = np.random.default_rng(2).choice(np.arange(8, dtype=np.uint8), n, p=[.72, .04, .04, .04, .04, .04, .04, .04])
    kw = PE_CASES["C2_adatrim_lowq"] if paired else se_kwargs(PE_CASES["C2_adatrim_lowq"])
    p = abi.default_params(paired=paired, max_read_len=150, rmdup=rmdup, **kw)
    o = T.run_oracle(p, d, dup=flags)
    assert_same(p, run_hip_device(p, d, kernel, dup=flags), o, paired)
    assert o["sum"][1] > 0 and o["sum"][2] > 0 and (o["sum"][0] > 0) == bool(rmdup)      # SNK_FS_TILE, SNK_FS_FOV, SNK_FS_DUP


@pytest.mark.parametrize("kernel", KERNELS)
def test_error_reporting(kernel):
    """unrecognized base / quality out of range: same first-offender as the oracle
    (the reference exit(1)s: src/read_filter.cpp:283)."""
    d = synth.make_batch(3000, 150, paired=True, seed=28)
    d["seq"][1][1234, 77] = ord("X")
    d["seq"][0][2000, 3] = ord("#")
    p = abi.default_params(paired=True, max_read_len=150)
    got = run_hip_device(p, d, kernel)
    want = T.run_oracle(p, d)
    assert want["err"] == (abi.E_BAD_BASE, 1, 1234)
    assert got["err"] == want["err"]
    d = synth.make_batch(3000, 150, paired=True, seed=29)
    d["qual"][0][17, 5] = 33 + 60
    got = run_hip_device(p, d, kernel)
    assert got["err"] == (abi.E_QUAL_RANGE, 0, 17)
    # below the Phred offset (the tiled kernel's underflow row), in each 64-position strip and in both mates
    for mate, row, pos, ch in ((0, 40, 2, 32), (1, 41, 70, 10), (0, 2999, 149, 0), (1, 7, 64, 31)):
        d = synth.make_batch(3000, 150, paired=True, seed=29)
        d["qual"][mate][row, pos] = ch
        got = run_hip_device(p, d, kernel)
        assert got["err"] == (abi.E_QUAL_RANGE, mate, row), (mate, row, pos, ch, got["err"])
    # the first offender wins, whatever the kind of offence
    d = synth.make_batch(3000, 150, paired=True, seed=29)
    d["qual"][1][900, 100] = 5
    d["qual"][0][1500, 3] = 33 + 70
    got = run_hip_device(p, d, kernel)
    assert got["err"] == (abi.E_QUAL_RANGE, 1, 900)


@pytest.mark.parametrize("kernel", KERNELS)
def test_lowercase_and_N_reads(kernel):
    """case-insensitive counting but case-sensitive adapter/polyX compares
    (src/read_filter.cpp:261,270-281,728)."""
    d = synth.make_batch(4000, 150, paired=True, seed=30)
    rng = np.random.default_rng(2)
    rows = rng.choice(4000, 400, replace=False)
    for m in range(2):
        blk = d["seq"][m][rows, :150]
        low = rng.random(blk.shape) < 0.3
        d["seq"][m][rows, :150] = np.where(low, blk | 0x20, blk)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C3_full"])
    assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), True)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("polyx", [8, 20, 50])
def test_polyx_runs_of_every_kind(kernel, polyx):
    """contig_base is the longest run of identical case-SENSITIVE characters, whatever they are
    (src/read_filter.cpp:255-269): runs of N, of one lower-case letter, mixed-case runs (two runs), runs
    across the 64-position strip borders and at the read ends."""
    L, n = 150, 2048
    d = synth.make_batch(n, L, paired=True, seed=33)
    rng = np.random.default_rng(5)
    for m in range(2):
        for i in range(0, n, 2):
            run = int(rng.integers(polyx - 2, polyx + 3))
            at = int(rng.choice([0, L - run, 64 - run // 2, 128 - run // 2, int(rng.integers(0, L - run))]))
            kind = i // 2 % 6
            ch = [b"N", b"a", b"G", b"t", b"C", b"n"][kind]
            d["seq"][m][i, at:at + run] = np.frombuffer(ch * run, dtype=np.uint8)
            if kind == 2 and run > 4:                      # mixed case: G..g..G is three runs
                d["seq"][m][i, at + run // 2] = ord("g")
    kw = dict(PE_CASES["C2_adatrim_lowq"], polyX_num=polyx, n_ratio=0.9)
    p = abi.default_params(paired=True, max_read_len=L, **kw)
    assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), True)


@pytest.mark.parametrize("case,var_len", [("C2_adatrim_lowq", 0), ("C3_full", 1)])
@pytest.mark.parametrize("wgs,every", [(2, 3), (5, 1), (256, 2)])
def test_multi_flush_launches(case, var_len, wgs, every):
    """The read-modify-write branch of the tiled kernel's histogram flush (a launch's second and later flushes) and the zero
    invariant of the per-workgroup partials across launches need > 16.5 M pairs per launch at their natural settings
    (ADVICE r3); the library's test hooks force them on 150 k pairs: few workgroups -> many iterations, a flush every
    `every` iterations.  Three launches on one stream slot against the oracle run three times."""
    import subprocess
    import sys
    env = dict(os.environ, SNK_TEST_MAX_WGS=str(wgs), SNK_TEST_FLUSH_EVERY=str(every),
               PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), T.ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "flush_hook_child.py"), case, str(var_len)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "multi-flush OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_full_size_properties():
    """BASELINE configs[1] shape (PE150, adapter-trim + lowQual) at a size the oracle
    cannot follow in seconds: size-independent invariants instead.  Stats are a pure
    sum over reads, so 4 chunks accumulated == the whole, the generic and the auto
    kernel agree on a checksum of the records, and gs totals equal histogram totals."""
    n = 1 << 20
    d = synth.make_batch(n, 150, paired=True, seed=31)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C2_adatrim_lowq"])
    a = run_hip_device(p, d, 0, chunks=1)
    b = run_hip_device(p, d, 0, chunks=4)
    c = run_hip_device(p, d, 1, chunks=3)
    for other in (b, c):
        assert np.array_equal(a["sum"], other["sum"]) and np.array_equal(a["max"], other["max"])
        for m in range(2):
            assert np.array_equal(a["rec"][m], other["rec"][m])
    lcap, nq = 150, 43
    for k in range(4):
        f = a["sum"][abi.file_off(lcap, nq, k):][:abi.file_block_u64(lcap, nq)]
        bs = f[abi.bs_off(lcap, nq):abi.qs_off(lcap, nq)].reshape(lcap, 5)
        qs = f[abi.qs_off(lcap, nq):abi.ts_off(lcap, nq)].reshape(lcap, nq)
        assert f[abi.GS_BASES] == bs.sum() == qs.sum()
        assert f[abi.GS_READS] == bs[0].sum()
        assert f[abi.GS_Q20] == qs[:, 20:].sum() and f[abi.GS_Q30] == qs[:, 30:].sum()
    kept = int((a["rec"][0]["reason"] == 0).sum())
    raw1 = a["sum"][abi.file_off(lcap, nq, 0):]
    clean1 = a["sum"][abi.file_off(lcap, nq, 2):]
    assert raw1[abi.GS_READS] == n and clean1[abi.GS_READS] == kept
    assert int(a["sum"][:abi.SNK_FS_N][[abi.FS_SHORT, abi.FS_NRATE, abi.FS_LOWQUAL]].sum()) == n - kept
    # and a slice of it against the oracle
    sub = {"n": 20000, "L": 150, "pitch": d["pitch"], "paired": True,
           "seq": [x[:20000] for x in d["seq"]], "qual": [x[:20000] for x in d["qual"]], "len": [None, None]}
    o = T.run_oracle(p, sub)
    for m in range(2):
        assert np.array_equal(a["rec"][m][:20000], o["rec"][m])


def test_many_streams_many_stats_blocks():
    """Launches from 12 streams (more than the context's 8 per-stream copies of the trimming-position
    counters) into 4 alternating stats blocks: every block must equal the oracle's for its own chunks."""
    import torch
    from soapnuke_amd.filter import FilterContext, records_to_numpy
    n, chunks, nblk = 60000, 24, 4
    d = synth.make_batch(n, 150, paired=True, var_len=True, seed=31)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C3_full"])
    ctx = FilterContext(p, device=0)
    dev = ctx.upload(d)
    rec = ctx.alloc_records(n)
    blocks = [(torch.zeros_like(ctx.sum), torch.zeros_like(ctx.max)) for _ in range(nblk)]
    streams = [torch.cuda.Stream() for _ in range(12)]
    torch.cuda.synchronize()
    edges = np.linspace(0, n, chunks + 1).astype(int)
    want = [T.new_stats(p) for _ in range(nblk)]
    for k, (a, z) in enumerate(zip(edges[:-1], edges[1:])):
        sub = {"n": int(z - a), "L": 150, "pitch": dev["pitch"], "seq": [x[a:z] for x in dev["seq"]],
               "qual": [x[a:z] for x in dev["qual"]], "len": [x[a:z] for x in dev["len"]]}
        ctx.bind(*blocks[k % nblk])
        with torch.cuda.stream(streams[k % len(streams)]):
            ctx.filter_batch(ctx.make_batch(sub, first_index=int(a)), [r[a:z] for r in rec])
        hs = {"n": int(z - a), "L": 150, "pitch": d["pitch"], "paired": True, "seq": [x[a:z] for x in d["seq"]],
              "qual": [x[a:z] for x in d["qual"]], "len": [x[a:z] for x in d["len"]]}
        T.run_oracle(p, hs, first_index=int(a), stats=want[k % nblk])
    torch.cuda.synchronize()
    for k in range(nblk):
        ctx.bind(*blocks[k])
        s, mx, err = ctx.fetch()
        assert err[0] == 0
        assert np.array_equal(s, want[k][0]), (k, T.describe_stats_diff(p, s, want[k][0]))
        assert np.array_equal(mx, want[k][1])


@pytest.mark.parametrize("name", ["C2_adatrim_lowq", "C3_full"])
@pytest.mark.parametrize("pitch,var_len", [(152, False), (156, True), (152, True)])
def test_pitch_not_multiple_of_16(name, pitch, var_len):
    """pitch % 16 != 0 (and, through a row offset, planes that are not 16-byte aligned): the tiled kernel
    takes its register path instead of the LDS-DMA staging."""
    import torch
    from soapnuke_amd.filter import FilterContext, records_to_numpy
    d = synth.make_batch(30000, 150, paired=True, var_len=var_len, seed=41, pitch=pitch)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES[name])
    for first in (0, 1):                                   # row offset 1: plane pointers are 8 (mod 16)
        sub_h = {"n": d["n"] - first, "L": 150, "pitch": pitch, "paired": True, "seq": [x[first:] for x in d["seq"]],
                 "qual": [x[first:] for x in d["qual"]], "len": [None if x is None else x[first:] for x in d["len"]]}
        ctx = FilterContext(p, device=0)
        dev = ctx.upload(d)
        sub = {"n": d["n"] - first, "L": 150, "pitch": pitch, "seq": [x[first:] for x in dev["seq"]],
               "qual": [x[first:] for x in dev["qual"]], "len": [None if x is None else x[first:] for x in dev["len"]]}
        rec = ctx.alloc_records(sub["n"])
        ctx.filter_batch(ctx.make_batch(sub), rec, kernel=2)
        s, mx, err = ctx.fetch()
        o = T.run_oracle(p, sub_h)
        got = dict(rec=[records_to_numpy(r) for r in rec], sum=s, max=mx, err=err)
        assert_same(p, got, o, True)


@pytest.mark.parametrize("name", ["C3_full", "hard_lq_trim", "defaults"])
def test_generic_anchor(name):
    """kernel = 3: the generic kernel alone, histograms by its own global atomics (kernel = 1 leaves them to the LDS histogram
    kernel behind it); both against the oracle"""
    d = synth.make_batch(8000, 150, paired=True, var_len=True, seed=23)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES[name])
    want = T.run_oracle(p, d)
    assert_same(p, run_hip_device(p, d, 3), want, True)
    assert_same(p, run_hip_device(p, d, 1, chunks=3), want, True)


@pytest.mark.parametrize("L", [51, 64, 65, 96, 97, 128, 129, 255, 256, 257, 1000])
def test_capacity_boundaries(L):
    """read-length capacities around the plane-word / strip / tiled-kernel limits (tiled up to 256, the long-read path beyond)"""
    n = 6000 if L <= 257 else 1500
    d = synth.make_batch(n, L, paired=True, var_len=True, seed=50 + L)
    p = abi.default_params(paired=True, max_read_len=L, **PE_CASES["C3_full"])
    assert_same(p, run_hip_device(p, d, 0), T.run_oracle(p, d), True)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", sorted(CONTAM_CASES))
@pytest.mark.parametrize("paired", [True, False])
def test_contaminant_screening(name, paired, kernel):
    """contam1/contam2 (+ctMatchR) and global_contams, SURVEY 8f N3: verdicts from the sequential matchers on the
    device -- inline in the generic kernel, or as their own pass in front of the tiled kernel"""
    kw = CONTAM_CASES[name]
    d = synth.make_batch(12000, 150, paired=paired, var_len=(name != "single"), seed=91)
    plant_contams(d, kw)
    p = abi.default_params(paired=paired, max_read_len=150, **contam_kwargs(kw, paired))
    assert_same(p, run_hip_device(p, d, kernel, chunks=3), T.run_oracle(p, d), paired)


def test_contaminant_list_size_mismatch_is_refused():
    from soapnuke_amd.filter import FilterContext, FilterError
    p = abi.default_params(paired=True, max_read_len=150, contam1="ACGTACGTACGT,GGGGGGGGGGGG", ct_match_r="0.5")
    with pytest.raises(FilterError):
        FilterContext(p, device=0)


# ---- BASELINE configs at their own sizes

@pytest.mark.parametrize("kernel", KERNELS)
def test_config0_se_100k_x150_defaults(kernel):
    """BASELINE configs[0]: SE 100k x 150 bp, `filter` default parameters -- the whole of it against the oracle."""
    d = synth.make_batch(100_000, 150, paired=False, seed=synth.SEED)
    p = abi.default_params(paired=False, max_read_len=150)
    assert_same(p, run_hip_device(p, d, kernel), T.run_oracle(p, d), False)


def test_config1_pe_10m_x150_full_size():
    """BASELINE configs[1] at its full size, laid out as bench.py lays it out: PE 10 M x 150, `-J -l 10 -q 0.1`, one
    launch over 10 M device-resident pairs (1 M unique pairs x 10 distinct HBM copies).  The 1 M unique pairs are checked
    against the oracle in full; the 10 M launch through linearity: every counter is 10 x the 1 M block, every replica's
    records equal the first one's, and the `last read seen` keys name pair 10 M - 1."""
    import torch
    from soapnuke_amd.filter import FilterContext, records_to_numpy
    n1, reps, L = 1_000_000, 10, 150
    d = synth.make_batch(n1, L, paired=True, seed=synth.SEED)
    p = abi.default_params(paired=True, max_read_len=L, **PE_CASES["C2_adatrim_lowq"])
    one = run_hip_device(p, d, 2)
    assert_same(p, one, T.run_oracle(p, d), True)
    ctx = FilterContext(p, device=0)
    dev = ctx.upload(d)
    dev["seq"] = [x.repeat(reps, 1) for x in dev["seq"]]
    dev["qual"] = [x.repeat(reps, 1) for x in dev["qual"]]
    dev["n"] = n1 * reps
    rec = ctx.alloc_records(n1 * reps)
    ctx.filter_batch(ctx.make_batch(dev), rec, kernel=2)
    s, mx, err = ctx.fetch()
    assert err[0] == 0
    assert np.array_equal(s, one["sum"] * np.uint64(reps)), T.describe_stats_diff(p, s, one["sum"] * np.uint64(reps))
    for m in range(2):
        r = rec[m].view(n1 * reps, 16).reshape(reps, n1, 16)
        assert bool((r == r[0:1]).all())
        assert np.array_equal(records_to_numpy(rec[m][:n1]), one["rec"][m])
    # max block: (index + 1) << 16 | length of the last read of each file (raw files: the last pair)
    assert int(mx[0]) >> 16 == n1 * reps and int(mx[0]) & 0xFFFF == L
    kept_last = int(np.nonzero(one["rec"][0]["reason"] == 0)[0][-1]) + (reps - 1) * n1
    assert int(mx[2]) >> 16 == kept_last + 1
    ctx.close()
    del dev, rec
    torch.cuda.empty_cache()
