"""GPU parity of the rmdup pre-pass through the C ABI (include/snk_rmdup.h): the hash kernel against
the oracle and the reference-made golden vectors, the marking kernels against the oracle's markDup --
all bit-exact -- and the size-independent property at BASELINE scale (a replicated batch: every
replica of a pair is a duplicate of the first)."""
import os

import numpy as np
import pytest

import snk_testlib as T
from soapnuke_amd import abi, synth

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rmdup.npz"))


def _ctx(paired, L):
    from soapnuke_amd.filter import FilterContext
    return FilterContext(abi.default_params(paired=paired, max_read_len=L, rmdup=1), device=0)


def _gpu_hash(d, paired, L):
    ctx = _ctx(paired, L)
    dev = ctx.upload(d)
    h = ctx.hash_batch(ctx.make_batch(dev))
    return h.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("name,paired,L,var", [("pe", True, 150, True), ("se", False, 100, True), ("pe_fixed", True, 150, False)])
def test_hash_golden(name, paired, L, var):
    d = synth.make_batch(1500, L, paired=paired, var_len=var, seed=424242)
    assert np.array_equal(_gpu_hash(d, paired, L), GOLD[name + "_hash"])


def _rand_batch(n, L, pitch, var, paired, seed):
    """random ACGTN rows (garbage beyond the read end must not matter) with arbitrary lengths 0..L"""
    rng = np.random.default_rng(seed)
    d = {"n": n, "L": L, "pitch": pitch, "paired": paired, "seq": [], "qual": [], "len": []}
    for m in range(2 if paired else 1):
        d["seq"].append(np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=(n, pitch))])
        d["qual"].append(np.full((n, pitch), 70, dtype=np.uint8))
        if var:
            ln = rng.integers(0, L + 1, size=n).astype(np.uint16)
            ln[:min(n, 2 * L + 2)] = (np.arange(min(n, 2 * L + 2)) % (L + 1)).astype(np.uint16)   # every seam residue, empty reads
            d["len"].append(ln)
        else:
            d["len"].append(None)
    return d


@pytest.mark.parametrize("paired", [True, False])
@pytest.mark.parametrize("L,pitch,var", [(150, 160, False), (150, 160, True), (250, 256, True), (100, 100, True),
                                          (36, 36, True), (64, 64, False), (8, 16, True), (1000, 1008, True), (151, 152, True)])
def test_hash_vs_oracle(paired, L, pitch, var):
    n = 5000 if L <= 250 else 2100
    d = _rand_batch(n, L, pitch, var, paired, seed=1000 + L)
    assert np.array_equal(_gpu_hash(d, paired, L), T.oracle_hash_batch(d, paired))


def test_hash_odd_tile_counts():
    for n in (1, 63, 64, 65, 127, 4097):
        d = synth.make_batch(n, 150, paired=True, var_len=True, seed=n)
        assert np.array_equal(_gpu_hash(d, True, 150), T.oracle_hash_batch(d, True))


def _gpu_mark(h, index=None, total_n=None, sentinel_total=-1):
    import torch
    ctx = _ctx(True, 150)
    ht = torch.from_numpy(np.ascontiguousarray(h).view(np.int64)).cuda()
    it = None if index is None else torch.from_numpy(np.ascontiguousarray(index, dtype=np.uint32).view(np.int32)).cuda()
    return ctx.mark_dups(ht, it, total_n, sentinel_total).cpu().numpy()


@pytest.mark.parametrize("key", ["mark0", "mark1", "mark2", "mark3", "mark4", "mark_lone"])
def test_mark_golden(key):
    assert np.array_equal(_gpu_mark(GOLD[key + "_hash"]), GOLD[key + "_dup"])


def test_mark_vs_oracle_random():
    rng = np.random.default_rng(3)
    for n in (1, 2, 9, 10, 1000, 300000, 2_000_000):
        for mode in range(3):
            h = rng.integers(0, max(2, n // 2), n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
            if mode == 1:
                h[rng.integers(0, n, max(1, n // 10))] = np.uint64(0xFFFFFFFFFFFFFFFF)
            if mode == 2:
                h = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(3)
                h[n // 2] = np.uint64(0xFFFFFFFFFFFFFFFF)
            assert np.array_equal(_gpu_mark(h), T.oracle_markdup(h)), (n, mode)


def test_mark_with_explicit_indices():
    # the multi-GPU owner side: elements arrive in arbitrary order with their global indices
    rng = np.random.default_rng(8)
    n = 100000
    h = rng.integers(0, n // 4, n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    want = T.oracle_markdup(h)
    perm = rng.permutation(n)
    got = _gpu_mark(h[perm], index=perm.astype(np.uint32), total_n=n)
    assert np.array_equal(got, want[perm])
    # indices above 2^31 (uint32 compare)
    big = (perm.astype(np.uint64) + np.uint64(4_000_000_000)).astype(np.uint32)
    got = _gpu_mark(h[perm], index=big, total_n=4_294_000_000)
    assert np.array_equal(got, want[perm])


def test_too_many_reads_is_refused():
    import torch
    ctx = _ctx(True, 150)
    h = torch.zeros(4, dtype=torch.int64, device="cuda")
    with pytest.raises(Exception):
        ctx.mark_dups(h, None, 1 << 32)


def test_full_size_property_and_filter_integration():
    """BASELINE-scale property: 1 M unique pairs replicated 4x -> exactly the replicas are duplicates;
    and the flags, fed to the filter as snk_batch.dup, reproduce the oracle's records and counters."""
    import torch
    from soapnuke_amd.filter import FilterContext, records_to_numpy
    n0, reps = 1_000_000, 4
    d = synth.make_batch(n0, 150, paired=True, seed=5)
    p = abi.default_params(paired=True, max_read_len=150, rmdup=1, adapters1=[synth.ADAPTER1], adapters2=[synth.ADAPTER2])
    ctx = FilterContext(p, device=0)
    dev = ctx.upload(d)
    dev["seq"] = [x.repeat(reps, 1) for x in dev["seq"]]
    dev["qual"] = [x.repeat(reps, 1) for x in dev["qual"]]
    dev["n"] = n0 * reps
    b = ctx.make_batch(dev)
    h = ctx.hash_batch(b)
    hn = h.cpu().numpy().view(np.uint64)
    h0 = T.oracle_hash_batch(d, True)
    assert np.array_equal(hn, np.tile(h0, reps))
    dup = ctx.mark_dups(h)
    first = T.oracle_markdup(h0)               # (synthetic pairs can collide among themselves: keep that exact)
    want = np.concatenate([first] + [np.ones(n0, np.uint8)] * (reps - 1))
    assert np.array_equal(dup.cpu().numpy(), want)
    # the discard cascade consumes the flags (src/sequence.cpp:207): first 200k pairs of replica 2 vs oracle
    m = 200_000
    sub = {"n": m, "L": 150, "pitch": dev["pitch"], "seq": [x[n0:n0 + m] for x in dev["seq"]],
           "qual": [x[n0:n0 + m] for x in dev["qual"]], "len": [None, None]}
    rec = ctx.alloc_records(m)
    ctx.filter_batch(ctx.make_batch(sub, first_index=n0, dup=dup[n0:n0 + m]), rec)
    s, mx, err = ctx.fetch()
    hd = {"n": m, "L": 150, "pitch": d["pitch"], "paired": True, "seq": [x[:m] for x in d["seq"]],
          "qual": [x[:m] for x in d["qual"]], "len": [None, None]}
    o = T.run_oracle(p, hd, first_index=n0, dup=np.ones(m, np.uint8))
    assert err[0] == 0
    for k in range(2):
        assert np.array_equal(records_to_numpy(rec[k]), o["rec"][k])
    assert np.array_equal(s, o["sum"]), T.describe_stats_diff(p, s, o["sum"])


def test_exchange_path_single_rank_nccl():
    """The multi-GPU exchange (shard.rmdup_exchange_mark) with the real device kernels, world_size 1 over
    RCCL: all-to-all to self, explicit global indices, flags back."""
    import torch
    import torch.distributed as dist
    from soapnuke_amd.shard import rmdup_exchange_mark
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(12)
        n = 200000
        h = rng.integers(0, n // 3, n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        h[[5, 77777]] = np.uint64(0xFFFFFFFFFFFFFFFF)
        ctx = _ctx(True, 150)
        ht = torch.from_numpy(h.view(np.int64)).cuda()
        flags = rmdup_exchange_mark(ht, 0, n, lambda hh, ii, tot, sen: ctx.mark_dups(hh, ii, tot, sen),
                                    lambda hh, tot: ctx.bucket_count(hh, tot))
        assert np.array_equal(flags.cpu().numpy(), T.oracle_markdup(h))
    finally:
        dist.destroy_process_group()


def test_one_pass_table_vs_oracle():
    """snk_rmdup_stream_*: the table lives across batches, fed in file order from alternating streams; the
    verdicts equal the two-pass ones (the first occurrence stays, every later one is marked), the table
    grows from its resident hashes, and the reference's sentinel hash is reported, not mis-marked."""
    import ctypes as C
    import torch
    lib = abi.load_library()
    rng = np.random.default_rng(21)
    for n, expected, sentinel in ((200_000, 1024, False), (1_500_000, 400_000, False), (50_000, 50_000, True), (1, 0, False)):
        h = rng.integers(0, max(2, n // 3), n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
        h[h == np.uint64(0xFFFFFFFFFFFFFFFF)] = np.uint64(5)
        want = T.oracle_markdup(h)
        if sentinel:
            h[n // 2] = np.uint64(0xFFFFFFFFFFFFFFFF)
        t = lib.snk_rmdup_stream_create(None, expected)
        assert t
        streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
        got = np.zeros(n, np.uint8)
        outs, pos, k = [], 0, 0
        while pos < n:
            m = int(min(n - pos, rng.integers(1, 150_000)))
            st = streams[k % 3]
            with torch.cuda.stream(st):
                hb = torch.from_numpy(h[pos:pos + m].view(np.int64).copy()).cuda()
                db = torch.empty(m, dtype=torch.uint8, device="cuda")
            rc = lib.snk_rmdup_stream_mark_device(t, hb.data_ptr(), pos, m, db.data_ptr(), st.cuda_stream)
            assert rc == 0, lib.snk_last_error()
            outs.append((pos, m, db, hb))
            pos += m
            k += 1
        marked, seen = C.c_uint64(0), C.c_int32(0)
        assert lib.snk_rmdup_stream_stats(t, C.byref(marked), C.byref(seen)) == 0
        torch.cuda.synchronize()
        for p0, m, db, _ in outs:
            got[p0:p0 + m] = db.cpu().numpy()
        lib.snk_rmdup_stream_destroy(t)
        assert seen.value == (1 if sentinel else 0)
        if sentinel:
            keep = np.arange(n) != n // 2
            # the sentinel element itself is left unmarked and takes no part
            assert got[n // 2] == 0
            assert np.array_equal(got[keep], T.oracle_markdup(h[keep]))
        else:
            assert np.array_equal(got, want), n
            assert marked.value == int(want.sum())


def test_one_pass_table_refuses_too_many_reads():
    import torch
    lib = abi.load_library()
    t = lib.snk_rmdup_stream_create(None, 1024)
    hb = torch.zeros(8, dtype=torch.int64, device="cuda")
    db = torch.zeros(8, dtype=torch.uint8, device="cuda")
    assert lib.snk_rmdup_stream_mark_device(t, hb.data_ptr(), 4294967290, 8, db.data_ptr(), None) != 0
    assert b"2^32-1" in lib.snk_last_error()
    lib.snk_rmdup_stream_destroy(t)


@T.first_contact
def test_one_pass_table_single_end_shift():
    """snk_rmdup_stream_mark_se_device: out[i] = the true flag of read i - 1 inside full patches (across batch borders, from
    alternating streams, through a regrown scratch), the true flag of read i in the file's partial last patch
    (src/seprocess.cpp:1086,1112,1159)."""
    import ctypes as C
    import torch
    lib = abi.load_library()
    rng = np.random.default_rng(22)
    for n, ps, per_batch in ((100_000, 250, 4000), (60_100, 500, 7500), (999, 1000, 5000), (3000, 1000, 1000)):
        h = rng.integers(0, max(2, n // 3), n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
        true = T.oracle_markdup(h)
        full_end = n // ps * ps
        want = true.copy()
        want[1:full_end] = true[:max(full_end - 1, 0)]
        if full_end:
            want[0] = 0
        t = lib.snk_rmdup_stream_create(None, 1024)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs, pos, k = [], 0, 0
        while pos < n:
            m = min(n - pos, per_batch)
            st = streams[k % 2]
            with torch.cuda.stream(st):
                hb = torch.from_numpy(h[pos:pos + m].view(np.int64).copy()).cuda()
                db = torch.empty(m, dtype=torch.uint8, device="cuda")
            assert lib.snk_rmdup_stream_mark_se_device(t, hb.data_ptr(), pos, m, m // ps * ps, db.data_ptr(), st.cuda_stream) == 0, lib.snk_last_error()
            outs.append((pos, m, db, hb))
            pos += m
            k += 1
        marked, seen = C.c_uint64(0), C.c_int32(0)
        assert lib.snk_rmdup_stream_stats(t, C.byref(marked), C.byref(seen)) == 0
        torch.cuda.synchronize()
        got = np.concatenate([db.cpu().numpy() for _, _, db, _ in outs])
        lib.snk_rmdup_stream_destroy(t)
        assert np.array_equal(got, want), (n, ps, np.nonzero(got != want)[0][:8])
        assert marked.value == int(true.sum())


@T.first_contact
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_exchange_helpers_against_the_oracle(world):
    """include/snk_rmdup.h "multi-GPU exchange helpers": every rank's shard grouped by owner = hash % world
    (snk_rmdup_partition_device), the groups handed to their owners (here: by hand, what ncclSend / ncclRecv or the host wire do
    in the CLI, host/snk_wire.h), marked there with explicit global indices and the summed sentinel population
    (snk_rmdup_mark_device), the flags brought home (snk_rmdup_flags_home_device): the flags are rmdup::markDup's of the whole
    input (src/rmdup.cpp:14-149), sentinel quirk included."""
    import ctypes as C
    import torch
    from soapnuke_amd.filter import FilterContext
    lib = abi.load_library()
    rng = np.random.default_rng(40 + world)
    n = 50_000
    h = rng.integers(0, n // 2, n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(7)
    h[rng.choice(n, 5, replace=False)] = np.uint64(0xFFFFFFFFFFFFFFFF)              # the sentinel value, several times
    want = T.oracle_markdup(h)
    bounds = [n * r // world for r in range(world + 1)]
    shards, counts, sent = [], [], 0
    p = abi.default_params(paired=True, max_read_len=150, rmdup=1)
    ctx = FilterContext(p, device=0)
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        m = hi - lo
        hd = torch.from_numpy(h[lo:hi].view(np.int64).copy()).cuda()
        sh = torch.empty(max(m, 1), dtype=torch.int64, device="cuda")
        si = torch.empty(max(m, 1), dtype=torch.int32, device="cuda")
        slot = torch.empty(max(m, 1), dtype=torch.int32, device="cuda")
        cnt = (C.c_uint64 * world)()
        assert lib.snk_rmdup_partition_device(ctx.ctx, hd.data_ptr(), m, lo, world, sh.data_ptr(), si.data_ptr(), slot.data_ptr(), cnt, None) == 0, lib.snk_last_error()
        cnt = list(cnt)
        assert sum(cnt) == m
        own = (h[lo:hi] % np.uint64(world)).astype(np.int64)
        assert cnt == np.bincount(own, minlength=world).tolist()
        shn, sin_, sl = sh.cpu().numpy().view(np.uint64)[:m], si.cpu().numpy().view(np.uint32)[:m], slot.cpu().numpy().view(np.uint32)[:m]
        assert np.array_equal(shn[sl], h[lo:hi]) and np.array_equal(sin_[sl], (lo + np.arange(m)).astype(np.uint32))      # every element sits where its slot says
        assert np.array_equal(np.sort(sl), np.arange(m))
        assert np.array_equal((shn % np.uint64(world)).astype(np.int64), np.repeat(np.arange(world), cnt))               # grouped by owner, in owner order
        c1 = torch.zeros(1, dtype=torch.int64, device="cuda")
        assert lib.snk_rmdup_bucket_count_device(ctx.ctx, hd.data_ptr(), m, n, c1.data_ptr(), None) == 0
        sent += int(c1.item())
        shards.append((hd, sh, si, slot, m))
        counts.append(cnt)
    flags_home = []
    owner_flags = []
    for o in range(world):                                                        # what owner o receives, in source-rank order
        rh = torch.cat([shards[r][1][sum(counts[r][:o]):sum(counts[r][:o + 1])] for r in range(world)])
        ri = torch.cat([shards[r][2][sum(counts[r][:o]):sum(counts[r][:o + 1])] for r in range(world)])
        k = rh.numel()
        rf = torch.zeros(max(k, 1), dtype=torch.uint8, device="cuda")
        if k:
            assert lib.snk_rmdup_mark_device(ctx.ctx, rh.contiguous().data_ptr(), ri.contiguous().data_ptr(), k, n, sent, rf.data_ptr(), None) == 0, lib.snk_last_error()
        torch.cuda.synchronize()
        owner_flags.append(rf)
    for r in range(world):
        hd, sh, si, slot, m = shards[r]
        pieces = []
        for o in range(world):
            before = sum(counts[q][o] for q in range(r))
            pieces.append(owner_flags[o][before:before + counts[r][o]])
        back = torch.cat(pieces) if m else torch.zeros(1, dtype=torch.uint8, device="cuda")
        dup = torch.empty(max(m, 1), dtype=torch.uint8, device="cuda")
        assert lib.snk_rmdup_flags_home_device(ctx.ctx, back.contiguous().data_ptr(), slot.data_ptr(), m, dup.data_ptr(), None) == 0
        torch.cuda.synchronize()
        flags_home.append(dup.cpu().numpy()[:m])
    ctx.close()
    assert np.array_equal(np.concatenate(flags_home), want)
