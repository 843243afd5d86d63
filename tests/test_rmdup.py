"""rmdup pre-pass, CPU side: the oracle's restatement of libstdc++ _Hash_bytes and of
rmdup::markDup against (a) the known answers recorded in SURVEY 8(c), (b) golden vectors produced by
the reference itself (tests/golden/rmdup.npz, make_golden_rmdup.py), (c) the compiled reference where
oracle/_ref exists; plus the world_size-2 gloo run of the (hash, index) exchange."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import snk_testlib as T
from soapnuke_amd import abi, synth
from soapnuke_amd.shard import rmdup_exchange_mark, shard_bounds

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rmdup.npz"))


def test_hash_known_answers():
    # std::hash<std::string> on this toolchain (GCC 11.4), SURVEY 8(c)
    kat = {b"": 0x553e93901e462a6e, b"A": 0x600668de4345e18e, b"ACGT": 0x1d6bef14b4102038,
           b"ACGTACGTN": 0x17f2d5c96672025a}
    for s, h in kat.items():
        assert T.oracle_lib().snk_oracle_hash_bytes(s, len(s)) == h


@pytest.mark.parametrize("name,paired,L,var", [("pe", True, 150, True), ("se", False, 100, True), ("pe_fixed", True, 150, False)])
def test_hash_batch_golden(name, paired, L, var):
    d = synth.make_batch(1500, L, paired=paired, var_len=var, seed=424242)
    assert np.array_equal(T.oracle_hash_batch(d, paired), GOLD[name + "_hash"])


@pytest.mark.parametrize("key", ["mark0", "mark1", "mark2", "mark3", "mark4", "mark_lone"])
def test_markdup_golden(key):
    assert np.array_equal(T.oracle_markdup(GOLD[key + "_hash"]), GOLD[key + "_dup"])


def test_prime():
    lib = T.oracle_lib()
    assert [lib.snk_oracle_rmdup_prime(n) for n in (1, 5, 9, 10, 11, 12, 100, 1000, 1 << 20)] == \
        [1, 5, 9, 7, 7, 11, 97, 997, 1048573]
    hip = abi.load_library()        # host-only helper of the C ABI: no device needed
    for n in (1, 9, 10, 11, 100, 12345, 1 << 20, 200_000_000):
        assert hip.snk_rmdup_prime(n) == lib.snk_oracle_rmdup_prime(n)


@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref/libsnkref.so not built")
def test_oracle_vs_reference_random():
    rng = np.random.default_rng(5)
    o, r = T.oracle_lib(), T.ref_lib()
    for _ in range(5000):
        n = int(rng.integers(0, 700))
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert o.snk_oracle_hash_bytes(b, n) == r.snkref_hash(b, n)
    for n in (1, 2, 9, 10, 11, 1000, 100000):
        for mode in range(3):
            h = rng.integers(0, max(2, n // 2), n, dtype=np.uint64)
            if mode == 1:
                h[rng.integers(0, n, max(1, n // 10))] = np.uint64(0xFFFFFFFFFFFFFFFF)
            if mode == 2:
                h = rng.integers(0, 2**63, n, dtype=np.uint64)
                h[n // 2] = np.uint64(0xFFFFFFFFFFFFFFFF)
            assert np.array_equal(T.oracle_markdup(h), T.ref_markdup(h)), (n, mode)


# ---- N > 1: the (hash, global index) all-to-all, world_size 2 over gloo.  The oracle stands in for the
# device-side marking (there is no GPU here and no CPU product path).

def _cpu_mark(hashes, index32, total_n, sentinel_total):
    h = hashes.numpy().view(np.uint64)
    idx = index32.numpy().view(np.uint32)
    order = np.argsort(idx, kind="stable")
    dup = np.zeros(len(h), dtype=np.uint8)
    seen = set()
    for k in order:                      # "an equal hash with a smaller global index exists"
        v = int(h[k])
        if v == 0xFFFFFFFFFFFFFFFF:
            dup[k] = 1 if sentinel_total > 1 else 0
        elif v in seen:
            dup[k] = 1
        else:
            seen.add(v)
    return torch.from_numpy(dup)


def _cpu_bucket_count(hashes, total_n):
    h = hashes.numpy().view(np.uint64)
    prime = T.oracle_lib().snk_oracle_rmdup_prime(int(total_n))
    return torch.tensor([int(np.sum(h % np.uint64(prime) == np.uint64(0xFFFFFFFFFFFFFFFF) % np.uint64(prime)))], dtype=torch.int64)


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h = np.load(os.path.join(tmp, "hash.npy"))
    lo, hi = shard_bounds(len(h), rank, world)
    mine = torch.from_numpy(h[lo:hi].view(np.int64).copy())
    flags = rmdup_exchange_mark(mine, lo, len(h), _cpu_mark, _cpu_bucket_count)
    np.save(os.path.join(tmp, f"dup{rank}.npy"), flags.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("sentinel", [False, True])
def test_two_rank_exchange_equals_global_markdup(tmp_path, sentinel):
    rng = np.random.default_rng(17)
    n, world = 20011, 2
    h = rng.integers(0, n // 3, n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)      # both parities, many repeats
    if sentinel:
        h[[7, 9000, 15000]] = np.uint64(0xFFFFFFFFFFFFFFFF)
    np.save(tmp_path / "hash.npy", h)
    port = 31500 + os.getpid() % 2000 + (1 if sentinel else 0)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"dup{r}.npy") for r in range(world)])
    assert np.array_equal(got, T.oracle_markdup(h))
