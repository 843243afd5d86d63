"""The N>1 path on CPU: world_size 2, gloo.  Each rank runs its contiguous shard (the oracle
stands in for the GPU kernel -- there is no GPU here and no CPU product path), then the
same all-reduce the GPU ranks issue over RCCL; the reduced block must equal the block of
the unsharded run bit for bit, including the "last read seen" max keys."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import snk_testlib as T
from cases import PE_CASES
from soapnuke_amd import abi, synth
from soapnuke_amd.shard import allreduce_stats, shard_bounds


def _worker(rank, world, port, n, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = synth.make_batch(n, 150, paired=True, var_len=True, seed=77)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C3_full"])
    lo, hi = shard_bounds(n, rank, world)
    sub = dict(n=hi - lo, L=150, pitch=d["pitch"], paired=True, seq=[x[lo:hi] for x in d["seq"]],
               qual=[x[lo:hi] for x in d["qual"]], len=[x[lo:hi] for x in d["len"]])
    o = T.run_oracle(p, sub, first_index=lo)
    s = torch.from_numpy(o["sum"].view(np.int64).copy())
    mx = torch.from_numpy(o["max"].view(np.int64).copy())
    allreduce_stats(s, mx)
    rec = [torch.from_numpy(o["rec"][m].view(np.uint8).reshape(-1, 16).copy()) for m in range(2)]
    if rank == 0:
        np.savez(os.path.join(tmp, "reduced.npz"), sum=s.numpy().view(np.uint64), max=mx.numpy().view(np.uint64))
    np.save(os.path.join(tmp, f"rec{rank}.npy"), np.stack([r.numpy() for r in rec]))
    dist.destroy_process_group()


def test_shard_bounds_cover_in_order():
    for n in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_two_rank_allreduce_equals_unsharded(tmp_path):
    n, world, port = 4001, 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    d = synth.make_batch(n, 150, paired=True, var_len=True, seed=77)
    p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C3_full"])
    whole = T.run_oracle(p, d)
    z = np.load(tmp_path / "reduced.npz")
    assert np.array_equal(z["sum"], whole["sum"]), T.describe_stats_diff(p, z["sum"], whole["sum"])
    assert np.array_equal(z["max"], whole["max"])
    rec = np.concatenate([np.load(tmp_path / f"rec{r}.npy") for r in range(world)], axis=1)
    for m in range(2):
        assert np.array_equal(rec[m].reshape(-1).view(abi.record_dtype()), whole["rec"][m])
