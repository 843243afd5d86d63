"""Multi-GPU layout of the path: read pairs shard embarrassingly -- rank g
takes the g-th contiguous range of pairs, runs the hot path on it, and the only collective
is one sum all-reduce of the stats block plus one max all-reduce of the tiny max block
(RCCL over xGMI on GPUs, `nccl` backend; gloo in the CPU tests)."""


def shard_bounds(n, rank, world):
    """Contiguous, order-preserving split: concatenating the shards' outputs in rank order
    reproduces input order (the reference's clean files are ordered, ChangeLog:151)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_stats(sum_t, max_t):
    """In place.  sum_t / max_t: int64 tensors viewing the uint64 blocks (two's-complement
    addition is the same operation; the max block's keys stay below 2**63)."""
    import torch.distributed as dist
    dist.all_reduce(sum_t, op=dist.ReduceOp.SUM)
    dist.all_reduce(max_t, op=dist.ReduceOp.MAX)
