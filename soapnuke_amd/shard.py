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


def _host_staged(t):
    """gloo moves host memory only: device tensors are staged through the host there (the CPU tests,
    and the two-ranks-on-one-GPU test); RCCL (`nccl`) takes the device tensors as they are."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce(t, op):
    import torch.distributed as dist
    if _host_staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def _all_to_all(out, inp, out_splits=None, in_splits=None):
    import torch.distributed as dist
    if _host_staged(inp):
        ho, hi = out.cpu(), inp.cpu()
        dist.all_to_all_single(ho, hi, output_split_sizes=out_splits, input_split_sizes=in_splits)
        out.copy_(ho)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits)


def allreduce_stats(sum_t, max_t):
    """In place.  sum_t / max_t: int64 tensors viewing the uint64 blocks (two's-complement
    addition is the same operation; the max block's keys stay below 2**63)."""
    import torch.distributed as dist
    _all_reduce(sum_t, dist.ReduceOp.SUM)
    _all_reduce(max_t, dist.ReduceOp.MAX)


def rmdup_exchange_mark(hashes, first_index, total_n, mark_fn, bucket_count_fn):
    """Global first-occurrence duplicate marking over all ranks (the one exchange step of
    the rmdup row).  Every rank holds the hashes of its contiguous shard (int64 tensor, uint64 bit
    patterns; global index of element k = first_index + k).  Elements travel to owner = hash % world
    (all-to-all of (hash, global index) over RCCL), the owner marks "an equal hash with a smaller
    global index exists" (rmdup::markDup semantics, src/rmdup.cpp:70-123, need the GLOBAL order), and
    the flags travel back the same way.  mark_fn(hashes, index_int32, total_n, sentinel_bucket_total)
    -> uint8 flags; bucket_count_fn(hashes, total_n) -> 1-element int64 tensor (both device side:
    FilterContext.mark_dups / .bucket_count).  Returns the uint8 flags of this rank's shard."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    n = hashes.numel()
    dev = hashes.device
    # the sentinel quirk needs the population of one bucket over ALL ranks
    cnt = bucket_count_fn(hashes, total_n).clone()
    _all_reduce(cnt, dist.ReduceOp.SUM)
    sentinel_total = int(cnt.item())
    # owner = unsigned(hash) % world, computed on the int64 bit pattern
    owner = (((((hashes >> 1) & 0x7FFFFFFFFFFFFFFF) % world) * 2 + (hashes & 1)) % world) if world > 1 else torch.zeros_like(hashes)
    order = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    _all_to_all(recv_counts, send_counts)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    gidx = (torch.arange(n, device=dev, dtype=torch.int64) + int(first_index))
    send_h = hashes[order].contiguous()
    send_i = gidx[order].contiguous()
    recv_h = torch.empty(sum(rc), dtype=torch.int64, device=dev)
    recv_i = torch.empty(sum(rc), dtype=torch.int64, device=dev)
    _all_to_all(recv_h, send_h, rc, sc)
    _all_to_all(recv_i, send_i, rc, sc)
    flags_owned = mark_fn(recv_h, recv_i.to(torch.int32), total_n, sentinel_total)   # indices < 2**32 (reference limit)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    _all_to_all(back, flags_owned.contiguous(), sc, rc)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    out[order] = back
    return out
