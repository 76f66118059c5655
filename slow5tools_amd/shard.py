"""Read sharding across GPUs: records are independent, exactly how work_db splits them
(/root/reference/src/thread.c:76-90), so rank g of G takes the contiguous index range
[g*B/G, (g+1)*B/G) and no data-path collective exists.  The only cross-rank traffic is timing plumbing."""


def shard_range(n_total, rank, world):
    """contiguous, balanced, order-preserving split of range(n_total)"""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    return lo, hi


def max_over_ranks(seconds, device=None):
    """the bench contract: elapsed time = MAX over ranks (torch.distributed already initialised or world 1)"""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(count, device=None):
    """units processed by all ranks together (bench: `value` = the units all ranks processed / the max-over-ranks time)"""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(count)
    t = torch.tensor([int(count)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
