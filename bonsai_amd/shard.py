"""Multi-GPU plumbing of the classify path: reads shard, the db is replicated.

One process per GPU under torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  There is no exchange step inside classification: the only collectives are the
one-off broadcast of the khash arrays (+ whatever the caller needs replicated) and the per-batch
gather of the per-unit results.
"""
import torch


def shard_range(n_units, rank, world):
    """Contiguous unit range [lo, hi) of `rank`; units are reads or mate pairs, so mates never split."""
    lo = (n_units * rank) // world
    hi = (n_units * (rank + 1)) // world
    return lo, hi


def shard_sizes(n_units, world):
    return [shard_range(n_units, r, world)[1] - shard_range(n_units, r, world)[0] for r in range(world)]


def broadcast_table(dist, flags, keys, vals, src=0):
    """Replicate the khash arrays (bns.db payload) from `src` to every rank: one message per array."""
    for t in (flags, keys, vals):
        dist.broadcast(t, src=src)


def gather_results(dist, local, sizes, dst=0):
    """Gather per-unit results (ragged: sizes[r] units on rank r) onto `dst`; returns the concatenated
    tensor on dst, None elsewhere.  Ragged shards are padded to the largest shard for the collective."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    m = max(sizes)
    buf = local
    if local.numel() != m:
        buf = torch.zeros(m, dtype=local.dtype, device=local.device)
        buf[:local.numel()] = local
    if rank == dst:
        parts = [torch.empty(m, dtype=local.dtype, device=local.device) for _ in range(world)]
        dist.gather(buf, parts, dst=dst)
        return torch.cat([p[:s] for p, s in zip(parts, sizes)])
    dist.gather(buf, None, dst=dst)
    return None
