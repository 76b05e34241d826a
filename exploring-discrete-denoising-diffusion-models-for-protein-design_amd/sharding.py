"""Host-side sharding helpers of the multi-GPU paths (sharded reverse sampling,
``bench.py --gpus N``).  Pure torch / torch.distributed: nothing here needs the
HIP library, so the helpers can be imported and tested on a machine without it."""
from __future__ import annotations

import torch


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n_total items for `rank`; sizes differ by at most one."""
    per, rem = divmod(n_total, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def equal_sub_batches(n_local: int, max_batch: int):
    """Sizes of the fewest sub-batches of at most `max_batch` items that cover `n_local`, differing by at most
    one (1250 + 1250 rather than 1024 + 1024 + 452: no short, launch-bound tail)."""
    if n_local <= 0:
        return []
    nsub = -(-n_local // max(1, max_batch))
    return [hi - lo for lo, hi in (shard_range(n_local, i, nsub) for i in range(nsub))]


def gather_token_shards(local: torch.Tensor, group=None):
    """All ranks' [n_r, L] token shards, concatenated in rank order, on every rank
    (uint8 on the wire: K <= 256).  Ragged shards are padded to the largest.  Without
    an initialised process group the local shard is returned."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    sizes = [int(x.item()) for x in sizes]
    padded = torch.zeros((max(sizes), local.shape[1]), dtype=torch.uint8, device=local.device)
    padded[:local.shape[0]] = local.to(torch.uint8)
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)]).to(local.dtype)
