"""Data-parallel sharding of profile batches: one process per GPU, no data-path collective.

Pairs are independent, so the profile axis is cut into contiguous blocks, one per rank;
the only communication is the gather of the (P_shard, F) result rows (RCCL over xGMI with
backend "nccl", gloo in the CPU tests).  SURVEY.md section 8e.
"""

from __future__ import annotations

import os


def env_rank():
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_bounds(n_items, world_size, rank):
    """Contiguous block [lo, hi) of rank ``rank``; the first ``n % world`` ranks get one more."""
    if not 0 <= rank < world_size:
        raise ValueError("rank outside [0, world_size)")
    q, r = divmod(int(n_items), int(world_size))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_counts(n_items, world_size):
    return [shard_bounds(n_items, world_size, r)[1] - shard_bounds(n_items, world_size, r)[0]
            for r in range(world_size)]


def gather_rows(local, n_total, group=None):
    """All-gather row shards (unequal row counts allowed) into ``(n_total, F)`` on every rank.

    ``local`` is this rank's ``(P_r, F)`` tensor, where ``P_r`` follows ``shard_bounds``.
    Equal shards go through one ``all_gather_into_tensor``; ragged ones are padded to the
    largest shard first.
    """
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    counts = shard_counts(n_total, world)
    width = local.shape[1]
    biggest = max(counts)
    if local.shape[0] != counts[dist.get_rank(group)]:
        raise ValueError("local shard does not match shard_bounds for this rank")
    if local.shape[0] < biggest:
        pad = torch.full((biggest - local.shape[0], width), float("nan"), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    device = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()          # rehearsal on one GPU box: gloo gathers host tensors only
    buf = torch.empty((world * biggest, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, local, group=group)
    buf = buf.to(device)
    if all(c == biggest for c in counts):
        return buf
    return torch.cat([buf[r * biggest: r * biggest + counts[r]] for r in range(world)], dim=0)
