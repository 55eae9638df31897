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


def _exchange(send, world, group, dst):
    """One collective over equal-sized blocks: every rank's ``send`` stacked as ``(world * rows, width)`` - on every
    rank (``dst`` None: ``all_gather_into_tensor``) or on rank ``dst`` alone (``dist.gather``: each peer's block goes
    to the root over its own xGMI link and nobody else holds the 205 MB of config 4; the others get None).
    ``dst`` is a global rank (a member of ``group``), the convention of ``torch.distributed.gather``; the blocks are
    stacked in the group's rank order."""
    import torch
    import torch.distributed as dist

    if dst is None:
        buf = torch.empty((world * send.shape[0], send.shape[1]), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(buf, send, group=group)
        return buf
    if dist.get_rank() == dst:                       # dst is a GLOBAL rank, as torch.distributed.gather takes it
        buf = torch.empty((world * send.shape[0], send.shape[1]), dtype=send.dtype, device=send.device)
        dist.gather(send, list(buf.view(world, send.shape[0], send.shape[1]).unbind(0)), dst=dst, group=group)
        return buf
    dist.gather(send, None, dst=dst, group=group)
    return None


def gather_rows(local, n_total, group=None, force=False, dst=None):
    """Gather row shards (unequal row counts allowed) into ``(n_total, F)`` on every rank, or - ``dst`` given - on
    that rank alone (the others return None).

    ``local`` is this rank's ``(P_r, F)`` tensor, where ``P_r`` follows ``shard_bounds``.
    Equal shards go through one ``all_gather_into_tensor``; ragged ones are padded to the
    largest shard first.  A group of one rank returns ``local`` itself unless ``force`` is set:
    then the rows go through the collective anyway (the N > 1 code path on one GPU).
    """
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
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
    buf = _exchange(local, world, group, dst)
    if buf is None:
        return None
    buf = buf.to(device)
    if all(c == biggest for c in counts):
        return buf
    return torch.cat([buf[r * biggest: r * biggest + counts[r]] for r in range(world)], dim=0)


def gather_scalars(values, device="cpu", group=None):
    """Every rank's small vector of float64 statistics (kernel time, gather time ...) as a ``(world, n)`` NumPy
    array on every rank; ``(1, n)`` without a process group.  One flat ``all_gather_into_tensor`` (the flat form is
    the one gloo and RCCL both take)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    mine = torch.as_tensor(np.asarray(values, dtype=np.float64).ravel())
    if not dist.is_initialized():
        return mine.numpy().reshape(1, -1).copy()
    world = dist.get_world_size(group)
    mine = mine.to(device)
    every = torch.empty(world * mine.numel(), dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(every, mine, group=group)
    return every.cpu().numpy().reshape(world, mine.numel())


def shard_segments(segments, world_size, rank):
    """Cost-balanced cut of a mixed work list (BASELINE config 5).

    The cost of a slice is proportional to its row count times ``n_points`` (times the reflecting
    fraction, which the synthetic and PyIRI ensembles spread evenly over rows), so giving every rank
    the ``shard_bounds`` block of EVERY segment gives every rank the same mix of cheap and
    expensive rows and lets all GPUs finish together - no rank ends up with only the
    ``n_points = 20000`` rows.

    ``segments``: sequence of ``(prof_begin, prof_end, mode, n_points)`` over the global rows.
    Returns ``(rows, local_segments)``: ``rows`` is the int64 array of global row numbers this rank
    evaluates, in local order, and ``local_segments`` the same slices re-based onto that local
    stack (ready for ``library.vertical_forward_operator_mixed``).  Empty cuts are dropped.
    """
    import numpy as np

    rows, local, off = [], [], 0
    for (p0, p1, mode, n_points) in segments:
        lo, hi = shard_bounds(int(p1) - int(p0), world_size, rank)
        if hi > lo:
            rows.append(np.arange(int(p0) + lo, int(p0) + hi, dtype=np.int64))
            local.append((off, off + hi - lo, mode, int(n_points)))
            off += hi - lo
    return (np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)), local


def gather_mixed(local, segments, n_total, group=None, force=False, dst=None):
    """Reassemble the ``(n_total, F)`` result of a mixed work list cut by ``shard_segments``.

    ``local`` holds this rank's rows in ``shard_segments`` order.  One padded collective, then each
    rank's rows go back to their global positions; rows no segment covers stay NaN.  ``force``: a
    group of one rank goes through the collective too (see ``gather_rows``); ``dst``: only that rank
    assembles the result, the others return None.
    """
    import numpy as np
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    cuts = [shard_segments(segments, world, r)[0] for r in range(world)]
    if local.shape[0] != cuts[rank].size:
        raise ValueError("local rows do not match shard_segments for this rank")
    if world == 1 and not (force and dist.is_initialized()):
        full = torch.full((int(n_total), local.shape[1]), float("nan"), dtype=local.dtype, device=local.device)
        full[torch.as_tensor(cuts[0], device=local.device)] = local
        return full
    biggest = max(max(c.size for c in cuts), 1)
    send = torch.full((biggest, local.shape[1]), float("nan"), dtype=local.dtype, device=local.device)
    send[: local.shape[0]] = local
    device = local.device
    if send.is_cuda and dist.get_backend(group) == "gloo":
        send = send.cpu()
    buf = _exchange(send.contiguous(), world, group, dst)
    if buf is None:
        return None
    buf = buf.to(device)
    full = torch.full((int(n_total), local.shape[1]), float("nan"), dtype=local.dtype, device=device)
    for r, rows in enumerate(cuts):
        if rows.size:
            full[torch.as_tensor(rows, device=device)] = buf[r * biggest: r * biggest + rows.size]
    return full
