"""The N > 1 path on CPU: world_size-2 gloo processes shard a batch by shard_bounds, evaluate
their rows (with the oracle standing in for the kernel - there is no GPU here) and gather;
rank 0 must hold the same (P, F) array as the unsharded evaluation."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyrayhf_amd import dist as pdist
from pyrayhf_amd import synth


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 100000, 12501):
        for world in (1, 2, 3, 8):
            edges = [pdist.shard_bounds(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == pdist.shard_counts(n, world)
    with pytest.raises(ValueError):
        pdist.shard_bounds(10, 2, 2)


def test_sharded_synthetic_rows_equal_full_batch():
    alt, den, bmag, bpsi = synth.chapman_profiles(11, 99)
    lo, hi = pdist.shard_bounds(11, 2, 1)
    alt2, den2, bmag2, bpsi2 = synth.chapman_profiles(11, 99, rows=slice(lo, hi))
    assert np.array_equal(den[lo:hi], den2) and np.array_equal(bmag[lo:hi], bmag2)
    assert np.array_equal(bpsi[lo:hi], bpsi2) and np.array_equal(alt, alt2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_prof, result_path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import vfo_numpy as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert pdist.env_rank()[:2] == (rank, world)
        freq = np.arange(1.0, 12.0, 0.5)
        lo, hi = pdist.shard_bounds(n_prof, world, rank)
        alt, den, bmag, bpsi = synth.chapman_profiles(n_prof, 321, rows=slice(lo, hi))
        local = torch.from_numpy(orc.virtual_heights_batch(freq, den, bmag, bpsi, alt, "X", 64))
        full = pdist.gather_rows(local, n_prof)
        if rank == 0:
            np.save(result_path, full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_prof", [6, 7])        # equal and ragged shards
def test_two_rank_gather_equals_unsharded(tmp_path, n_prof):
    from oracle import vfo_numpy as orc
    path = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), n_prof, path), nprocs=2, join=True)
    got = np.load(path)
    alt, den, bmag, bpsi = synth.chapman_profiles(n_prof, 321)
    want = orc.virtual_heights_batch(np.arange(1.0, 12.0, 0.5), den, bmag, bpsi, alt, "X", 64)
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)


MIXED = [(0, 5, "O", 32), (5, 8, "X", 64), (9, 13, "O", 64), (13, 14, "X", 128)]     # row 8 uncovered, 1-row slice


def test_shard_segments_partition_every_slice():
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            rows, local = pdist.shard_segments(MIXED, world, r)
            assert sum(hi - lo for lo, hi, _, _ in local) == rows.size
            assert [lo for lo, _, _, _ in local] == list(np.cumsum([0] + [hi - lo for lo, hi, _, _ in local])[:-1])
            seen.append(rows)
            for (p0, p1, mode, npts) in MIXED:          # every rank holds its block of every slice
                mine = rows[(rows >= p0) & (rows < p1)]
                lo, hi = pdist.shard_bounds(p1 - p0, world, r)
                assert np.array_equal(mine, np.arange(p0 + lo, p0 + hi))
        allrows = np.sort(np.concatenate(seen))
        assert np.array_equal(allrows, np.array([0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13]))


def _oracle_mixed(freq, den, bmag, bpsi, alt, segments):
    from oracle import vfo_numpy as orc
    out = np.full((den.shape[0], freq.size), np.nan)
    for (p0, p1, mode, npts) in segments:
        out[p0:p1] = orc.virtual_heights_batch(freq, den[p0:p1], bmag[p0:p1], bpsi[p0:p1], alt, mode, npts)
    return out


def _mixed_worker(rank, world, port, result_path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        freq = np.arange(1.0, 12.0, 0.5)
        alt, den, bmag, bpsi = synth.chapman_profiles(14, 555)
        rows, local_segs = pdist.shard_segments(MIXED, world, rank)
        local = torch.from_numpy(_oracle_mixed(freq, den[rows], bmag[rows], bpsi[rows], alt, local_segs))
        full = pdist.gather_mixed(local, MIXED, 14)
        if rank == 0:
            np.save(result_path, full.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_mixed_worklist_equals_unsharded(tmp_path):
    path = str(tmp_path / "mixed.npy")
    mp.spawn(_mixed_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    got = np.load(path)
    alt, den, bmag, bpsi = synth.chapman_profiles(14, 555)
    want = _oracle_mixed(np.arange(1.0, 12.0, 0.5), den, bmag, bpsi, alt, MIXED)
    assert np.isnan(got[8]).all()
    assert np.array_equal(got, want, equal_nan=True)


def test_gather_mixed_single_process():
    rows, local_segs = pdist.shard_segments(MIXED, 1, 0)
    local = torch.arange(rows.size * 3, dtype=torch.float64).reshape(rows.size, 3)
    full = pdist.gather_mixed(local, MIXED, 14)
    assert torch.isnan(full[8]).all() and torch.equal(full[torch.as_tensor(rows)], local)


def _stats_worker(rank, world, port, result_path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        every = pdist.gather_scalars([10.0 + rank, 0.5 * (rank + 1)])        # what bench.py sends: kernel ms, gather ms
        # the one-rank short cut and its `force`d collective on a group of two: both go through the collective here
        local = torch.full((3, 2), float(rank), dtype=torch.float64)
        forced = pdist.gather_rows(local, 6, force=True)
        if rank == 0:
            np.savez(result_path, every=every, forced=forced.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_statistics_gather(tmp_path):
    """bench.py's per-rank statistics (kernel_ms_per_rank, gather_ms) at N > 1: the flat all-gather both backends take."""
    path = str(tmp_path / "stats.npz")
    mp.spawn(_stats_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    got = np.load(path)
    assert np.array_equal(got["every"], np.array([[10.0, 0.5], [11.0, 1.0]]))
    assert np.array_equal(got["forced"], np.repeat([[0.0, 0.0], [1.0, 1.0]], 3, axis=0))
    assert np.array_equal(pdist.gather_scalars([1.0, 2.0, 3.0]), np.array([[1.0, 2.0, 3.0]]))      # no process group


def _one_rank_worker(rank, world, port, result_path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        local = torch.arange(12, dtype=torch.float64).reshape(4, 3)
        plain, forced = pdist.gather_rows(local, 4), pdist.gather_rows(local, 4, force=True)
        rows, segs = pdist.shard_segments(MIXED, 1, 0)
        mixed = torch.arange(rows.size * 2, dtype=torch.float64).reshape(rows.size, 2)
        full = pdist.gather_mixed(mixed, MIXED, 14, force=True)
        ok = ((plain is local) and (forced is not local) and torch.equal(forced, local)
              and torch.equal(full[torch.as_tensor(rows)], mixed) and bool(torch.isnan(full[8]).all()))
        np.save(result_path, np.array([ok]))
    finally:
        dist.destroy_process_group()


def test_one_rank_group_forced_through_the_collective(tmp_path):
    """`force=True`: a group of ONE rank still goes through all_gather_into_tensor (the N > 1 code path on one device)."""
    path = str(tmp_path / "one.npy")
    mp.spawn(_one_rank_worker, args=(1, _free_port(), path), nprocs=1, join=True)
    assert bool(np.load(path)[0])


# ---- N = 8 rehearsed on CPU: the cuts and collectives of BASELINE configs 4 and 5 at their real row counts -----------------
def _rows_value(rows, width):
    """A (len(rows), width) array that names its global row and column: what a rank 'computed' for those rows."""
    return rows[:, None].astype(np.float64) * 8.0 + np.arange(width, dtype=np.float64)[None, :]


def _n8_worker(rank, world, port, result_dir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import CONFIG5_SEGMENTS
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (a) ragged row shards: 100 003 rows of 4 columns (100 003 = 8 x 12 500 + 3: three ranks hold one row more)
        n = 100003
        lo, hi = pdist.shard_bounds(n, world, rank)
        local = torch.from_numpy(_rows_value(np.arange(lo, hi), 4))
        everywhere = pdist.gather_rows(local, n)
        at_root = pdist.gather_rows(local, n, dst=0)
        assert (at_root is None) == (rank != 0)
        # (b) config 5's segment table, cut for 8 ranks, 3 columns
        rows, local_segs = pdist.shard_segments(CONFIG5_SEGMENTS, world, rank)
        assert rows.size == 6250 and [b - a for a, b, _, _ in local_segs] == [2500, 1875, 1250, 625]
        mixed_local = torch.from_numpy(_rows_value(rows, 3))
        mixed_all = pdist.gather_mixed(mixed_local, CONFIG5_SEGMENTS, 50000)
        mixed_root = pdist.gather_mixed(mixed_local, CONFIG5_SEGMENTS, 50000, dst=0)
        assert (mixed_root is None) == (rank != 0)
        # every rank holds the whole array after the all-gather variants
        ok = bool(torch.equal(everywhere, torch.from_numpy(_rows_value(np.arange(n), 4)))
                  and torch.equal(mixed_all, torch.from_numpy(_rows_value(np.arange(50000), 3))))
        if rank == 0:
            ok = ok and bool(torch.equal(at_root, everywhere) and torch.equal(mixed_root, mixed_all))
        np.save(os.path.join(result_dir, f"ok{rank}.npy"), np.array([ok]))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_ragged_rows_and_the_config5_cut(tmp_path):
    """World size 8 (gloo, CPU): `gather_rows` on 100 003 rows (ragged shards) and `gather_mixed` on config 5's real
    segment table, to every rank and to rank 0 alone (`dst=0`, what bench.py uses): no 8-GPU node is available to the
    build, so this is where an N = 8 cut of the launcher's helpers has run."""
    mp.spawn(_n8_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    assert all(bool(np.load(str(tmp_path / f"ok{r}.npy"))[0]) for r in range(8))


def test_root_only_gather_two_ranks(tmp_path):
    """`dst=0` with equal and ragged shards on two ranks, against the all-gather."""
    mp.spawn(_root_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert bool(np.load(str(tmp_path / "root.npy"))[0])


def _root_worker(rank, world, port, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for n in (6, 7):
            lo, hi = pdist.shard_bounds(n, world, rank)
            local = torch.from_numpy(_rows_value(np.arange(lo, hi), 5))
            a = pdist.gather_rows(local, n)
            r = pdist.gather_rows(local, n, dst=0)
            ok = ok and ((r is None) if rank else bool(torch.equal(a, r)))
        rows, _ = pdist.shard_segments(MIXED, world, rank)
        m = torch.from_numpy(_rows_value(rows, 2))
        a = pdist.gather_mixed(m, MIXED, 14)
        r = pdist.gather_mixed(m, MIXED, 14, dst=0)
        ok = ok and ((r is None) if rank else bool(torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(r, nan=-1.0))))
        if rank == 0:
            np.save(os.path.join(result_dir, "root.npy"), np.array([ok]))
    finally:
        dist.destroy_process_group()
