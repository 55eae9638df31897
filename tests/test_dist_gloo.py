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
