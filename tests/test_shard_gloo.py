"""CPU, world_size 2 over gloo: the N>1 path of the benchmark -- CTU-row sharding with
no data-path collective, MAX-over-ranks timing -- checked end to end with the oracle
standing in for the GPU kernels (this is a test; the product never uses the oracle)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from kvazaar_amd import shard  # noqa: E402


def test_row_ranges_partition_exactly():
    for h in (1080, 2160, 64, 65, 130):
        rows = shard.ctu_rows(h)
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                lo, hi = shard.row_range(rows, world, r)
                assert 0 <= lo <= hi <= rows
                covered += list(range(lo, hi))
            assert covered == list(range(rows))
            sizes = [shard.row_range(rows, world, r)[1] - shard.row_range(rows, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    # 4K at 8 GPUs: 34 CTU rows -> 4 or 5 rows per GPU
    assert sorted({shard.row_range(34, 8, r)[1] - shard.row_range(34, 8, r)[0] for r in range(8)}) == [4, 5]
    assert shard.halo_rows(2160, 8, 0, 80) == (0, 5 * 64 + 80)
    assert shard.pixel_rows(2160, 8, 7) == (30 * 64, 2160)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    g = np.random.default_rng(5)
    W, H = 256, 200                       # 4 CTU rows, last one ragged (8 px)
    cur = g.integers(0, 256, (H, W), dtype=np.uint8)
    ref = g.integers(0, 256, (H, W), dtype=np.uint8)
    y_lo, y_hi = shard.pixel_rows(H, world, rank)
    # each rank: sad_8x8 of every full 8x8 block of its CTU rows (block pairs made contiguous, as the batched ABI wants)
    ys = [y for y in range(0, H - 7, 8) if y_lo <= y < y_hi]
    blk_c = np.stack([cur[y:y + 8, x:x + 8].ravel() for y in ys for x in range(0, W, 8)]) if ys else np.zeros((0, 64), np.uint8)
    blk_r = np.stack([ref[y:y + 8, x:x + 8].ravel() for y in ys for x in range(0, W, 8)]) if ys else np.zeros((0, 64), np.uint8)
    assert blk_c.shape[0] == shard.blocks_in_rows(W, y_lo, y_hi, 8)
    costs = O.cost_nxn_batch("sad", 8, blk_c, blk_r)
    # no collective on the data path; for the check only, gather the shard results on rank 0
    gathered = [None] * world
    dist.all_gather_object(gathered, (y_lo, y_hi, costs))
    dt = shard.max_over_ranks(0.010 * (rank + 1), dist)
    if rank == 0:
        full_c = np.stack([cur[y:y + 8, x:x + 8].ravel() for y in range(0, H - 7, 8) for x in range(0, W, 8)])
        full_r = np.stack([ref[y:y + 8, x:x + 8].ravel() for y in range(0, H - 7, 8) for x in range(0, W, 8)])
        want = O.cost_nxn_batch("sad", 8, full_c, full_r)
        got = np.concatenate([c for (_, _, c) in sorted(gathered, key=lambda t: t[0])])
        q.put((bool((got == want).all()), dt, [(a, b) for (a, b, _) in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ctu_row_sharding_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    ok, dt, ranges = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert ok, "sharded result differs from the single-rank result"
    assert abs(dt - 0.020) < 1e-9          # MAX over ranks
    assert ranges[0][1] == ranges[1][0]    # contiguous, disjoint shards


def _halo_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.random.default_rng(9)
    W, H, margin = 128, 64 * 7 + 24, 40            # 8 CTU rows, the last one ragged
    plane = torch.from_numpy(g.integers(0, 256, (H, W), dtype=np.uint8))
    lo, hi = shard.pixel_rows(H, world, rank)
    ext, off = shard.exchange_halo(plane[lo:hi].clone(), margin, dist)
    want_lo, want_hi = shard.halo_rows(H, world, rank, margin)
    ok = (lo - off == want_lo) and (ext.shape[0] == want_hi - want_lo) and bool(torch.equal(ext, plane[want_lo:want_hi]))
    res = [None] * world
    dist.all_gather_object(res, ok)
    if rank == 0:
        q.put(all(res))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_of_reconstructed_rows(world):
    """the one exchange step of a CTU-row sharded encoder: neighbour rows arrive in place (gloo stands in for RCCL)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + world
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    assert q.get(timeout=10) is True
