"""CPU, world_size 2 and 3 over gloo: the N>1 path of the benchmark (bench.py's shard_4k leg) -- the CTU-row partition,
the deterministic per-CTU-row workload, the in-place halo exchange of reconstructed rows and the search of a shard's PUs
inside its rows + halo -- through the very functions of kvazaar_amd/shard.py that bench.py runs on the GPUs, with the
oracle standing in for the GPU kernels (this is a test; the product never uses the oracle).  Asserted: sharded results
== unsharded results, checksums independent of the world size, MAX-over-ranks timing."""
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


# ---- the shard leg of bench.py, rehearsed: same shard.py code path, CPU tensors, oracle as the kernels ----
def _shard_leg_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from patterns import ME_PU, me_params
    dev = torch.device("cpu")
    W, H, margin, frames, seed = 192, 64 * 5 + 40, 48, 2, 77        # 6 CTU rows, the last one ragged (40 px)
    sh = shard.RowShard(W, H, world, rank, margin)
    # -- block kernels: this rank's blocks only; checksums summed over ranks
    cur, ref, res = [], [], []
    for r, h in sh.ctu_row_heights():
        c, f = shard.block_pairs_of_ctu_row(torch, dev, seed, r, h, W, frames, 8)
        cur.append(c); ref.append(f)
        res.append(shard.residual_blocks_of_ctu_row(torch, dev, seed, r, h, W, frames, 32))
    cur, ref, res = torch.cat(cur), torch.cat(ref), torch.cat(res)
    assert cur.shape[0] == sh.blocks(8) * frames and res.shape[0] == sh.blocks(32) * frames
    sad = torch.from_numpy(O.cost_nxn_many("sad", 8, cur.numpy(), ref.numpy(), threads=1).astype(np.int64))
    satd = torch.from_numpy(O.cost_nxn_many("satd", 8, cur.numpy(), ref.numpy(), threads=1).astype(np.int64))
    coef = torch.from_numpy(O.transform_many("dct", 32, res.numpy(), threads=1))
    ca, cw = shard.coeff_checksum(torch, coef)
    sums = torch.tensor([shard.cost_checksum(sad), shard.cost_checksum(satd), ca, cw, cur.shape[0], res.shape[0]], dtype=torch.int64)
    dist.all_reduce(sums)
    # -- search: frame by frame, rec rows of the previous frame -> extended buffer -> exchange -> search inside rows + halo
    pus, spans = shard.shard_pus(np, sh, (8, 16, 32, 64), ME_PU)
    pus = pus[::7]                                                  # the oracle takes ~1 ms per PU
    prm = me_params(lambda_cost=20, mv_constraint=4)                # tile 0 x 0: the extended buffer is the tile
    ext_ref = torch.zeros((sh.ext_rows, W), dtype=torch.uint8)
    results = []
    for f in range(1, frames + 1):
        own = shard.shard_plane(torch, dev, sh, seed, f - 1, 1, extended=False)
        ext_ref[sh.top:sh.top + sh.rows] = own
        shard.exchange_halo_into(ext_ref, sh, dist)
        whole = shard.full_plane(torch, dev, W, H, seed, f - 1, 1)
        assert torch.equal(ext_ref, whole[sh.ext_lo:sh.ext_hi]), "halo rows differ from the neighbour's rows"
        pic = shard.shard_plane(torch, dev, sh, seed, f, 0)
        results.append(O.search_pu_batch(pic.numpy(), ext_ref.numpy(), pus, prm))
    gathered = [None] * world
    dist.all_gather_object(gathered, (sh.describe(), sh.tile_in_frame(), sh.ext_lo, pus, results))
    if rank == 0:
        ok = True
        # unsharded: the same PUs on the whole frame, under the tile rectangle of their shard
        for (_, tile, ext_lo, p, res_sh) in gathered:
            pf = p.copy()
            pf["y"] += ext_lo                                       # extended-buffer -> frame coordinates
            prm_full = me_params(lambda_cost=20, mv_constraint=4, tile=tile)
            for f in range(1, frames + 1):
                pic = shard.full_plane(torch, dev, W, H, seed, f, 0).numpy()
                ref_full = shard.full_plane(torch, dev, W, H, seed, f - 1, 1).numpy()
                want = O.search_pu_batch(pic, ref_full, pf, prm_full)
                ok = ok and bool((want.view(np.int32) == res_sh[f - 1].view(np.int32)).all())
        found = sum(int((r["cost"] != 0xFFFFFFFF).sum()) for g_ in gathered for r in g_[4])
        total = sum(len(r) for g_ in gathered for r in g_[4])
        q.put((ok, sums.tolist(), [g_[0] for g_ in gathered], found, total))
    dist.barrier()
    dist.destroy_process_group()


def _run_shard_leg(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + world + (os.getpid() % 500)
    procs = [ctx.Process(target=_shard_leg_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    out = q.get(timeout=300)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return out


def test_shard_leg_sharded_equals_unsharded_and_checksums_do_not_depend_on_world():
    single = _run_shard_leg(1)
    for world in (2, 3):
        ok, sums, descr, found, total = _run_shard_leg(world)
        assert ok, "a shard's search differs from the unsharded search under its tile rectangle (world %d)" % world
        assert sums == single[1], "kernel checksums depend on the partition (world %d)" % world
        assert descr[0]["pixel_rows"][0] == 0 and descr[-1]["pixel_rows"][1] == 64 * 5 + 40
        assert all(descr[i]["pixel_rows"][1] == descr[i + 1]["pixel_rows"][0] for i in range(world - 1))
        assert found == total > 0
    assert single[0] and single[3] == single[4] > 0


def test_row_shard_geometry_4k():
    """3840x2160 over 8 ranks (BASELINE config 5): 34 CTU rows -> 5,5,4,4,4,4,4,4; 80 halo rows towards each neighbour"""
    sizes = []
    for r in range(8):
        sh = shard.RowShard(3840, 2160, 8, r)
        sizes.append(sh.ctu_hi - sh.ctu_lo)
        assert sh.ext_lo == (sh.y_lo - 80 if r else 0) and sh.ext_hi == (sh.y_hi + 80 if r < 7 else 2160)
        assert sh.top == (80 if r else 0)
    assert sizes == [5, 5, 4, 4, 4, 4, 4, 4]
    assert sum(shard.RowShard(3840, 2160, 8, r).blocks(8) for r in range(8)) == 129600
    assert sum(shard.RowShard(3840, 2160, 8, r).blocks(32) for r in range(8)) == 8040
    assert shard.HALO_ROWS * 3840 == 307200            # bytes per boundary per frame each way (luma)
    with pytest.raises(ValueError):
        shard.RowShard(3840, 2160, 34, 3, margin=80)   # one CTU row per rank is thinner than the halo
