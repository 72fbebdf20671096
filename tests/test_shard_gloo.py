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
    # -- block kernels: this rank's blocks only (a raster span of CTUs, bench.py's default partition); checksums summed over ranks
    sp = shard.SpanShard(W, H, world, rank)
    cur, ref = shard.block_pairs_of_ctu_span(torch, dev, seed, sp.ctus(), frames, 8)
    res = shard.residual_blocks_of_ctu_span(torch, dev, seed, sp.ctus(), frames, 32)
    assert cur.shape[0] == sp.blocks(8) * frames and res.shape[0] == sp.blocks(32) * frames
    sad = torch.from_numpy(O.cost_nxn_many("sad", 8, cur.numpy(), ref.numpy(), threads=1).astype(np.int64))
    satd = torch.from_numpy(O.cost_nxn_many("satd", 8, cur.numpy(), ref.numpy(), threads=1).astype(np.int64))
    coef = torch.from_numpy(O.transform_many("dct", 32, res.numpy(), threads=1))
    ca, cw = shard.coeff_checksum(torch, coef)
    sums = torch.tensor([shard.cost_checksum(sad), shard.cost_checksum(satd), ca, cw, cur.shape[0], res.shape[0]], dtype=torch.int64)
    dist.all_reduce(sums)
    # -- search: frame by frame, rec rows of the previous frame -> extended buffer -> exchange -> search inside rows + halo
    pus, spans = shard.shard_pus(np, sh, (8, 16, 32, 64), ME_PU)
    pus = pus[::7]                                                  # the oracle takes ~1 ms per PU
    groups = shard.search_groups(np, sh, pus, boundary_ctu_rows=1)   # margin 48 < 64: one CTU row next to a shared edge reads halo rows
    ext_ref = torch.zeros((sh.ext_rows, W), dtype=torch.uint8)
    results = []
    for f in range(1, frames + 1):
        own = shard.shard_plane(torch, dev, sh, seed, f - 1, 1, extended=False)
        ext_ref[sh.top:sh.top + sh.rows] = own
        pic = shard.shard_plane(torch, dev, sh, seed, f, 0)
        res_f = {}
        # the interior CTU rows BEFORE the halo rows are there (they are zero at this point): confined to the rank's own rows
        name, idx, tile = groups[0]
        assert name == "interior"
        halo_poisoned = ext_ref.clone()
        if rank > 0:
            halo_poisoned[:sh.top] = 255 - halo_poisoned[:sh.top]
        res_f[name] = O.search_pu_batch(pic.numpy(), halo_poisoned.numpy(), pus[idx], me_params(lambda_cost=20, mv_constraint=4, tile=tile))
        shard.exchange_halo_into(ext_ref, sh, dist)
        whole = shard.full_plane(torch, dev, W, H, seed, f - 1, 1)
        assert torch.equal(ext_ref, whole[sh.ext_lo:sh.ext_hi]), "halo rows differ from the neighbour's rows"
        for name, idx, tile in groups[1:]:
            res_f[name] = O.search_pu_batch(pic.numpy(), ext_ref.numpy(), pus[idx], me_params(lambda_cost=20, mv_constraint=4, tile=tile))
        results.append(res_f)
    gathered = [None] * world
    dist.all_gather_object(gathered, (sh.describe(), [(n, i, t) for (n, i, t) in groups], sh.ext_lo, pus, results))
    if rank == 0:
        ok = True
        # unsharded: the same PUs on the whole frame, under the rectangle of their group moved to frame coordinates
        for (_, grp, ext_lo, p, res_sh) in gathered:
            pf = p.copy()
            pf["y"] += ext_lo                                       # extended-buffer -> frame coordinates
            for (name, idx, tile) in grp:
                prm_full = me_params(lambda_cost=20, mv_constraint=4, tile=(tile[0], tile[1] + ext_lo, tile[2], tile[3]))
                for f in range(1, frames + 1):
                    pic = shard.full_plane(torch, dev, W, H, seed, f, 0).numpy()
                    ref_full = shard.full_plane(torch, dev, W, H, seed, f - 1, 1).numpy()
                    want = O.search_pu_batch(pic, ref_full, pf[idx], prm_full)
                    ok = ok and bool((want.view(np.int32) == res_sh[f - 1][name].view(np.int32)).all())
        found = sum(int((r["cost"] != 0xFFFFFFFF).sum()) for g_ in gathered for rf in g_[4] for r in rf.values())
        total = sum(len(r) for g_ in gathered for rf in g_[4] for r in rf.values())
        n_boundary = [sum(len(i) for (n, i, t) in g_[1] if n == "boundary") for g_ in gathered]
        q.put((ok, sums.tolist(), [g_[0] for g_ in gathered], found, total, n_boundary))
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
    assert single[5] == [0]                       # a rank without neighbours has no boundary PUs
    for world in (2, 3):
        ok, sums, descr, found, total, n_boundary = _run_shard_leg(world)
        assert all(b > 0 for b in n_boundary)
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


def test_span_shards_balance_to_one_ctu_and_cover_the_frame():
    """the block-kernel partition of bench.py's shard_4k leg: raster spans of CTUs; 4K over 8 ranks = 255 CTUs each (whole CTU rows: 5,5,4,...)"""
    for (w, h) in ((3840, 2160), (1920, 1080), (200, 136)):
        for world in (1, 2, 3, 4, 8):
            shards = [shard.SpanShard(w, h, world, r) for r in range(world)]
            sizes = [s.ctu_hi - s.ctu_lo for s in shards]
            assert sum(sizes) == shards[0].n_ctus and max(sizes) - min(sizes) <= 1
            assert [s.ctu_lo for s in shards[1:]] == [s.ctu_hi for s in shards[:-1]]
            for n in (8, 32):
                assert sum(s.blocks(n) for s in shards) == (w // n) * (h // 64) * (64 // n) + (w // n) * ((h % 64) // n)
    sh8 = [shard.SpanShard(3840, 2160, 8, r) for r in range(8)]
    assert [s.ctu_hi - s.ctu_lo for s in sh8] == [255] * 8
    assert shard.ideal_speedup([s.blocks(8) for s in sh8]) > 7.9
    assert abs(shard.ideal_speedup([shard.RowShard(3840, 2160, 8, r).blocks(8) for r in range(8)]) - 6.75) < 1e-9
    # generated per CTU: the union over any partition is the same data
    a = torch.cat([shard.block_pairs_of_ctu_span(torch, "cpu", 3, shard.SpanShard(200, 136, 3, r).ctus(), 2, 8)[0] for r in range(3)])
    b = shard.block_pairs_of_ctu_span(torch, "cpu", 3, shard.SpanShard(200, 136, 1, 0).ctus(), 2, 8)[0]
    assert torch.equal(a, b)
