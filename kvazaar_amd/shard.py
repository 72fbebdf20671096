"""Multi-GPU partition of the hot path: blocks are independent, so a
frame (or a batch of frames) is cut into contiguous CTU-row shards, one per rank,
and there is no collective on the data path -- costs / coefficients stay on the
rank that produced them.  The helpers here are pure host logic (no GPU), shared by
bench.py and the gloo tests."""

CTU = 64   # LCU_WIDTH, src/global.h:137


def ctu_rows(frame_height, ctu=CTU):
    return (frame_height + ctu - 1) // ctu


def row_range(n_rows, world, rank):
    """contiguous [lo, hi) of CTU rows owned by `rank`; the first n_rows % world ranks get one extra row"""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pixel_rows(frame_height, world, rank, ctu=CTU):
    lo, hi = row_range(ctu_rows(frame_height, ctu), world, rank)
    return lo * ctu, min(hi * ctu, frame_height)


def halo_rows(frame_height, world, rank, margin, ctu=CTU):
    """pixel rows of the REFERENCE frame a rank needs for motion search: its own rows +- margin
    (one CTU row + filter taps + deblock/SAO delay: SURVEY 8e), clipped to the frame"""
    lo, hi = pixel_rows(frame_height, world, rank, ctu)
    return max(0, lo - margin), min(frame_height, hi + margin)


def blocks_in_rows(frame_width, y_lo, y_hi, n):
    """number of full n x n blocks whose top-left lies in pixel rows [y_lo, y_hi) on the n-grid"""
    first = (y_lo + n - 1) // n
    last = y_hi // n
    return max(0, last - first) * (frame_width // n)


def max_over_ranks(dt, dist=None, device=None):
    """bench contract: the reported time is the MAX over ranks"""
    if dist is None or not dist.is_initialized():
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def exchange_halo(own_rows, margin, dist):
    """The one exchange step a CTU-row sharded ENCODER needs: after a frame is reconstructed, every rank
    sends the `margin` pixel rows at the top / bottom of its shard to the rank above / below, so that the next frame's
    motion search (kvz_hip_search_pu_batch) can read margin rows beyond its own CTU rows of the reference.

    own_rows: torch uint8 tensor [rows, width] (this rank's rows of the reconstructed plane; on the GPU with the nccl
    backend -- RCCL send/recv between ring neighbours over one xGMI link -- or on the CPU with gloo).  Returns
    (extended tensor, first_row_offset): own rows with up to `margin` neighbour rows attached above and below.
    Point-to-point only: 2 * margin * width bytes per interior boundary per plane (4K luma, margin 80: 0.3 MB)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1 or margin <= 0:
        return own_rows, 0
    rows = own_rows.shape[0]
    if rows < margin:
        raise ValueError("shard of %d rows is thinner than the halo margin %d: use fewer ranks or a multi-hop exchange" % (rows, margin))
    up = torch.empty((margin,) + tuple(own_rows.shape[1:]), dtype=own_rows.dtype, device=own_rows.device) if rank > 0 else None
    down = torch.empty((margin,) + tuple(own_rows.shape[1:]), dtype=own_rows.dtype, device=own_rows.device) if rank < world - 1 else None
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, own_rows[:margin].contiguous(), rank - 1))
        ops.append(dist.P2POp(dist.irecv, up, rank - 1))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, own_rows[rows - margin:].contiguous(), rank + 1))
        ops.append(dist.P2POp(dist.irecv, down, rank + 1))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    parts = ([up] if up is not None else []) + [own_rows] + ([down] if down is not None else [])
    return torch.cat(parts, dim=0), (margin if up is not None else 0)
