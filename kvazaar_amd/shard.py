"""Multi-GPU partition of the hot path (BASELINE.json configs[4]): a frame -- or a batch of frames -- is
cut into contiguous CTU-row shards, one per rank (one process per GPU).  Blocks are independent, so the block kernels
need no collective at all: costs / coefficients stay on the rank that produced them.  The one exchange the path has is
the reconstructed rows a neighbour's motion search may read: `margin` pixel rows on either side of a shard boundary,
sent point to point between ring neighbours (torch.distributed isend / irecv: RCCL over one xGMI link per pair on the
GPUs, gloo on the CPU rehearsal).  The reference's analogue is the bounded cross-row read of its WPP / tile
parallelism (src/encoderstate.c:777-828, src/encoder.c:240-241, src/search_inter.c:87-172).

Everything here is host / torch logic that runs unchanged on CPU tensors (tests/test_shard_gloo.py, world 2 and 3
over gloo) and on GPU tensors (bench.py --gpus N): the partition, the deterministic per-CTU-row synthetic workload
(its content does not depend on how many ranks share the frame, so checksums of the results must not either), the
in-place halo exchange, the PU lists of a shard and the checksums.  No kernel is called from here."""

CTU = 64            # LCU_WIDTH, src/global.h:137
HALO_ROWS = 80      # 1 CTU row + 4 filter taps + 10 rows of deblock / SAO delay, rounded up (global.h:163,175)


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


def span_range(n_units, world, rank):
    """contiguous [lo, hi) of `n_units` units in raster order owned by `rank`; sizes differ by at most one"""
    return row_range(n_units, world, rank)


def ideal_speedup(sizes):
    """the best speed-up a partition into shards of these sizes can reach over one rank: total / largest"""
    sizes = list(sizes)
    return float(sum(sizes)) / float(max(sizes)) if sizes and max(sizes) > 0 else 0.0


class SpanShard:
    """One rank's share of a frame for the BLOCK kernels: a contiguous raster span of CTUs [ctu_lo, ctu_hi).  Blocks are
    independent, so a shard need not be a rectangle: spans balance to within one CTU (4K over 8 ranks: 2040 CTUs -> 255
    each, ideal speed-up 8.0, where whole CTU rows give 5,5,4,... of 34 and at best 6.8).  The reference's analogue is a
    slice of consecutive LCUs (slices end at arbitrary LCU addresses, encoder.c:513-560 `slice_addresses_in_ts`); its tiles
    split in both directions (encoder.c:430-510).  A span longer than one CTU row still has only its two raster neighbours
    as neighbours, so the exchange pattern of a row shard (rank r <-> r +- 1) carries over."""

    def __init__(self, width, height, world, rank, ctu=CTU):
        self.width, self.height, self.world, self.rank, self.ctu = width, height, world, rank, ctu
        self.cols, self.rows_ctu = (width + ctu - 1) // ctu, (height + ctu - 1) // ctu
        self.n_ctus = self.cols * self.rows_ctu
        self.ctu_lo, self.ctu_hi = span_range(self.n_ctus, world, rank)

    def ctus(self):
        """[(ctu index, width, height of the CTU inside the frame)] of this shard, raster order"""
        out = []
        for i in range(self.ctu_lo, self.ctu_hi):
            cy, cx = divmod(i, self.cols)
            out.append((i, min(self.ctu, self.width - cx * self.ctu), min(self.ctu, self.height - cy * self.ctu)))
        return out

    def blocks(self, n):
        """full n x n blocks of one frame that lie in this shard"""
        return sum((w // n) * (h // n) for (_, w, h) in self.ctus())

    def describe(self):
        return {"ctus": [self.ctu_lo, self.ctu_hi], "of": self.n_ctus}


def max_over_ranks(dt, dist=None, device=None):
    """bench contract: the reported time is the MAX over ranks"""
    if dist is None or not dist.is_initialized():
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class RowShard:
    """One rank's share of a width x height frame: CTU rows [ctu_lo, ctu_hi) = pixel rows [y_lo, y_hi), and the
    extended row range [ext_lo, ext_hi) = own rows + `margin` halo rows towards each neighbour (none at a frame edge).
    Buffers that hold a shard of a plane hold the EXTENDED range; `top` is where the own rows start inside them."""

    def __init__(self, width, height, world, rank, margin=HALO_ROWS, ctu=CTU):
        self.width, self.height, self.world, self.rank, self.margin, self.ctu = width, height, world, rank, margin, ctu
        self.ctu_lo, self.ctu_hi = row_range(ctu_rows(height, ctu), world, rank)
        self.y_lo, self.y_hi = self.ctu_lo * ctu, min(self.ctu_hi * ctu, height)
        self.rows = self.y_hi - self.y_lo
        self.ext_lo = max(0, self.y_lo - margin) if rank > 0 else self.y_lo
        self.ext_hi = min(height, self.y_hi + margin) if rank < world - 1 else self.y_hi
        self.top = self.y_lo - self.ext_lo
        self.ext_rows = self.ext_hi - self.ext_lo
        # every rank evaluates every rank's share, so that all of them raise together (one rank raising alone would leave the
        # others waiting in the exchange until its timeout)
        if world > 1:
            n_rows = ctu_rows(height, ctu)
            thinnest = min(min(row_range(n_rows, world, r)[1] * ctu, height) - row_range(n_rows, world, r)[0] * ctu for r in range(world))
            if thinnest < margin:
                raise ValueError("a shard of %d rows is thinner than the halo margin %d: use fewer ranks" % (thinnest, margin))

    def blocks(self, n):
        """full n x n blocks of one frame that lie in this shard"""
        return blocks_in_rows(self.width, self.y_lo, self.y_hi, n)

    def ctu_row_heights(self):
        """[(ctu_row, pixel rows of it inside the frame)] for the rows of this shard (the frame's last row may be ragged)"""
        return [(r, min(self.ctu, self.height - r * self.ctu)) for r in range(self.ctu_lo, self.ctu_hi)]

    def tile_in_frame(self):
        """(x, y, w, h) of the extended range in FRAME coordinates: the tile rectangle under which an unsharded search of
        this shard's PUs reads exactly what the sharded one can (kvz_hip_me_params.tile_*, mv_constraint 4)"""
        return (0, self.ext_lo, self.width, self.ext_rows)

    def describe(self):
        return {"ctu_rows": [self.ctu_lo, self.ctu_hi], "pixel_rows": [self.y_lo, self.y_hi], "with_halo": [self.ext_lo, self.ext_hi]}


# ---------------------------------------------------------------------------------------------------------------------
# deterministic synthetic workload, generated per CTU row so that every rank can make exactly its own rows and the
# union over ranks is the same data for every world size
# ---------------------------------------------------------------------------------------------------------------------
def _gen(torch, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def block_pairs_of_ctu_row(torch, device, seed, ctu_row, row_height, width, frames, n=8):
    """(cur, ref): uint8 [frames * blocks, n*n] -- the n x n luma block pairs of one CTU row of `frames` frames, in the
    contiguous layout of cost_pixel_nxn_func (strategies-picture.h:102); ref = cur + small noise"""
    count = frames * (row_height // n) * (width // n)
    g = _gen(torch, device, seed * 1000003 + ctu_row * 2 + 0)
    cur = torch.randint(0, 256, (count, n * n), dtype=torch.uint8, device=device, generator=g)
    noise = torch.randint(-8, 9, (count, n * n), dtype=torch.int16, device=device, generator=g)
    ref = (cur.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    return cur, ref


def residual_blocks_of_ctu_row(torch, device, seed, ctu_row, row_height, width, frames, n=32):
    """int16 [frames * blocks, n*n] residual blocks in [-255, 255] of one CTU row (dct_func layout, strategies-dct.h:31)"""
    count = frames * (row_height // n) * (width // n)
    g = _gen(torch, device, seed * 1000003 + ctu_row * 2 + 1)
    return torch.randint(-255, 256, (count, n * n), dtype=torch.int16, device=device, generator=g)


def block_pairs_of_ctu_span(torch, device, seed, ctus, frames, n=8):
    """(cur, ref): the n x n block pairs of the CTUs `ctus` = [(index, w, h)] of `frames` frames, generated PER CTU from a seed
    that names the CTU, so that the union over any partition is the same data (each CTU keeps its own random stream)."""
    cur, ref = [], []
    for (i, w, h) in ctus:
        count = frames * (h // n) * (w // n)
        g = _gen(torch, device, seed * 1000003 + 500009 + i * 2)
        c = torch.randint(0, 256, (count, n * n), dtype=torch.uint8, device=device, generator=g)
        noise = torch.randint(-8, 9, (count, n * n), dtype=torch.int16, device=device, generator=g)
        cur.append(c)
        ref.append((c.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8))
    if not cur:
        z = torch.zeros((0, n * n), dtype=torch.uint8, device=device)
        return z, z.clone()
    return torch.cat(cur), torch.cat(ref)


def residual_blocks_of_ctu_span(torch, device, seed, ctus, frames, n=32):
    """int16 [blocks, n*n] residual blocks in [-255, 255] of the CTUs `ctus`, one stream per CTU (see block_pairs_of_ctu_span)"""
    res = []
    for (i, w, h) in ctus:
        count = frames * (h // n) * (w // n)
        g = _gen(torch, device, seed * 1000003 + 500009 + i * 2 + 1)
        res.append(torch.randint(-255, 256, (count, n * n), dtype=torch.int16, device=device, generator=g))
    if not res:
        return torch.zeros((0, n * n), dtype=torch.int16, device=device)
    return torch.cat(res)


NOMINAL_MV = (12, 4)      # quarter-pel: the synthetic sequence moves 3 px / 1 px per frame (plane_rows_of_ctu_row)


def plane_rows_of_ctu_row(torch, device, seed, frame, ctu_row, row_height, width, kind):
    """uint8 [row_height, width]: one CTU row of the luma plane `kind` (0 source, 1 reconstruction) of frame `frame`.
    The source is a smooth texture (sums of shifted random rows) whose window moves 3 px / 1 px per frame, the
    reconstruction is the source plus coding-like noise: motion search finds real minima, and a row's content depends on
    (frame, ctu_row) only."""
    g = _gen(torch, device, seed * 7919 + ctu_row)
    base = torch.randint(0, 256, (row_height + 16, width + 64), dtype=torch.int32, device=device, generator=g)
    sm = base
    for _ in range(2):
        sm = (sm + sm.roll(1, 0) + sm.roll(1, 1) + sm.roll((1, 1), (0, 1))) // 4
    sm = ((sm - 128) * 4 + 128).clamp_(0, 255)
    dx, dy = (3 * frame) % 48, frame % 12
    rows = sm[dy:dy + row_height, dx:dx + width]
    if kind == 1:
        gn = _gen(torch, device, seed * 104729 + frame * 4099 + ctu_row)
        rows = (rows + torch.randint(-3, 4, (row_height, width), dtype=torch.int32, device=device, generator=gn)).clamp_(0, 255)
    return rows.to(torch.uint8).contiguous()


def shard_plane(torch, device, shard, seed, frame, kind, extended=True):
    """this rank's rows of one plane, in an extended buffer [ext_rows, width] (halo rows zero: they arrive by exchange)
    or as the own rows only"""
    own = torch.cat([plane_rows_of_ctu_row(torch, device, seed, frame, r, h, shard.width, kind) for r, h in shard.ctu_row_heights()], dim=0)
    if not extended:
        return own
    ext = torch.zeros((shard.ext_rows, shard.width), dtype=torch.uint8, device=device)
    ext[shard.top:shard.top + shard.rows] = own
    return ext


def full_plane(torch, device, width, height, seed, frame, kind):
    """the whole plane, as one rank of a world of one would hold it"""
    return shard_plane(torch, device, RowShard(width, height, 1, 0), seed, frame, kind)


# ---------------------------------------------------------------------------------------------------------------------
# the exchange
# ---------------------------------------------------------------------------------------------------------------------
def exchange_halo_into(ext, shard, dist, staging=None):
    """The one exchange step of a CTU-row sharded encoder, in place: `ext` is this rank's extended buffer
    [ext_rows, width] whose own rows [top, top + rows) hold the newly reconstructed plane; the `margin` rows at the top
    / bottom edge of the own rows go to the rank above / below, whose rows arrive in this buffer's halo ranges.
    Point-to-point only -- 2 * margin * width bytes per interior boundary per plane (4K luma, 80 rows: 0.3 MB each way),
    one grouped isend / irecv batch.  With the nccl backend (RCCL) the transfers are enqueued behind the work already
    on the current torch stream and the stream waits for them on the device: the host does not block.
    staging: for a rehearsal with GPU tensors over a CPU backend (gloo), a dict for the pinned bounce buffers."""
    world, rank, m = shard.world, shard.rank, shard.margin
    if world == 1 or m <= 0:
        return
    import torch
    top, rows = shard.top, shard.rows
    send_up, recv_up = ext[top:top + m], ext[top - m:top] if rank > 0 else None
    send_dn, recv_dn = ext[top + rows - m:top + rows], ext[top + rows:top + rows + m] if rank < world - 1 else None
    via_host = ext.is_cuda and dist.get_backend() != "nccl"
    if via_host:
        if staging is None:
            staging = {}
        def host(name, like):
            if name not in staging:
                staging[name] = torch.empty(like.shape, dtype=like.dtype).pin_memory()
            return staging[name]
        h_su, h_sd = host("su", send_up).copy_(send_up), host("sd", send_dn).copy_(send_dn)
        h_ru = host("ru", send_up) if rank > 0 else None
        h_rd = host("rd", send_dn) if rank < world - 1 else None
        torch.cuda.current_stream().synchronize()
        s_up, s_dn, r_up, r_dn = h_su, h_sd, h_ru, h_rd
    else:
        s_up, s_dn, r_up, r_dn = send_up, send_dn, recv_up, recv_dn
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, s_up, rank - 1))
        ops.append(dist.P2POp(dist.irecv, r_up, rank - 1))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, s_dn, rank + 1))
        ops.append(dist.P2POp(dist.irecv, r_dn, rank + 1))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if via_host:
        if rank > 0:
            recv_up.copy_(r_up, non_blocking=True)
        if rank < world - 1:
            recv_dn.copy_(r_dn, non_blocking=True)


def exchange_halo(own_rows, margin, dist):
    """Convenience form: own rows [rows, width] in, (extended tensor, index of the first own row) out."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1 or margin <= 0:
        return own_rows, 0
    rows = own_rows.shape[0]
    if rows < margin:
        raise ValueError("shard of %d rows is thinner than the halo margin %d: use fewer ranks or a multi-hop exchange" % (rows, margin))
    up = margin if rank > 0 else 0
    dn = margin if rank < world - 1 else 0
    ext = torch.empty((up + rows + dn,) + tuple(own_rows.shape[1:]), dtype=own_rows.dtype, device=own_rows.device)
    ext[up:up + rows] = own_rows

    class _S:          # the fields exchange_halo_into reads
        pass
    s = _S()
    s.world, s.rank, s.margin, s.top, s.rows = world, rank, margin, up, rows
    exchange_halo_into(ext, s, dist)
    return ext, up


# ---------------------------------------------------------------------------------------------------------------------
# the PUs a shard searches, and checksums that do not depend on the partition
# ---------------------------------------------------------------------------------------------------------------------
def shard_pus(np, shard, sizes=(8, 16, 32, 64), me_pu_dtype=None):
    """kvz_hip_me_pu records (include/kvz_hip.h) of every full n x n PU of the shard's own rows, for each n in `sizes`,
    in EXTENDED-buffer coordinates (y - ext_lo).  AMVP / merge candidates: the zero vector and the sequence's nominal
    motion, the same for every PU (candidate derivation is the host encoder's business: inter.c:1209,1314).
    Returns (records, {n: (first, count)})."""
    dt = me_pu_dtype
    recs, spans = [], {}
    for n in sizes:
        ys = list(range((shard.y_lo + n - 1) // n * n, shard.y_hi - n + 1, n))
        xs = list(range(0, shard.width - n + 1, n))
        a = np.zeros(len(ys) * len(xs), dtype=dt)
        if len(a):
            a["x"] = np.tile(np.asarray(xs, dtype=np.int32), len(ys))
            a["y"] = np.repeat(np.asarray(ys, dtype=np.int32), len(xs)) - shard.ext_lo
            a["width"], a["height"] = n, n
            a["mv_cand"][:, 0] = (0, 0)
            a["mv_cand"][:, 1] = NOMINAL_MV
            a["num_merge_cand"] = 1
            a["merge"]["mv"][:, 0] = NOMINAL_MV
            a["merge"]["usable"][:, 0] = 1
            a["merge"]["same_ref"][:, 0] = 1
        spans[n] = (sum(len(r) for r in recs), len(a))
        recs.append(a)
    return np.concatenate(recs), spans


BOUNDARY_CTU_ROWS = 2      # HALO_ROWS = 80 > 64: the two CTU rows next to a shared edge are the ones whose search may read halo rows


def search_groups(np, shard, pus, boundary_ctu_rows=BOUNDARY_CTU_ROWS):
    """Splits a shard's PUs (shard_pus: extended-buffer coordinates) for an exchange that overlaps with the search:
    -> [(name, index array, tile)] with tile = (x, y, w, h) in EXTENDED-buffer coordinates.
      "interior": PUs of CTU rows at least `boundary_ctu_rows` away from every edge the shard shares with a neighbour; their
                  vectors are confined to the shard's OWN rows (kvz_hip_me_params.tile_*, mv_constraint 4), so this group
                  can be searched while the halo rows are still in flight;
      "boundary": the PUs of the CTU rows next to a shared edge, searched under own rows + halo once the exchange has landed.
    A rank without neighbours (world 1) has only interior PUs.  An unsharded search of the same PUs under the same
    rectangles (moved to frame coordinates by ext_lo) gives the same results: tests/test_shard_gloo.py."""
    ctu = shard.ctu
    row = (pus["y"] + shard.ext_lo) // ctu
    near_top = (row < shard.ctu_lo + boundary_ctu_rows) if shard.rank > 0 else np.zeros(len(pus), bool)
    near_bot = (row >= shard.ctu_hi - boundary_ctu_rows) if shard.rank < shard.world - 1 else np.zeros(len(pus), bool)
    boundary = near_top | near_bot
    own = (0, shard.top, shard.width, shard.rows)
    ext = (0, 0, shard.width, shard.ext_rows)
    out = [("interior", np.nonzero(~boundary)[0], own)]
    if boundary.any():
        out.append(("boundary", np.nonzero(boundary)[0], ext))
    return out


def cost_checksum(costs):
    """int64 sum of a cost tensor: additive over blocks, hence over shards"""
    return int(costs.long().sum().item()) if costs.numel() else 0


def coeff_checksum(torch, coef):
    """(sum |c|, sum c * w(position)) over int16 [count, n*n] coefficient blocks, w = 1 + position % 251: additive over
    blocks, sensitive to transposed / permuted coefficients"""
    if coef.numel() == 0:
        return 0, 0
    w = (torch.arange(coef.shape[1], device=coef.device, dtype=torch.int64) % 251) + 1
    c = coef.long()
    return int(c.abs().sum().item()), int((c * w).sum().item())
