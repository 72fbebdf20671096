"""Deterministic input patterns shared by tests, bench and the golden generator.
The patterns restate the reference's own test inputs (cited per function)."""
import numpy as np


def rng(seed):
    return np.random.default_rng(seed)


def lcg_bytes(n, seed=12345):
    """x = 1664525*x + 1013904223 (mod 2^32), byte = bits 8..15"""
    out = np.empty(n, dtype=np.uint8)
    # vectorised LCG via jump-ahead is overkill here; chunked python loop is fine for test sizes
    x = seed & 0xFFFFFFFF
    a, c = 1664525, 1013904223
    for i in range(n):
        x = (a * x + c) & 0xFFFFFFFF
        out[i] = (x >> 8) & 0xFF
    return out


def satd_test_bufs(log_w):
    """tests/satd_tests.c:61-90: (black/white, checkers, gradient) buffer pairs for width 1<<log_w"""
    w = 1 << log_w
    size = w * w
    i = np.arange(size)
    bw = (np.zeros(size, np.uint8), np.full(size, 255, np.uint8))
    c0 = (255 * ((((i >> log_w) % 2) + (i % 2)) % 2)).astype(np.uint8)
    c1 = ((c0.astype(np.int32) + 1) % 2).astype(np.uint8)
    col, row = i % w, i // w
    r = np.sqrt(row * row + col * col).astype(np.int64)       # int r = sqrt(...)
    g0 = (255 // (r + 1)).astype(np.uint8)
    g1 = (255 - 255 // (r + 1)).astype(np.uint8)
    return bw, (c0, c1), (g0, g1)


SATD_GOLDEN_BW = {2: 2040, 3: 4080, 4: 16320, 5: 65280, 6: 261120}          # satd_tests.c:109,127
SATD_GOLDEN_GRADIENT = {2: 3140, 3: 9004, 4: 20481, 5: 67262, 6: 258672}    # satd_tests.c:146


def intra_sad_gradient(width):
    """tests/intra_sad_tests.c:48-58 init_gradient(3, 1, width, 1, buf) vs a flat 128 buffer (:91-96)"""
    y, x = np.mgrid[0:width, 0:width]
    val = (np.sqrt((3 - x) ** 2 + (1 - y) ** 2) + 0.5 + 1).astype(np.int64)
    return np.clip(val, 0, 255).astype(np.uint8).ravel(), np.full(width * width, 128, np.uint8)


def sad_test_frames():
    """tests/sad_tests.c:38-58,77-105: the 8x8 pic/ref pair (+48) and the 64x64 big pair"""
    ref = np.array([1, 2, 2, 2, 2, 2, 2, 3] + [4, 5, 5, 5, 5, 5, 5, 6] * 6 + [7, 8, 8, 8, 8, 8, 8, 9],
                   dtype=np.uint8).reshape(8, 8) + 48
    pic = np.ones((8, 8), np.uint8) + 48
    i = np.arange(64 * 64, dtype=np.int64)
    big_pic = ((i * i // 32 + i) % 255).astype(np.uint8).reshape(64, 64)
    big_ref = ((i * i // 16 + i) % 255).astype(np.uint8).reshape(64, 64)
    return pic, ref, big_pic, big_ref


# tests/sad_tests.c:121-259: (mv_x, mv_y) -> closed-form expected kvz_image_calc_sad(pic, ref, 0,0, x,y, 8,8)
SAD_EDGE_KAT = {
    (-3, -3): 1 * 16 + (2 + 4) * 16 + 5 * 16 - 64,
    (0, -3): (1 + 3) * 4 + 2 * 24 + (4 + 6) * 4 + 5 * 24 - 64,
    (3, -3): 3 * 16 + (2 + 6) * 16 + 5 * 16 - 64,
    (-3, 0): (1 + 7) * 4 + 4 * 24 + (2 + 8) * 4 + 5 * 24 - 64,
    (0, 0): (1 + 3 + 7 + 9) + (2 + 4 + 6 + 8) * 6 + 5 * 36 - 64,
    (3, 0): (3 + 9) * 4 + 6 * 24 + (2 + 8) * 4 + 5 * 24 - 64,
    (-3, 3): 7 * 16 + (4 + 8) * 16 + 5 * 16 - 64,
    (0, 3): (7 + 9) * 4 + 8 * 24 + (4 + 6) * 4 + 5 * 24 - 64,
    (3, 3): 9 * 16 + (6 + 8) * 16 + 5 * 16 - 64,
    (-10, -10): 1 * 64 - 64,
    (0, -10): (1 + 3) * 8 + 2 * 48 - 64,
    (10, -10): 3 * 64 - 64,
    (-10, 0): (1 + 7) * 8 + 4 * 48 - 64,
    (10, 0): (3 + 9) * 8 + 6 * 48 - 64,
    (-10, 10): 7 * 64 - 64,
    (0, 10): (7 + 9) * 8 + 8 * 48 - 64,
    (10, 10): 9 * 64 - 64,
}

# tests/sad_tests.c:369-376
REG_SAD_DIMS = [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (32, 64), (32, 16), (16, 32), (16, 8), (8, 16),
                (8, 4), (4, 8), (48, 16), (16, 48), (24, 16), (16, 24), (12, 4), (4, 12)]


def dct_test_input():
    """tests/dct_tests.c:55-79: 64x64 radial gradient init_gradient(64, 64, 64, 255/64, buf); the
    dct of size N reads its first N*N int16."""
    y, x = np.mgrid[0:64, 0:64]
    slope = 255 // 64
    val = (slope * np.sqrt((64 - x) ** 2 + (64 - y) ** 2) + 0.5).astype(np.int64)
    return np.clip(val, 0, 255).astype(np.int16).ravel()


def coeff_sum_input():
    """tests/coeff_sum_tests.c:29-43"""
    c = (np.arange(64 * 64, dtype=np.int64) * 16 - 32768).astype(np.int16)
    expected = 2048 * (16 + 32768) // 2 + 2048 * 2047 * 16 // 2
    return c, expected


def intra_ref_positions(log2_width, color, pic_w, pic_h):
    """every luma position kvz_intra_build_reference can be asked for in a pic_w x pic_h picture: the PU grid of the size
    (chroma PUs of N pixels cover 2N luma pixels; a 4x4 chroma PU belongs to an 8x8 CU, search_intra.c:735-741)"""
    step = (1 << log2_width) << (1 if color else 0)
    return np.array([(x, y) for y in range(0, pic_h - step + 1, step) for x in range(0, pic_w - step + 1, step)], dtype=np.int32)


def intra_ref_cases(log2_width, count, seed):
    """kvz_intra_ref arrays {left[65], top[65]} with left[0] == top[0]: random, flat, ramps, 0/255 extremes"""
    g = rng(seed)
    n = 1 << log2_width
    refs = np.zeros((count, 130), dtype=np.uint8)
    for i in range(count):
        kind = i % 5
        if kind == 0:
            r = g.integers(0, 256, 130)
        elif kind == 1:
            r = np.full(130, g.integers(0, 256))
        elif kind == 2:
            base, sl, st = g.integers(0, 200), g.integers(-3, 4), g.integers(-3, 4)
            r = np.concatenate([base + sl * np.arange(65), base + st * np.arange(65)])
        elif kind == 3:
            r = np.where(g.integers(0, 2, 130) > 0, 255, 0)
        else:
            r = np.clip(128 + np.cumsum(g.integers(-6, 7, 130)), 0, 255)
        refs[i] = np.clip(r, 0, 255)
        refs[i, 65] = refs[i, 0]
        refs[i, 2 * n + 1:65] = 0            # beyond the 2N+1 entries the reference never reads
        refs[i, 65 + 2 * n + 1:] = 0
    return refs


# ---- motion search: layouts of kvz_hip_me_pu / _params / _result (= orc_me_*) ----
ME_PU = np.dtype([("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4"), ("mv_cand", "<i2", (2, 2)),
                  ("extra_mv", "<i2", (2,)), ("num_merge_cand", "<i2"), ("reserved", "<i2"),
                  ("merge", [("mv", "<i2", (2,)), ("usable", "u1"), ("same_ref", "u1")], (5,)), ("pad", "<i2")])
ME_PARAMS = np.dtype([("lambda_cost", "<i4"), ("early_termination", "<i4"), ("max_steps", "<u4"), ("fme_level", "<i4"),
                      ("wpp_owf", "<i4"), ("ref_delay_px", "<i4"), ("max_ref_lcu_down", "<i4"), ("max_ref_lcu_right", "<i4"),
                      ("algorithm", "<i4"), ("search_range", "<i4"), ("size_classes", "<i4"), ("mv_constraint", "<i4"),
                      ("tile_x", "<i4"), ("tile_y", "<i4"), ("tile_w", "<i4"), ("tile_h", "<i4"),
                      ("mv_rdo", "<i4"), ("ref_idx", "<i4"), ("refs_before", "<i4"), ("n_cabac", "<i4"), ("cabac", "<u8"),
                      ("cost_to_beat", "<u8")])
ME_CABAC = np.dtype([("range", "<u2"), ("ctx", "u1", (8,)), ("pad", "u1", (6,))])
ME_RESULT = np.dtype([("mv", "<i4", (2,)), ("cost", "<u4"), ("bitcost", "<u4"), ("merged", "<i4"), ("merge_idx", "<i4"),
                      ("mv_cand", "<i4"), ("reserved", "<i4")])
assert ME_PU.itemsize == 64 and ME_PARAMS.itemsize == 96 and ME_RESULT.itemsize == 32 and ME_CABAC.itemsize == 16
# the search service (kvz_hip_me_request / _service_config / _service_stats)
ME_REQUEST = np.dtype([("pic_slot", "<i4"), ("n_refs", "<i4"), ("ref_slot", "<i4", (16,)), ("cost_to_beat", "<u4"), ("reserved", "<i4"),
                       ("params", ME_PARAMS), ("pu", ME_PU, (16,))])
ME_SERVICE_CONFIG = np.dtype([("width", "<i4"), ("height", "<i4"), ("max_pictures", "<i4"), ("max_threads", "<i4"), ("reserved", "<i4", (4,))])
ME_SERVICE_STATS = np.dtype([("requests", "<u8"), ("units", "<u8"), ("batches", "<u8"), ("launches", "<u8"), ("max_batch_units", "<u8"),
                             ("rects", "<u8"), ("rect_bytes", "<u8"), ("wait_ns", "<u8"), ("tables", "<u8"), ("table_bytes", "<u8"), ("table_ns", "<u8")])
assert ME_REQUEST.itemsize == 1200


def me_params(lambda_cost=20, early_termination=1, max_steps=0xFFFFFFFF, fme_level=4, wpp_owf=0, ref_delay_px=0,
              max_ref_lcu_down=1, max_ref_lcu_right=1, algorithm=0, search_range=0, mv_constraint=0, tile=None,
              mv_rdo=0, ref_idx=0, refs_before=1):
    """tile: (x, y, w, h) of state->tile in the picture, None = the picture is one tile.  mv_rdo: the caller sets p["cabac"] to the
    address of an ME_CABAC array (host for the oracle / reference, device for the GPU entry) and pus["reserved"] to each PU's entry."""
    p = np.zeros(1, dtype=ME_PARAMS)
    p["lambda_cost"], p["early_termination"], p["max_steps"], p["fme_level"] = lambda_cost, early_termination, max_steps, fme_level
    p["wpp_owf"], p["ref_delay_px"], p["max_ref_lcu_down"], p["max_ref_lcu_right"] = wpp_owf, ref_delay_px, max_ref_lcu_down, max_ref_lcu_right
    p["algorithm"] = algorithm
    p["search_range"] = search_range
    p["mv_constraint"] = mv_constraint
    if tile is not None:
        p["tile_x"], p["tile_y"], p["tile_w"], p["tile_h"] = tile
    p["mv_rdo"], p["ref_idx"], p["refs_before"] = mv_rdo, ref_idx, refs_before
    return p


def me_params_from(raw):
    """a stored parameter record (any earlier, shorter layout: the struct only ever grew at its end) -> ME_PARAMS"""
    b = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    p = np.zeros(1, dtype=ME_PARAMS)
    p.view(np.uint8).reshape(-1)[:len(b)] = b[:ME_PARAMS.itemsize]
    if p["refs_before"][0] == 0:
        p["refs_before"] = 1
    return p


def me_cabac_states(count, seed):
    """random but valid CABAC snapshots: range in 256..510, context states 0..62 with either MPS"""
    g = np.random.default_rng(seed)
    c = np.zeros(count, dtype=ME_CABAC)
    c["range"] = g.integers(256, 511, count)
    c["ctx"] = (g.integers(0, 63, (count, 8)) << 1) | g.integers(0, 2, (count, 8))
    c["ctx"][:, 7] = 0
    return c


def me_frames(w, h, seed, motion=(3, -2)):
    """current / reference luma planes: band-limited texture, the reference displaced by `motion` (+ half a pixel of
    blur) and noisy, so the search has a real minimum away from the start vectors"""
    g = np.random.default_rng(seed)
    big = g.integers(0, 256, (h + 64, w + 64)).astype(np.float64)
    for _ in range(2):
        big = (big + np.roll(big, 1, 0) + np.roll(big, 1, 1) + np.roll(big, (1, 1), (0, 1))) / 4
    big = np.clip((big - big.mean()) * 5 + 128, 0, 255)
    cur = big[32:32 + h, 32:32 + w]
    dx, dy = motion
    r = big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w]
    r = (r + np.roll(r, 1, 1)) / 2 + g.normal(0, 2.0, (h, w))
    return np.clip(cur, 0, 255).astype(np.uint8), np.clip(r, 0, 255).astype(np.uint8)


def me_random_pus(w, h, count, seed, hint=None,
                  sizes=((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (64, 32), (24, 32), (32, 8))):
    """hint: a quarter-pel MV (the sequence's true motion) planted as merge / AMVP candidate in a third of the PUs,
    so that the merged and cheap-MVD paths of calc_mvd_cost are taken"""
    g = np.random.default_rng(seed)
    pus = np.zeros(count, dtype=ME_PU)
    for i in range(count):
        bw, bh = sizes[int(g.integers(0, len(sizes)))]
        pus[i]["width"], pus[i]["height"] = bw, bh
        pus[i]["x"] = int(g.integers(0, (w - bw) // 8 + 1)) * 8
        pus[i]["y"] = int(g.integers(0, (h - bh) // 8 + 1)) * 8
        style = i % 4
        rng_mv = (lambda: g.integers(-40, 41, 2)) if style else (lambda: g.integers(-6, 7, 2))
        pus[i]["mv_cand"][0] = rng_mv()
        pus[i]["mv_cand"][1] = pus[i]["mv_cand"][0] if style == 1 else rng_mv()
        pus[i]["extra_mv"] = rng_mv() if style != 2 else (0, 0)
        n = int(g.integers(0, 6))
        pus[i]["num_merge_cand"] = n
        for k in range(n):
            pus[i]["merge"][k]["mv"] = rng_mv() if g.integers(0, 3) else pus[i]["mv_cand"][int(g.integers(0, 2))]
            pus[i]["merge"][k]["usable"] = int(g.integers(0, 4) != 0)
            pus[i]["merge"][k]["same_ref"] = int(g.integers(0, 3) != 0)
        if hint is not None and i % 3 == 0:
            if n and i % 2 == 0:
                k = int(g.integers(0, n))
                pus[i]["merge"][k]["mv"] = hint
                pus[i]["merge"][k]["usable"] = 1
                pus[i]["merge"][k]["same_ref"] = int(i % 4 != 0)
            else:
                pus[i]["mv_cand"][int(g.integers(0, 2))] = hint
    return pus


def me_pus_in_tile(pus, prm):
    """with tiles the reference only ever searches PUs of the tile: move the random PUs into it"""
    tw, th = int(prm["tile_w"][0]), int(prm["tile_h"][0])
    if tw == 0:
        return pus
    tx, ty = int(prm["tile_x"][0]), int(prm["tile_y"][0])
    pus = pus.copy()
    pus["width"] = np.minimum(pus["width"], tw)
    pus["height"] = np.minimum(pus["height"], th)
    pus["x"] = tx + np.minimum(pus["x"] % tw, tw - pus["width"]) // 8 * 8
    pus["y"] = ty + np.minimum(pus["y"] % th, th - pus["height"]) // 8 * 8
    return pus


# ---- SAO ----
def sao_blocks(bw, bh, count, seed):
    """(orig, rec) uint8 [count, bh*bw]: rec = orig + coding-like noise; flat, ramp and 0/255 extreme cases included"""
    g = np.random.default_rng(seed)
    orig = g.integers(0, 256, (count, bh * bw), dtype=np.uint8)
    for i in range(count):
        k = i % 4
        if k == 1:
            yy, xx = np.mgrid[0:bh, 0:bw]
            orig[i] = np.clip(40 + 2 * xx + 3 * yy + g.integers(-3, 4, (bh, bw)), 0, 255).astype(np.uint8).ravel()
        elif k == 2:
            orig[i] = np.where(g.integers(0, 2, bh * bw) > 0, 255, 0)
    rec = np.clip(orig.astype(np.int32) + g.integers(-9, 10, orig.shape), 0, 255).astype(np.uint8)
    rec[::5] = orig[::5]
    return orig, rec


def sao_records(count, seed):
    """int32 [count, 14]: type (1 band / 2 edge), eo_class, band_position[2], offsets[10]"""
    g = np.random.default_rng(seed)
    s = np.zeros((count, 14), dtype=np.int32)
    s[:, 0] = 1 + (np.arange(count) % 2)
    s[:, 1] = g.integers(0, 4, count)
    s[:, 2:4] = g.integers(0, 29, (count, 2))
    s[:, 4:] = g.integers(-7, 8, (count, 10))
    s[:, 4] = 0
    s[:, 9] = 0
    return s


# ---- deblocking (filter.c): a frame with a random CU / PU / TU quadtree and blocky content ----
CU_INFO = np.dtype([("type", "u1"), ("depth", "u1"), ("part_size", "u1"), ("tr_depth", "u1"), ("cbf_y", "u1"), ("mv_dir", "u1"),
                    ("qp", "u1"), ("reserved", "u1"), ("mv", "<i2", (2, 2)), ("mv_ref", "u1", (2,)), ("pad", "u1", (2,))])
DEBLOCK_PARAMS = np.dtype([("beta_offset_div2", "<i4"), ("tc_offset_div2", "<i4"), ("qp", "<i4"), ("frame_qp", "<i4"),
                           ("per_cu_qp", "<i4"), ("slice_is_b", "<i4"), ("chroma", "<i4"), ("reserved", "<i4"), ("ref_LX", "u1", (2, 16))])
assert CU_INFO.itemsize == 20 and DEBLOCK_PARAMS.itemsize == 64


def deblock_params(qp=34, beta=0, tc=0, per_cu_qp=0, slice_is_b=0, chroma=1):
    p = np.zeros(1, dtype=DEBLOCK_PARAMS)
    p["beta_offset_div2"], p["tc_offset_div2"], p["qp"], p["frame_qp"] = beta, tc, qp, qp
    p["per_cu_qp"], p["slice_is_b"], p["chroma"] = per_cu_qp, slice_is_b, chroma
    p["ref_LX"][0, 0, :4] = (0, 1, 2, 3)
    p["ref_LX"][0, 1, :4] = (1, 0, 3, 2)          # L1 index i is L0's picture i ^ 1: the same picture under two names
    return p


def deblock_case(w, h, seed, intra_share=0.35, slice_is_b=0, qp=34):
    """-> (y, u, v planes, cus [h/4, w/4] CU_INFO): random quadtree (CU 64..8, all part modes, TU down to 4x4), vectors from a
    small pool so that neighbours often differ by less than one pixel, blocky pictures with little noise so that the
    on/off decisions, the strong and the weak filter all occur"""
    g = np.random.default_rng(seed)
    cus = np.zeros((h // 4, w // 4), dtype=CU_INFO)
    pool = [(0, 0), (1, 0), (3, -2), (4, 0), (-5, 7), (12, -9), (2, 2), (-3, 1)]

    def leaf(x, y, size, depth):
        intra = g.random() < intra_share
        if intra:
            part = 3 if (size == 8 and g.random() < 0.4) else 0
        else:
            opts = [0, 0, 1, 2] + ([4, 5, 6, 7] if size >= 16 else [])
            part = int(opts[int(g.integers(0, len(opts)))])
        lo = max(depth, 1)
        trd = 4 if (intra and part == 3) else int(g.integers(lo, min(depth + 2, 4 if size == 8 else 3) + 1))
        cu_qp = int(np.clip(qp + g.integers(-6, 7), 10, 51))
        n_parts = (1, 2, 2, 4, 2, 2, 2, 2)[part]
        offs = (((0, 0),), ((0, 0), (0, 2)), ((0, 0), (2, 0)), ((0, 0), (2, 0), (0, 2), (2, 2)), ((0, 0), (0, 1)), ((0, 0), (0, 3)),
                ((0, 0), (1, 0)), ((0, 0), (3, 0)))[part]
        sizes = (((4, 4),), ((4, 2), (4, 2)), ((2, 4), (2, 4)), ((2, 2),) * 4, ((4, 1), (4, 3)), ((4, 3), (4, 1)), ((1, 4), (3, 4)),
                 ((3, 4), (1, 4)))[part]
        for i in range(n_parts):
            px, py = x + offs[i][0] * size // 4, y + offs[i][1] * size // 4
            pw, ph = sizes[i][0] * size // 4, sizes[i][1] * size // 4
            blk = cus[py // 4:min(py + ph, h) // 4, px // 4:min(px + pw, w) // 4]
            blk["type"] = 1 if intra else 2
            blk["depth"], blk["part_size"], blk["tr_depth"], blk["qp"] = depth, part, trd, cu_qp
            if not intra:
                d = int(g.integers(1, 4)) if slice_is_b else 1
                blk["mv_dir"] = d
                blk["mv"][..., 0, :] = pool[int(g.integers(0, len(pool)))]
                blk["mv"][..., 1, :] = pool[int(g.integers(0, len(pool)))]
                blk["mv_ref"][..., 0] = int(g.integers(0, 2))
                blk["mv_ref"][..., 1] = int(g.integers(0, 2))
        # coded-block flags per TU
        tu = 64 >> trd
        for ty in range(y, min(y + size, h), max(tu, 4)):
            for tx in range(x, min(x + size, w), max(tu, 4)):
                cus[ty // 4:min(ty + tu, h) // 4, tx // 4:min(tx + tu, w) // 4]["cbf_y"] = int(g.random() < 0.5)

    def split(x, y, size, depth):
        if x >= w or y >= h:
            return
        must = x + size > w or y + size > h
        if size > 8 and (must or g.random() < (0.85, 0.6, 0.4)[depth]):
            half = size // 2
            for dy in (0, half):
                for dx in (0, half):
                    split(x + dx, y + dy, half, depth + 1)
        else:
            leaf(x, y, size, depth)

    for ly in range(0, h, 64):
        for lx in range(0, w, 64):
            split(lx, ly, 64, 0)

    def plane(pw, ph, cell):
        yy, xx = np.mgrid[0:ph, 0:pw]
        base = 90 + (0.3 * xx + 0.2 * yy) % 100            # a ramp that wraps, so that large frames do not saturate at 255
        steps = g.integers(-7, 8, (ph // cell + 1, pw // cell + 1))
        img = base + steps[yy // cell, xx // cell] + g.integers(-1, 2, (ph, pw))
        rough = g.random((ph // cell + 1, pw // cell + 1)) < 0.15               # some busy cells: the filter must stay off there
        img = np.where(rough[yy // cell, xx // cell], g.integers(0, 256, (ph, pw)), img)
        return np.clip(img, 0, 255).astype(np.uint8)

    return plane(w, h, 8), plane(w // 2, h // 2, 4), plane(w // 2, h // 2, 4), cus


# ---- searches recorded from real encodes (oracle/ref_harness.c recorder; tests/golden/fronts.npz) ----
def fronts_fixture(d, which):
    """-> list of per-frame (pic, ref, pus, results, meta, params): which = "small" (3 P frames of 192x128) or "hd" (one 1080p P frame;
    its planes are rebuilt from the deterministic synthetic sequence + the stored reconstruction delta)"""
    pus = np.ascontiguousarray(d[which + "_pus"]).view(ME_PU).reshape(-1)
    res = np.ascontiguousarray(d[which + "_results"]).view(ME_RESULT).reshape(-1)
    meta = d[which + "_meta"]
    prm0 = me_params_from(d[which + "_params"])
    if which == "small":
        pics, refs = d["small_pic"], d["small_ref"]
    else:
        from ref_lib import synthetic_sequence          # pure numpy
        fr = synthetic_sequence(1920, 1080, 2)
        pics = fr[1:2, :1080]
        refs = (fr[0:1, :1080].astype(np.int16) + d["hd_ref_delta"][None].astype(np.int16)).astype(np.uint8)
    out = []
    for f in range(len(pics)):
        sel = np.where(meta[:, 0] == f)[0]
        prm = prm0.copy()
        prm["lambda_cost"] = meta[sel[0], 4]
        assert (meta[sel, 4] == meta[sel[0], 4]).all()
        out.append((np.ascontiguousarray(pics[f]), np.ascontiguousarray(refs[f]), pus[sel], res[sel], meta[sel], prm))
    return out


def front_groups(meta):
    """Dependency fronts of a frame's recorded searches: LCU (x, y) may start when (x - 1, y) and (x + 1, y - 1) are done (WPP,
    encoderstate.c:807-817) -> wavefront index x + 2y; inside an LCU the searches of the quadtree walk follow each other (seq).
    Returns the list of index arrays, in execution order."""
    key = (meta[:, 1] + 2 * meta[:, 2]).astype(np.int64) * 1000 + meta[:, 3]
    order = np.argsort(key, kind="stable")
    cuts = np.flatnonzero(np.diff(key[order])) + 1
    return np.split(order, cuts)


# ---- AMVP / merge candidate derivation (inter.c:1209-1446): flattened encoder state ----
INTER_PARAMS = np.dtype([("poc", "<i4"), ("slice_is_b", "<i4"), ("tmvp_enable", "<i4"), ("num_refs", "<i4"), ("ref_pocs", "<i4", (16,)),
                         ("ref_LX", "u1", (2, 16)), ("ref_LX_size", "u1", (2,)), ("pad", "u1", (2,)), ("col_ref_pocs", "<i4", (16,)),
                         ("col_ref_LX", "u1", (2, 16)), ("pic_width", "<i4"), ("pic_height", "<i4"), ("in_width", "<i4"),
                         ("in_height", "<i4"), ("tile_x", "<i4"), ("tile_y", "<i4"), ("ref_idx", "<i4"), ("cus_stride", "<i4"),
                         ("col_stride", "<i4"), ("reserved", "<i4")])
MERGE_CAND = np.dtype([("dir", "u1"), ("ref", "u1", (2,)), ("pad", "u1"), ("mv", "<i2", (2, 2))])
assert INTER_PARAMS.itemsize == 252 and MERGE_CAND.itemsize == 12


def inter_params(w, h, poc=8, ref_pocs=(7,), l0=(0,), l1=(), slice_is_b=0, tmvp=1, ref_idx=0, col_ref_pocs=(6,), col_l0=(0,), col_l1=(),
                 tile=(0, 0)):
    """state->frame / state->tile / cfg fields kvz_inter_get_mv_cand and kvz_inter_get_merge_cand read.  ref_pocs = the pictures
    of state->frame->ref, l0 / l1 = state->frame->ref_LX as indices into them; col_* describe the collocated picture
    (ref_LX[0][0]): the POCs of ITS references and its two lists.  SCU maps are ceil(w / 64) * 16 records wide."""
    p = np.zeros(1, dtype=INTER_PARAMS)
    p["poc"], p["slice_is_b"], p["tmvp_enable"], p["num_refs"] = poc, slice_is_b, tmvp, len(ref_pocs)
    p["ref_pocs"][0, :len(ref_pocs)] = ref_pocs
    p["ref_LX"][0, 0, :len(l0)] = l0
    p["ref_LX"][0, 1, :len(l1)] = l1
    p["ref_LX_size"][0] = (len(l0), len(l1))
    p["col_ref_pocs"][0, :len(col_ref_pocs)] = col_ref_pocs
    p["col_ref_LX"][0, 0, :len(col_l0)] = col_l0
    p["col_ref_LX"][0, 1, :len(col_l1)] = col_l1
    p["pic_width"], p["pic_height"], p["in_width"], p["in_height"] = w, h, tile[0] + w, tile[1] + h
    p["tile_x"], p["tile_y"], p["ref_idx"] = tile[0], tile[1], ref_idx
    p["cus_stride"] = p["col_stride"] = ((tile[0] + w + 63) // 64) * 16
    return p


def inter_cu_map(w, h, seed, n_l0=1, n_l1=0, intra_share=0.2, unset_share=0.05):
    """-> (cus [ceil(h/64)*16, ceil(w/64)*16] CU_INFO, PUs): a random quadtree of CUs 64..8 with every inter partition mode; each
    PU carries its own motion (quarter-pel vectors from a small pool so that duplicates and equal AMVP candidates occur, and from a
    wide range so that the POC scaling saturates), reference indices within the two list lengths; some CUs intra, some not set.
    PUs = ME_PU records (x, y, width, height, pad = the barred merge neighbour of a CU's second PU) of every inter PU."""
    g = np.random.default_rng(seed)
    rows, stride = ((h + 63) // 64) * 16, ((w + 63) // 64) * 16
    cus = np.zeros((rows, stride), dtype=CU_INFO)
    pool = [(0, 0), (4, 0), (-5, 7), (12, -9), (-3, 1), (260, -120), (-32768, 32767), (32767, -32768)]
    pus = []

    def leaf(x, y, size, depth):
        r = g.random()
        if r < unset_share:
            return
        intra = r < unset_share + intra_share
        part = 0
        if not intra:
            opts = [0, 0, 1, 2] + ([4, 5, 6, 7] if size >= 16 else [])
            part = int(opts[int(g.integers(0, len(opts)))])
        n_parts = (1, 2, 2, 4, 2, 2, 2, 2)[part]
        offs = (((0, 0),), ((0, 0), (0, 2)), ((0, 0), (2, 0)), ((0, 0), (2, 0), (0, 2), (2, 2)), ((0, 0), (0, 1)), ((0, 0), (0, 3)),
                ((0, 0), (1, 0)), ((0, 0), (3, 0)))[part]
        sizes = (((4, 4),), ((4, 2), (4, 2)), ((2, 4), (2, 4)), ((2, 2),) * 4, ((4, 1), (4, 3)), ((4, 3), (4, 1)), ((1, 4), (3, 4)),
                 ((3, 4), (1, 4)))[part]
        for i in range(n_parts):
            px, py = x + offs[i][0] * size // 4, y + offs[i][1] * size // 4
            pw, ph = sizes[i][0] * size // 4, sizes[i][1] * size // 4
            blk = cus[py // 4:(py + ph) // 4, px // 4:(px + pw) // 4]
            blk["type"] = 1 if intra else 2
            blk["depth"], blk["part_size"] = depth, part
            if intra:
                continue
            d = int(g.integers(1, 4)) if n_l1 else 1
            blk["mv_dir"] = d
            for l, n in ((0, n_l0), (1, n_l1)):
                mv = pool[int(g.integers(0, len(pool)))] if g.random() < 0.6 else tuple(int(v) for v in g.integers(-400, 401, 2))
                blk["mv"][..., l, :] = mv                       # the unused list keeps a vector too: the derivation must ignore it
                blk["mv_ref"][..., l] = int(g.integers(0, max(n, 1)))
            if px + pw <= w and py + ph <= h:
                barred = 0
                if i > 0 and n_parts == 2:
                    barred = 1 if pw < ph else 2            # second PU: A1 barred for the vertical splits, B1 for the horizontal ones
                pus.append((px, py, pw, ph, barred))

    def split(x, y, size, depth):
        if x >= w or y >= h:
            return
        must = x + size > w or y + size > h
        if size > 8 and (must or g.random() < (0.85, 0.6, 0.4)[depth]):
            half = size // 2
            for dy in (0, half):
                for dx in (0, half):
                    split(x + dx, y + dy, half, depth + 1)
        else:
            leaf(x, y, size, depth)

    for ly in range(0, h, 64):
        for lx in range(0, w, 64):
            split(lx, ly, 64, 0)
    out = np.zeros(len(pus), dtype=ME_PU)
    for i, (px, py, pw, ph, barred) in enumerate(pus):
        out[i]["x"], out[i]["y"], out[i]["width"], out[i]["height"], out[i]["pad"] = px, py, pw, ph, barred
    return cus, out


# (name, picture size, inter_params keywords, list lengths of the current and of the collocated picture's CUs)
INTER_CAND_CONFIGS = [
    ("p_one_ref", (168, 136), dict(poc=8, ref_pocs=(7,), l0=(0,), col_ref_pocs=(6,), col_l0=(0,)), (1, 0), (1, 0)),
    ("p_no_tmvp", (128, 64), dict(poc=3, ref_pocs=(2,), l0=(0,), tmvp=0), (1, 0), (1, 0)),
    ("p_second_frame", (72, 72), dict(poc=1, ref_pocs=(0,), l0=(0,), col_ref_pocs=(), col_l0=()), (1, 0), (0, 0)),
    ("p_three_refs", (192, 136), dict(poc=12, ref_pocs=(11, 9, 4), l0=(0, 1, 2), ref_idx=1, col_ref_pocs=(9, 4, 2), col_l0=(0, 1, 2)), (3, 0), (3, 0)),
    ("p_three_refs_far", (136, 128), dict(poc=200, ref_pocs=(199, 60, 2), l0=(0, 1, 2), ref_idx=2, col_ref_pocs=(60, 2), col_l0=(0, 1)), (3, 0), (2, 0)),
    ("b_two_sided", (192, 136), dict(poc=4, ref_pocs=(0, 8), l0=(0, 1), l1=(1, 0), slice_is_b=1, ref_idx=1, col_ref_pocs=(8,), col_l0=(0,), col_l1=(0,)), (2, 2), (1, 1)),
    ("b_lowdelay", (168, 72), dict(poc=9, ref_pocs=(8, 7, 4), l0=(0, 1, 2), l1=(0, 1, 2), slice_is_b=1, ref_idx=0, col_ref_pocs=(7, 4), col_l0=(0, 1), col_l1=(0, 1)), (3, 3), (2, 2)),
    ("b_hier", (136, 136), dict(poc=6, ref_pocs=(4, 8, 0, 16), l0=(0, 2, 1), l1=(1, 3, 0), slice_is_b=1, ref_idx=3, col_ref_pocs=(0, 8), col_l0=(0, 1), col_l1=(1, 0)), (3, 3), (2, 2)),
    ("p_tile", (128, 72), dict(poc=5, ref_pocs=(4,), l0=(0,), col_ref_pocs=(3,), col_l0=(0,), tile=(64, 64)), (1, 0), (1, 0)),
]


def inter_cand_case(name, seed=0):
    """-> (params, cus, col_cus, ref_cus, pus): one configuration of INTER_CAND_CONFIGS with seeded random CU maps"""
    for (n, (w, h), kw, cur_lists, col_lists) in INTER_CAND_CONFIGS:
        if n != name:
            continue
        p = inter_params(w, h, **kw)
        tile = kw.get("tile", (0, 0))
        cus, pus = inter_cu_map(w, h, 9000 + seed, cur_lists[0], cur_lists[1])
        full_w, full_h = tile[0] + w, tile[1] + h
        # a collocated picture without references is an intra picture (a vector in it would make the reference divide by zero)
        col, _ = inter_cu_map(full_w, full_h, 9100 + seed, max(col_lists[0], 1), col_lists[1],
                              intra_share=0.3 if col_lists[0] else 1.0, unset_share=0.1)
        if int(p["ref_idx"][0]) == int(p["ref_LX"][0, 0, 0]):
            refm = col
        else:
            refm, _ = inter_cu_map(full_w, full_h, 9200 + seed, 1, 0, intra_share=0.3)
        # the current picture's map has the stride of the whole picture's like the collocated ones
        wide = np.zeros((cus.shape[0], int(p["cus_stride"][0])), dtype=CU_INFO)
        wide[:, :cus.shape[1]] = cus
        pus["x"] += tile[0]                    # descriptors carry picture coordinates; the CU map of the tile is tile-relative
        pus["y"] += tile[1]
        return p, wide, col, refm, pus
    raise KeyError(name)


def place_lcu_snapshot(cu_map, snap, lcu_x, lcu_y):
    """writes the 290 records of one lcu->cu snapshot (cu.h:324-344: 17 x 17 records = the LCU's 16 x 16 SCUs with the row above,
    the column to the left and the corner at index 0, then the top-right SCU) where they belong in a picture's SCU map"""
    rows, stride = cu_map.shape
    grid = snap[:289].reshape(17, 17)
    oy, ox = lcu_y * 16, lcu_x * 16
    y0, x0 = max(oy - 1, 0), max(ox - 1, 0)
    y1, x1 = min(oy + 16, rows), min(ox + 16, stride)
    cu_map[y0:y1, x0:x1] = grid[y0 - (oy - 1):y1 - (oy - 1), x0 - (ox - 1):x1 - (ox - 1)]
    if oy > 0 and ox + 16 < stride:
        cu_map[oy - 1, ox + 16] = snap[289]


def recorded_cand_fixture(d, derive, seed=5):
    """tests/golden/recorded_cand.npz (oracle/gen_golden.py: recorded_cand): every snapshot's lcu->cu put back into a picture-sized SCU
    map whose other records are noise; `derive(params, cus, col, ref_cus, pus)` completes the descriptor.  -> (derived, recorded)"""
    g = np.random.default_rng(seed)
    meta, want = d["meta"], d["pus"]
    noise = np.zeros(d["snap_col"][0].shape, dtype=CU_INFO)
    noise["type"], noise["mv_dir"] = 2, 1
    noise["mv"] = g.integers(-500, 500, noise["mv"].shape)
    got = []
    for k in range(len(want)):
        f = int(meta[k, 0])
        cus = noise.copy()
        place_lcu_snapshot(cus, d["snap_cus"][k].view(CU_INFO).reshape(-1), int(meta[k, 1]), int(meta[k, 2]))
        pu = np.zeros(1, dtype=ME_PU)
        for fld in ("x", "y", "width", "height"):
            pu[fld] = want[k][fld]
        col = d["snap_col"][f].view(CU_INFO).reshape(noise.shape)
        got.append(derive(d["snap_params"][f:f + 1].view(INTER_PARAMS), cus, col, col, pu))
    return np.concatenate(got), want.view(ME_PU).reshape(-1)


def cost_to_beat_case(integer_costs, seed):
    """per PU *inter_cost values around the integer-stage cost of an unconstrained search (fme_level 0 reports bits * lambda +
    SATD, close to but not the SAD-based integer cost the test is made on): far above, just above, equal, just below, zero,
    and the MAX_INT the first picture of a frame starts from"""
    g = np.random.default_rng(seed)
    c = np.asarray(integer_costs, dtype=np.int64)
    pick = g.integers(0, 6, len(c))
    out = np.where(pick == 0, c * 4 + 1000, np.where(pick == 1, c + g.integers(1, 40, len(c)), np.where(pick == 2, c, np.where(
        pick == 3, np.maximum(c - g.integers(1, 400, len(c)), 0), np.where(pick == 4, 0, 0x7fffffff)))))
    return np.clip(out, 0, 0xffffffff).astype(np.uint32)


# ---- known answers of the reference's candidate helpers: tests/mv_cand_tests.c ----
# test_get_spatial_merge_cand (:26-49): an LCU of inter CUs, PU (x, y, w, h) in a picture (pic_w, pic_h) ->
# indices into lcu_t.cu of b0, b1, b2, a0, a1
MV_CAND_KAT_SPATIAL = ((96, 64, 32, 24, 1920, 1080), (289, 16, 8, 127, 110))
# test_is_a0_cand_coded (:51-133): (x, y, width, height) -> expected
MV_CAND_KAT_A0 = (
    ((32, 64, 16, 16), True), ((32, 64, 32, 16), True), ((32, 64, 32, 8), True), ((32, 64, 32, 24), True),
    ((16, 0, 16, 16), False),
    ((48, 16, 16, 16), False), ((48, 0, 16, 32), False), ((40, 0, 24, 32), False), ((56, 0, 8, 32), False),
    ((32, 16, 16, 16), False), ((32, 8, 32, 24), False), ((32, 24, 32, 8), False),
    ((32, 0, 16, 32), False), ((32, 0, 8, 32), False), ((32, 0, 24, 32), False),
    ((32, 8, 8, 8), True), ((32, 4, 16, 12), True), ((32, 12, 16, 4), True),
    ((32, 0, 8, 16), True), ((32, 0, 4, 16), True), ((32, 0, 12, 16), True),
)
# test_is_b0_cand_coded (:135-213)
MV_CAND_KAT_B0 = (
    ((32, 64, 16, 16), True), ((32, 64, 16, 32), True), ((32, 64, 24, 32), True), ((32, 64, 8, 32), True),
    ((32, 16, 16, 16), True),
    ((48, 16, 16, 16), False), ((32, 16, 32, 16), False), ((32, 8, 32, 24), False), ((32, 24, 32, 8), False),
    ((48, 32, 16, 16), False), ((32, 32, 32, 8), False), ((32, 32, 32, 24), False), ((56, 32, 8, 32), False), ((40, 32, 24, 32), False),
    ((16, 0, 16, 16), True), ((0, 0, 32, 8), True), ((0, 0, 32, 24), True), ((8, 0, 24, 32), True), ((24, 0, 8, 32), True),
)


def valid_pu_geometries(pic=192):
    """every PU (x, y, w, h) of every partition mode (2Nx2N, 2NxN, Nx2N, NxN, the four AMP modes from 16x16 up) of every CU
    of 8x8 .. 64x64 in a pic x pic picture"""
    out = []
    for n in (8, 16, 32, 64):
        q = n // 4
        parts = [(0, 0, n, n), (0, 0, n, n // 2), (0, n // 2, n, n // 2), (0, 0, n // 2, n), (n // 2, 0, n // 2, n)]
        parts += [(dx, dy, n // 2, n // 2) for dy in (0, n // 2) for dx in (0, n // 2)]
        if n >= 16:
            parts += [(0, 0, n, q), (0, q, n, n - q), (0, 0, n, n - q), (0, n - q, n, q),
                      (0, 0, q, n), (q, 0, n - q, n), (0, 0, n - q, n), (n - q, 0, q, n)]
        for cy in range(0, pic, n):
            for cx in range(0, pic, n):
                out += [(cx + dx, cy + dy, w, h) for (dx, dy, w, h) in parts]
    return np.array(out, dtype=np.int32)


def mv_cand_unique_map_case(pic=192):
    """-> (params, cus, pus): a pic x pic P picture whose 4x4 units are all inter with a vector that names the unit (x / 4, y / 4),
    no temporal candidates, and every PU of valid_pu_geometries(pic): which neighbours a derivation used can be read off the
    vectors of its merge list"""
    p = inter_params(pic, pic, tmvp=0)
    n = pic // 4
    cus = np.zeros((n, int(p["cus_stride"][0])), dtype=CU_INFO)
    cus["type"], cus["mv_dir"] = 2, 1
    ys, xs = np.mgrid[0:n, 0:cus.shape[1]]
    cus["mv"][:, :, 0, 0], cus["mv"][:, :, 0, 1] = xs, ys
    geoms = valid_pu_geometries(pic)
    pus = np.zeros(len(geoms), dtype=ME_PU)
    pus["x"], pus["y"], pus["width"], pus["height"] = geoms[:, 0], geoms[:, 1], geoms[:, 2], geoms[:, 3]
    return p, cus, pus


def check_unique_map_merge_lists(d, pus, merge):
    """merge: MERGE_CAND records [n, 5]; d: mv_cand.npz"""
    merge = np.ascontiguousarray(merge).view(MERGE_CAND).reshape(len(pus), 5)
    for i in range(0, len(pus), 7):
        x, y, w, h = (int(pus[i][k]) for k in ("x", "y", "width", "height"))
        b0i, b1i, b2i, a0i, a1i = (int(v) for v in d["idx"][i])
        place = dict(a1=(x - 1, y + h - 1), b1=(x + w - 1, y - 1), b0=(x + w, y - 1), a0=(x - 1, y + h), b2=(x - 1, y - 1))
        have = dict(a1=a1i >= 0, b1=b1i >= 0, b0=b0i >= 0, a0=a0i >= 0, b2=b2i >= 0)
        want = []
        for k in ("a1", "b1", "b0", "a0", "b2"):
            if have[k] and not (k == "b2" and len(want) == 4):
                v = (place[k][0] // 4, place[k][1] // 4)
                if v not in want:                              # duplicates are pruned; with unique vectors only a repeated unit repeats
                    want.append(v)
        got = [tuple(int(c) for c in merge[i, k]["mv"][0]) for k in range(min(len(want), 5))]
        assert got == want[:5], "PU %s: merge list %s, places %s" % ((x, y, w, h), got, want)


# ---- tests/inter_recon_bipred_tests.c:32-121: the blend of two 16x16 predictions at (0, 0) of an LCU, both vectors (3, 3)
# (fractional in luma and chroma: all four sources are 14-bit samples), against the test's own plain restatement
# (s0 + s1 + offset) >> shift through kvz_fast_clip_32bit_to_pixel ----
BIPRED_TEST_GEOMETRY = dict(width=16, height=16, xpos=0, ypos=0, mv=((3, 3), (3, 3)))
# the test's case first (zero-initialised buffers, as its static arrays are), then seeded ones: (seed, w, h, x, y, (hi luma 0, 1, chroma 0, 1))
BIPRED_CASES = ((None, 16, 16, 0, 0, (1, 1, 1, 1)), (1, 16, 16, 0, 0, (1, 1, 1, 1)), (2, 16, 16, 0, 0, (0, 1, 0, 1)), (3, 16, 16, 0, 0, (1, 0, 1, 0)),
                (4, 16, 16, 0, 0, (0, 0, 0, 0)), (5, 8, 8, 24, 40, (1, 1, 1, 1)), (6, 64, 64, 0, 0, (1, 0, 0, 1)), (7, 32, 16, 96, 72, (0, 1, 1, 0)))


def bipred_case_inputs(seed):
    """-> (hp0, hp1, rec, tmp): three planes each (4096, 1024, 1024 samples), 14-bit int16 samples / pixels.  seed None: all zero,
    what the reference's test runs on; else values over the whole range a 14-bit interpolation produces, clipping both ways"""
    if seed is None:
        z16 = [np.zeros(n, np.int16) for n in (4096, 1024, 1024)]
        z8 = [np.zeros(n, np.uint8) for n in (4096, 1024, 1024)]
        return z16, [a.copy() for a in z16], z8, [a.copy() for a in z8]
    g = np.random.default_rng(8800 + seed)
    hp0 = [g.integers(-3000, 20000, n).astype(np.int16) for n in (4096, 1024, 1024)]
    hp1 = [g.integers(-3000, 20000, n).astype(np.int16) for n in (4096, 1024, 1024)]
    rec = [g.integers(0, 256, n, dtype=np.uint8) for n in (4096, 1024, 1024)]
    tmp = [g.integers(0, 256, n, dtype=np.uint8) for n in (4096, 1024, 1024)]
    return hp0, hp1, rec, tmp


def bipred_expected(hi, w, h, x, y, hp0, hp1, rec, tmp):
    """the test file's restatement (inter_recon_bipred_tests.c:74-121) in numpy: source 0 = hi-prec buffer 0 or the temporary
    LCU's pixels << 6, source 1 = hi-prec buffer 1 or the pixels already in rec << 6; (s0 + s1 + 64) >> 7, clipped to 0..255"""
    out = [a.copy() for a in rec]
    for plane, (stride, sh) in enumerate(((64, 0), (32, 1), (32, 1))):
        hi0, hi1 = (hi[0], hi[1]) if plane == 0 else (hi[2], hi[3])
        for ty in range(h >> sh):
            yy = ((y >> sh) + ty) & (stride - 1)
            for tx in range(w >> sh):
                xx = ((x >> sh) + tx) & (stride - 1)
                i = yy * stride + xx
                s0 = int(hp0[plane][i]) if hi0 else int(np.int16(int(tmp[plane][i]) << 6))
                s1 = int(hp1[plane][i]) if hi1 else int(np.int16(int(out[plane][i]) << 6))
                out[plane][i] = min(255, max(0, (s0 + s1 + 64) >> 7))
    return out


def bipred_case_blocks(k):
    """BIPRED_CASES[k] as contiguous blocks per plane for the plane-wise blend entries: -> list of (w, h, hi0, s0, hi1, s1, (rows, cols))
    for Y, U, V; s = int16 samples when hi, else pixels; (rows, cols) = where the block lies in the LCU plane"""
    seed, w, h, x, y, hi = BIPRED_CASES[k]
    hp0, hp1, rec, tmp = bipred_case_inputs(seed)
    out = []
    for plane, (stride, sh) in enumerate(((64, 0), (32, 1), (32, 1))):
        hi0, hi1 = (hi[0], hi[1]) if plane == 0 else (hi[2], hi[3])
        bw, bh, bx, by = w >> sh, h >> sh, (x >> sh) & (stride - 1), (y >> sh) & (stride - 1)
        rows, cols = slice(by, by + bh), slice(bx, bx + bw)
        s0 = (hp0[plane] if hi0 else tmp[plane]).reshape(stride, stride)[rows, cols]
        s1 = (hp1[plane] if hi1 else rec[plane]).reshape(stride, stride)[rows, cols]
        out.append((bw, bh, hi0, np.ascontiguousarray(s0), hi1, np.ascontiguousarray(s1), (rows, cols)))
    return out


# ---- the reference's own benchmark workload: tests/speed_tests.c ----
SPEED_NUM_TESTS, SPEED_NUM_CHUNKS = 113, 36          # speed_tests.c:33-34


def speed_test_bufs():
    """setup_tests (speed_tests.c:77-92): bufs[test][chunk] = 64x64 radial gradient init_gradient(64 - x, y, 64, 255 / 64, .) (:63-74) with
    x = (test + chunk) % 64, y = (test + chunk) / 64 -> uint8 [113, 36 * 4096]"""
    yy, xx = np.mgrid[0:64, 0:64]
    out = np.zeros((SPEED_NUM_TESTS, SPEED_NUM_CHUNKS, 4096), dtype=np.uint8)
    slope = 255 // 64
    for test in range(SPEED_NUM_TESTS):
        for chunk in range(SPEED_NUM_CHUNKS):
            x, y = (test + chunk) % 64, (test + chunk) // 64
            val = (slope * np.sqrt(((64 - x) - xx) ** 2 + (y - yy) ** 2) + 0.5).astype(np.int64)
            out[test, chunk] = np.clip(val, 0, 255).astype(np.uint8).ravel()
    return out.reshape(SPEED_NUM_TESTS, SPEED_NUM_CHUNKS * 4096)


def speed_test_intra_pairs(bufs, n):
    """test_intra_speed (speed_tests.c:116-153) for one pass over all 113 tests: the first chunk of every group of 36 n x n chunks against
    the 35 others -> (blk1, blk2) uint8 [113 * (4096 / n^2) * 35, n * n]"""
    size = n * n
    groups = bufs.reshape(SPEED_NUM_TESTS, 4096 // size, SPEED_NUM_CHUNKS, size)
    b1 = np.repeat(groups[:, :, :1, :], SPEED_NUM_CHUNKS - 1, axis=2)
    b2 = groups[:, :, 1:, :]
    return np.ascontiguousarray(b1.reshape(-1, size)), np.ascontiguousarray(b2.reshape(-1, size))


def speed_test_dct_residuals(bufs, n):
    """dct_speed (speed_tests.c:252-300): residual = first chunk - chunk, all 36 chunks of every group -> int16 [113 * (4096 / n^2) * 36, n * n]"""
    size = n * n
    groups = bufs.reshape(SPEED_NUM_TESTS, 4096 // size, SPEED_NUM_CHUNKS, size).astype(np.int16)
    return np.ascontiguousarray((groups[:, :, :1, :] - groups).reshape(-1, size))


def speed_test_inter_frame(w=3840, h=2160):
    """setup_tests (speed_tests.c:94-103): inter_a, the 4K luma plane the reg_sad benchmark runs on (unsigned 32-bit i * i)"""
    i = np.arange(w * h, dtype=np.uint64)
    sq = (i * i) & np.uint64(0xffffffff)
    pattern1 = (((sq >> np.uint64(10)) % np.uint64(255)) >> np.uint64(2)).astype(np.uint8)
    gradient = (((i >> np.uint64(12)) + i) & np.uint64(255)).astype(np.uint8)
    return ((pattern1.astype(np.int32) + gradient.astype(np.int32)) % 255).astype(np.uint8).reshape(h, w)


def speed_test_inter_pairs(bw, bh, iterations, w=3840, h=2160):
    """test_inter_speed (speed_tests.c:198-236): iteration i takes LCU (1 + i % (w/64 - 2), 1 + (i / (h/64 - 2)) % (h/64 - 2)) and the 25
    vectors {-6, -3, 0, 3, 6}^2, both blocks in the same frame -> kvz_hip_block_pair rows (x1, y1, x2, y2, bw, bh) int32 [25 * iterations, 6]"""
    dx, dy = w // 64 - 2, h // 64 - 2
    i = np.arange(iterations, dtype=np.int64)
    lx, ly = 1 + i % dx, 1 + (i // dy) % dy
    mv = np.array([(mx, my) for my in range(-6, 7, 3) for mx in range(-6, 7, 3)], dtype=np.int64)
    x1 = np.repeat(lx * 64, 25); y1 = np.repeat(ly * 64, 25)
    x2 = x1 + np.tile(mv[:, 0], iterations); y2 = y1 + np.tile(mv[:, 1], iterations)
    out = np.stack([x1, y1, x2, y2, np.full_like(x1, bw), np.full_like(x1, bh)], axis=1)
    return np.ascontiguousarray(out.astype(np.int32))
