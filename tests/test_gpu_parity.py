"""GPU parity: every batched entry of the C ABI (HIP kernels on a real MI355X)
against the oracle on the same seeded inputs -- bit-exact, all integer work.
Includes the reference's own test patterns and known-answer values."""
import numpy as np
import pytest

import oracle_lib as O
from patterns import (SAD_EDGE_KAT, SATD_GOLDEN_BW, SATD_GOLDEN_GRADIENT, REG_SAD_DIMS, coeff_sum_input,
                      dct_test_input, intra_sad_gradient, rng, sad_test_frames, satd_test_bufs)

pytestmark = pytest.mark.gpu

SIZES = (4, 8, 16, 32, 64)


@pytest.fixture(scope="module")
def api():
    from kvazaar_amd import api as a, _lib
    _lib.init(0)
    return a


def _blocks(n, count, seed, mode):
    g = rng(seed)
    if mode == "random":
        a = g.integers(0, 256, (count, n * n), dtype=np.uint8)
        b = g.integers(0, 256, (count, n * n), dtype=np.uint8)
    elif mode == "extreme":
        a = np.zeros((count, n * n), np.uint8)
        b = np.full((count, n * n), 255, np.uint8)
        a[1::2], b[1::2] = 255, 0
    else:
        a = g.integers(0, 256, (count, n * n), dtype=np.uint8)
        b = np.clip(a.astype(np.int32) + g.integers(-6, 7, a.shape), 0, 255).astype(np.uint8)
    return a, b


# ragged counts exercise the wave / group tails of every kernel
@pytest.mark.parametrize("count", [1, 3, 63, 64, 65, 257, 1000])
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kind", ["sad", "satd"])
def test_cost_nxn_counts(api, kind, n, count):
    if n == 64 and count > 300:
        count = 300
    a, b = _blocks(n, count, 1000 + n + count, "random")
    np.testing.assert_array_equal(api.cost_nxn_batch(kind, n, a, b), O.cost_nxn_batch(kind, n, a, b))


@pytest.mark.parametrize("mode", ["extreme", "near"])
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kind", ["sad", "satd"])
def test_cost_nxn_modes(api, kind, n, mode):
    a, b = _blocks(n, 130, 7 + n, mode)
    np.testing.assert_array_equal(api.cost_nxn_batch(kind, n, a, b), O.cost_nxn_batch(kind, n, a, b))


def test_cost_nxn_empty(api):
    assert api.cost_nxn_batch("sad", 8, np.zeros((0, 64), np.uint8), np.zeros((0, 64), np.uint8)).shape == (0,)


@pytest.mark.parametrize("log_w", [2, 3, 4, 5, 6])
def test_satd_reference_known_answers(api, log_w):
    """tests/satd_tests.c:109,127,146 hard-coded golden values; symmetric in the arguments"""
    n = 1 << log_w
    bw, ck, gr = satd_test_bufs(log_w)
    for (x, y), want in ((bw, SATD_GOLDEN_BW[log_w]), (ck, SATD_GOLDEN_BW[log_w]), (gr, SATD_GOLDEN_GRADIENT[log_w])):
        assert api.cost_nxn_batch("satd", n, x[None], y[None])[0] == want
        assert api.cost_nxn_batch("satd", n, y[None], x[None])[0] == want


@pytest.mark.parametrize("log_w", [2, 3, 4, 5, 6])
def test_intra_sad_reference_patterns(api, log_w):
    """tests/intra_sad_tests.c:124-167"""
    n = 1 << log_w
    z, m = np.zeros(n * n, np.uint8), np.full(n * n, 255, np.uint8)
    assert api.cost_nxn_batch("sad", n, z[None], m[None])[0] == 255 * n * n
    assert api.cost_nxn_batch("sad", n, m[None], z[None])[0] == 255 * n * n
    ga, gb = intra_sad_gradient(n)
    want = int(np.abs(ga.astype(np.int64) - gb.astype(np.int64)).sum())
    assert api.cost_nxn_batch("sad", n, ga[None], gb[None])[0] == want
    assert api.cost_nxn_batch("sad", n, gb[None], ga[None])[0] == want


@pytest.mark.parametrize("n", [4, 8, 16, 32])
@pytest.mark.parametrize("kind", ["sad", "satd"])
def test_cost_nxn_dual(api, kind, n):
    g = rng(70 + n)
    for count in (1, 5, 64, 129):
        orig = g.integers(0, 256, (count, n * n), dtype=np.uint8)
        preds = g.integers(0, 256, (count, 2048), dtype=np.uint8)
        np.testing.assert_array_equal(api.cost_nxn_dual_batch(kind, n, preds, orig),
                                      O.cost_nxn_dual_batch(kind, n, preds, orig))


def test_reg_sad_reference_shapes(api):
    """tests/sad_tests.c:261-320,369-376: 18 (w,h) shapes on the 64x64 patterns + 0-vs-255 overflow"""
    _, _, big_pic, big_ref = sad_test_frames()
    dims = REG_SAD_DIMS + [(64, 63), (1, 1), (7, 3), (13, 5)]
    pairs = [(0, 0, 0, 0, w, h) for (w, h) in dims]
    got = api.reg_sad_batch(big_pic, big_ref, pairs)
    for (w, h), v in zip(dims, got):
        want = int(np.abs(big_pic[:h, :w].astype(np.int64) - big_ref[:h, :w].astype(np.int64)).sum())
        assert v == want, (w, h)
    z, m = np.zeros((64, 64), np.uint8), np.full((64, 64), 255, np.uint8)
    got = api.reg_sad_batch(z, m, pairs)
    for (w, h), v in zip(dims, got):
        assert v == 255 * w * h


def test_reg_sad_random_offsets(api):
    g = rng(11)
    p1 = g.integers(0, 256, (120, 200), dtype=np.uint8)
    p2 = g.integers(0, 256, (90, 160), dtype=np.uint8)
    pairs = []
    for _ in range(300):
        w, h = int(g.integers(1, 65)), int(g.integers(1, 65))
        pairs.append((int(g.integers(0, 200 - w + 1)), int(g.integers(0, 120 - h + 1)),
                      int(g.integers(0, 160 - w + 1)), int(g.integers(0, 90 - h + 1)), w, h))
    got = api.reg_sad_batch(p1, p2, pairs)
    for (x1, y1, x2, y2, w, h), v in zip(pairs, got):
        assert v == O.reg_sad(p1, p2, y1 * 200 + x1, y2 * 160 + x2, w, h, 200, 160)


def test_image_calc_sad_reference_known_answers(api):
    """tests/sad_tests.c:121-259: 17 closed-form values for MVs overlapping / outside the frame"""
    pic, ref, _, _ = sad_test_frames()
    mvs = list(SAD_EDGE_KAT.keys())
    got = api.image_calc_sad_batch(pic, ref, [(0, 0, x, y, 8, 8) for (x, y) in mvs])
    for mv, v in zip(mvs, got):
        assert v == SAD_EDGE_KAT[mv], mv


def test_image_calc_sad_and_satd_edges(api):
    g = rng(8)
    pic = g.integers(0, 256, (48, 64), dtype=np.uint8)
    ref = g.integers(0, 256, (48, 64), dtype=np.uint8)
    pairs = []
    for (bw, bh) in ((8, 8), (16, 16), (16, 8), (32, 32), (64, 48), (4, 4), (12, 16), (8, 12), (24, 32)):
        for (px, py) in ((0, 0), (min(16, 64 - bw), min(8, 48 - bh)), (64 - bw, 48 - bh)):
            for (dx, dy) in ((0, 0), (-3, -3), (5, -70), (-100, 2), (70, 70), (3, 0), (0, 60), (-bw, -bh), (64, 48),
                             (63 - px, 47 - py)):
                pairs.append((px, py, px + dx, py + dy, bw, bh))
    sad = api.image_calc_sad_batch(pic, ref, pairs)
    satd = api.image_calc_satd_batch(pic, ref, pairs)
    for p, a, b in zip(pairs, sad, satd):
        assert a == O.image_calc("sad", pic, ref, *p), p
        assert b == O.image_calc("satd", pic, ref, *p), p


@pytest.mark.parametrize("threads", [128, 256, 512])
def test_image_calc_satd_chunks_of_8x8_pairs(api, threads):
    """the grid-stride descriptor kernel takes 64 descriptors per workgroup: chunks of nothing but 8x8 pairs are scored one pair
    per lane (also across the frame's edges), a single other size in a chunk sends the whole chunk down the general split, the
    last chunk is ragged -- for each workgroup size of the tuning knob"""
    from kvazaar_amd import _lib
    L = _lib.init(0)
    g = rng(81)
    pic = g.integers(0, 256, (96, 160), dtype=np.uint8)
    ref = g.integers(0, 256, (96, 160), dtype=np.uint8)
    pairs = []
    for i in range(4096 + 64 * 9 + 37):                           # beyond the one-wave-per-descriptor limit; ragged tail
        w, h = 8, 8
        if 640 <= i < 704 and i % 7 == 0:
            w, h = ((16, 16), (8, 16), (4, 4), (64, 64))[i % 4]    # one chunk with strangers in it
        if 1024 <= i < 1088:
            w, h = 16, 8                                          # one chunk without a single 8x8 pair
        x1, y1 = int(g.integers(0, 160 - w + 1)), int(g.integers(0, 96 - h + 1))
        dx, dy = (int(g.integers(-200, 200)), int(g.integers(-120, 120))) if i % 5 == 0 else (int(g.integers(-6, 7)), int(g.integers(-6, 7)))
        pairs.append((x1, y1, x1 + dx, y1 + dy, w, h))
    want = [O.image_calc("satd", pic, ref, *p) for p in pairs]
    assert L.kvz_hip_set_tuning(b"pair_satd_threads", threads) == 0
    try:
        np.testing.assert_array_equal(api.image_calc_satd_batch(pic, ref, pairs), want)
    finally:
        L.kvz_hip_set_tuning(b"pair_satd_threads", -1)


def test_pixels_calc_ssd(api):
    g = rng(6)
    a = g.integers(0, 256, (70, 70), dtype=np.uint8)
    b = g.integers(0, 256, (70, 66), dtype=np.uint8)
    pairs = [(2, 1, 1, 3, w, w) for w in (4, 8, 16, 32, 64)]
    got = api.pixels_calc_ssd_batch(a, b, pairs)
    for (x1, y1, x2, y2, w, _), v in zip(pairs, got):
        assert v == O.pixels_calc_ssd(a, y1 * 70 + x1, b, y2 * 66 + x2, 70, 66, w)
    z, m = np.zeros((64, 64), np.uint8), np.full((64, 64), 255, np.uint8)
    assert api.pixels_calc_ssd_batch(z, m, [(0, 0, 0, 0, 64, 64)])[0] == 64 * 64 * 255 * 255


def test_satd_any_size_quad_incl_quirk(api):
    g = rng(5)
    orig = g.integers(0, 256, (80, 100), dtype=np.uint8)
    dims = [(w, h) for w in (4, 8, 12, 16, 24, 32, 64) for h in (4, 8, 12, 16, 24, 32, 64)]
    preds = g.integers(0, 256, (len(dims) * 4, 64 * 64), dtype=np.uint8)
    pairs = [(11, 3, 0, 0, w, h) for (w, h) in dims]
    got = api.satd_any_size_quad_batch(preds, orig, pairs)
    for i, (w, h) in enumerate(dims):
        want = O.satd_any_size_quad(w, h, [preds[4 * i + k] for k in range(4)], 64, orig, 3 * 100 + 11, 100)
        np.testing.assert_array_equal(got[i], want, err_msg="w=%d h=%d" % (w, h))


def test_bipred_blend(api):
    g = rng(12)
    for (w, h) in ((8, 8), (16, 4), (64, 64), (5, 3)):
        hp0 = g.integers(-3000, 20000, (3, h, w)).astype(np.int16)
        hp1 = g.integers(-3000, 20000, (3, h, w)).astype(np.int16)
        px0 = g.integers(0, 256, (3, h, w), dtype=np.uint8)
        px1 = g.integers(0, 256, (3, h, w), dtype=np.uint8)
        for hi0, s0 in ((1, hp0), (0, px0)):
            for hi1, s1 in ((1, hp1), (0, px1)):
                got = api.bipred_blend_batch(w, h, hi0, s0, hi1, s1)
                for k in range(3):
                    np.testing.assert_array_equal(got[k], O.bipred_blend_plane(w, h, hi0, s0[k], hi1, s1[k]))


# ------------------------------------------------------------------ dct
@pytest.mark.parametrize("kind,n", [("dct", 4), ("dct", 8), ("dct", 16), ("dct", 32), ("idct", 4), ("idct", 8),
                                     ("idct", 16), ("idct", 32), ("dst", 4), ("idst", 4)])
def test_transform(api, kind, n):
    g = rng(20 + n)
    for count in (1, 7, 64, 130):
        res = g.integers(-255, 256, (count, n * n)).astype(np.int16)
        np.testing.assert_array_equal(api.transform_batch(kind, n, res), O.transform_batch(kind, n, res))
    full = g.integers(-32768, 32768, (40, n * n)).astype(np.int16)          # wrap (forward) / clip (inverse)
    np.testing.assert_array_equal(api.transform_batch(kind, n, full), O.transform_batch(kind, n, full))
    edge = np.array([[32767] * (n * n), [-32768] * (n * n),
                     [32767 if (i + i // n) % 2 else -32768 for i in range(n * n)]], dtype=np.int16)
    np.testing.assert_array_equal(api.transform_batch(kind, n, edge), O.transform_batch(kind, n, edge))


def test_transform_reference_pattern(api):
    """tests/dct_tests.c:55-175: radial gradient, expected = generic output (here: the oracle, pinned to generic)"""
    src = dct_test_input()
    for n in (4, 8, 16, 32):
        x = src[:n * n][None]
        for kind in ("dct", "idct") + (("dst", "idst") if n == 4 else ()):
            np.testing.assert_array_equal(api.transform_batch(kind, n, x), O.transform_batch(kind, n, x))


# ------------------------------------------------------------------ quant
@pytest.mark.parametrize("w", [4, 8, 16, 32])
@pytest.mark.parametrize("signhide", [0, 1])
def test_quant(api, w, signhide):
    g = rng(30 + w)
    coef = g.integers(-2000, 2001, (70, w * w)).astype(np.int16)
    coef[3] = g.integers(-32768, 32768, w * w)
    coef[4] = 0
    coef[5, ::7] = 1
    for qp in (0, 22, 37, 51):
        for type_ in ((0,) if w == 32 else (0, 2)):
            for scan in (0, 1, 2):
                for intra_slice in (0, 1):
                    got = api.quant_batch(coef, w, qp, type_, scan, intra_slice, signhide)
                    want = O.quant_batch(coef, w, qp, type_, scan, intra_slice, signhide)
                    np.testing.assert_array_equal(got, want, err_msg="qp=%d type=%d scan=%d" % (qp, type_, scan))


@pytest.mark.parametrize("w", [4, 8, 16, 32])
def test_dequant(api, w):
    g = rng(40 + w)
    q = g.integers(-300, 301, (33, w * w)).astype(np.int16)
    q[2] = g.integers(-32768, 32768, w * w)
    for qp in (0, 7, 22, 36, 51):
        for type_ in ((0,) if w == 32 else (0, 2, 3)):
            np.testing.assert_array_equal(api.dequant_batch(q, w, qp, type_), O.dequant_batch(q, w, qp, type_))


@pytest.mark.parametrize("w", [4, 8, 16, 32])
def test_quant_dequant_scaling_tables(api, w):
    """per-coefficient factor tables (the scaling-list path) with synthetic tables"""
    g = rng(45 + w)
    coef = g.integers(-3000, 3001, (9, w * w)).astype(np.int16)
    qt = (g.integers(800, 30000, w * w)).astype(np.int32)
    dt = (g.integers(16, 2000, w * w)).astype(np.int32)
    for qp in (4, 22, 40, 51):
        got = api.quant_batch(coef, w, qp, 0, 0, 0, 0, quant_coeff=qt)
        want = O.quant_batch(coef, w, qp, 0, 0, 0, 0, quant_coeff=qt)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(api.dequant_batch(want, w, qp, 0, dequant_coeff=dt),
                                      O.dequant_batch(want, w, qp, 0, dequant_coeff=dt))


def test_coeff_abs_sum(api):
    c, expected = coeff_sum_input()        # tests/coeff_sum_tests.c:29-43
    assert api.coeff_abs_sum_batch(c, 64 * 64)[0] == expected
    g = rng(3)
    x = g.integers(-32768, 32768, (37, 256)).astype(np.int16)
    got = api.coeff_abs_sum_batch(x, 256)
    for i in range(37):
        assert got[i] == O.coeff_abs_sum(x[i])


@pytest.mark.parametrize("w", [4, 8, 16, 32])
def test_quantize_residual(api, w):
    g = rng(50 + w)
    ref_in = g.integers(0, 256, (70, w * w), dtype=np.uint8)
    pred = np.clip(ref_in.astype(np.int32) + g.integers(-40, 41, ref_in.shape), 0, 255).astype(np.uint8)
    pred[0] = ref_in[0]
    pred[1] = 255 - ref_in[1]
    for qp in (12, 22, 32, 45):
        for color in ((0,) if w == 32 else (0, 1, 2)):
            for intra in (0, 1):
                for signhide in (0, 1):
                    for trskip in ((0, 1) if w == 4 else (0,)):
                        got = api.quantize_residual_batch(ref_in, pred, w, qp, color, 0, intra, intra, signhide, trskip)
                        want = O.quantize_residual_batch(ref_in, pred, w, qp, color, 0, intra, intra, signhide, trskip)
                        for a, b, nm in zip(got, want, ("rec", "coeff", "has")):
                            np.testing.assert_array_equal(a, b, err_msg="%s qp=%d color=%d intra=%d sh=%d ts=%d" %
                                                          (nm, qp, color, intra, signhide, trskip))
    # rec_out aliasing pred_in (transform.c:398-399)
    got = api.quantize_residual_batch(ref_in, pred, w, 27, 0, 0, 0, alias_rec=True)
    want = O.quantize_residual_batch(ref_in, pred, w, 27, 0, 0, 0)
    np.testing.assert_array_equal(got[0], want[0])


@pytest.mark.parametrize("n", [8, 16, 32])
def test_quantize_residual_fast_and_lds_kernels(api, n):
    """8x8 TUs run in registers (8 per wave), 16x16 / 32x32 on the matrix cores (four 16x16 TUs or one 32x32 per MFMA tile) by
    default; the LDS butterfly kernel stays selectable.  Counts that leave the last wave step / tile partly empty, TUs with and
    without coefficients inside one step, both rd=0 cost outputs, luma and chroma QP mapping."""
    from kvazaar_amd import _lib
    L = _lib.init(0)
    g = rng(160 + n)
    key = b"qr8_reg_kernel" if n == 8 else b"qr_tile_kernel"
    try:
        for count in (1, 2, 3, 5, 7, 9, 64, 67):
            ref_in = g.integers(0, 256, (count, n * n), dtype=np.uint8)
            pred = np.clip(ref_in.astype(np.int32) + g.integers(-50, 51, ref_in.shape), 0, 255).astype(np.uint8)
            pred[::2] = ref_in[::2]                              # every other TU quantises to nothing
            if count > 4:
                pred[3] = 255 - ref_in[3]                        # extreme residuals
            for qp, color in ((10, 0), (27, 0), (44, 0), (33, 1), (51, 2)):
                want = O.quantize_residual_batch(ref_in, pred, n, qp, color, 0, 0)
                for use in (1, 0):
                    _lib.check(L.kvz_hip_set_tuning(key, use), "tuning")
                    got = api.quantize_residual_batch(ref_in, pred, n, qp, color, 0, 0, with_costs=True)
                    for a, b, nm in zip(got[:3], want, ("rec", "coeff", "has")):
                        np.testing.assert_array_equal(a, b, err_msg="%s n=%d count=%d qp=%d fast=%d" % (nm, n, count, qp, use))
                    for i in range(count):
                        assert got[3][i] == O.pixels_calc_ssd(ref_in[i], 0, want[0][i], 0, n, n, n)
                        assert got[4][i] == O.coeff_abs_sum(want[1][i])
                    # the plain entry: 8x8 TUs ride sixteen to a matrix-core tile by default ("qr8_tile_kernel" 0: the register kernel)
                    for tile8 in ((1, 0) if n == 8 else (1,)):
                        _lib.check(L.kvz_hip_set_tuning(b"qr8_tile_kernel", tile8), "tuning")
                        plain = api.quantize_residual_batch(ref_in, pred, n, qp, color, 0, 0)
                        for a, b, nm in zip(plain[:3], want, ("rec", "coeff", "has")):
                            np.testing.assert_array_equal(a, b, err_msg="plain %s n=%d count=%d qp=%d fast=%d tile8=%d" % (nm, n, count, qp, use, tile8))
    finally:
        L.kvz_hip_set_tuning(key, -1)
        L.kvz_hip_set_tuning(b"qr8_tile_kernel", -1)


def test_quantize_residual_4_both_kernels(api):
    """4x4 TUs run one lane per TU in registers by default; the LDS kernel stays selectable (and serves sign hiding)"""
    from kvazaar_amd import _lib
    L = _lib.init(0)
    g = rng(170)
    for count in (1, 3, 257, 1000):
        ref_in = g.integers(0, 256, (count, 16), dtype=np.uint8)
        pred = np.clip(ref_in.astype(np.int32) + g.integers(-70, 71, ref_in.shape), 0, 255).astype(np.uint8)
        pred[::3] = ref_in[::3]
        for qp in (5, 27, 48):
            for (color, intra, ts) in ((0, 1, 0), (0, 0, 0), (1, 1, 0), (0, 0, 1), (2, 1, 1)):
                want = O.quantize_residual_batch(ref_in, pred, 4, qp, color, 0, intra, intra, 0, ts)
                for use in (1, 0):
                    _lib.check(L.kvz_hip_set_tuning(b"qr4_lane_kernel", use), "tuning")
                    got = api.quantize_residual_batch(ref_in, pred, 4, qp, color, 0, intra, intra, 0, ts, with_costs=True)
                    for a, b, nm in zip(got[:3], want, ("rec", "coeff", "has")):
                        np.testing.assert_array_equal(a, b, err_msg="%s count=%d qp=%d color=%d intra=%d ts=%d lane=%d" % (nm, count, qp, color, intra, ts, use))
                    if count <= 3:
                        for i in range(count):
                            assert got[3][i] == O.pixels_calc_ssd(ref_in[i], 0, want[0][i], 0, 4, 4, 4)
                            assert got[4][i] == O.coeff_abs_sum(want[1][i])
    L.kvz_hip_set_tuning(b"qr4_lane_kernel", -1)


@pytest.mark.parametrize("w", [4, 8, 16, 32])
def test_quantize_residual_fused_rd0_costs(api, w):
    """kvz_hip_quantize_residual_cost_batch: the SSD(ref, rec) and coeff_abs_sum the rd=0 TU cost is made of
    (search.c:291, rdo.c:219) equal the oracle's pixels_calc_ssd / coeff_abs_sum of the same launch's outputs"""
    g = rng(150 + w)
    ref_in = g.integers(0, 256, (77, w * w), dtype=np.uint8)
    pred = np.clip(ref_in.astype(np.int32) + g.integers(-60, 61, ref_in.shape), 0, 255).astype(np.uint8)
    pred[0] = ref_in[0]
    pred[1] = 255 - ref_in[1]
    for qp in (17, 32, 47):
        for (intra, signhide, trskip) in ((0, 0, 0), (1, 1, 0)) + (((0, 0, 1),) if w == 4 else ()):
            rec, coeff, has, ssd, sab = api.quantize_residual_batch(ref_in, pred, w, qp, 0, 0, intra, intra, signhide, trskip,
                                                                    with_costs=True)
            want = O.quantize_residual_batch(ref_in, pred, w, qp, 0, 0, intra, intra, signhide, trskip)
            for a, b in zip((rec, coeff, has), want):
                np.testing.assert_array_equal(a, b)
            for i in range(ref_in.shape[0]):
                assert ssd[i] == O.pixels_calc_ssd(ref_in[i], 0, want[0][i], 0, w, w, w), (qp, i)
                assert sab[i] == O.coeff_abs_sum(want[1][i]), (qp, i)


# ------------------------------------------------------------------ ipol
@pytest.mark.parametrize("kind", ["luma", "luma14", "chroma", "chroma14"])
def test_sample_filters(api, kind):
    g = rng(60)
    frame = g.integers(0, 256, (96, 96), dtype=np.uint8)
    frame[40:60, 40:60] = np.where(g.integers(0, 2, (20, 20)) > 0, 255, 0)
    luma = kind.startswith("luma")
    nfrac = 4 if luma else 8
    sizes = ((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 4)) if luma else \
            ((4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (2, 2))
    blocks = [(12, 10, fx, fy, w, h) for (w, h) in sizes for fx in range(nfrac) for fy in range(nfrac)]
    got = api.sample_batch(kind, frame, blocks)
    for b, o in zip(blocks, got):
        x, y, fx, fy, w, h = b
        np.testing.assert_array_equal(o, O.sample(kind, frame, x, y, w, h, fx, fy), err_msg=str(b))


def test_sample_filters_edge_replication(api):
    """blocks whose filter window leaves the frame: kvz_get_extended_block semantics"""
    g = rng(61)
    frame = g.integers(0, 256, (40, 56), dtype=np.uint8)
    pad = 80
    padded = np.pad(frame, pad, mode="edge")
    blocks = [(x, y, fx, fy, 16, 8) for (x, y) in ((-5, -3), (50, 36), (-20, 10), (30, -9), (56, 40), (0, 0))
              for (fx, fy) in ((1, 2), (3, 3), (0, 1))]
    got = api.sample_batch("luma", frame, blocks)
    for b, o in zip(blocks, got):
        x, y, fx, fy, w, h = b
        np.testing.assert_array_equal(o, O.sample("luma", padded, x + pad, y + pad, w, h, fx, fy), err_msg=str(b))


def test_search_frac(api):
    g = rng(80)
    ref = g.integers(0, 256, (72, 96), dtype=np.uint8)
    pic = ((ref.astype(np.int32) + np.roll(ref, 1, axis=1)) // 2).astype(np.uint8)
    pairs, meta = [], []
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 32), (8, 4), (4, 8), (16, 4), (4, 16), (16, 12), (12, 16)):
        for (x, y) in ((0, 0), (32 - w // 2, 24 - h // 4), (96 - w, 72 - h)):
            for (mvx, mvy) in ((0, 0), (-2, 1), (5, -3), (-40, -40), (90, 70)):
                pairs.append((x, y, x + mvx, y + mvy, w, h))
                meta.append((x, y, w, h, mvx, mvy))
    costs, best = api.search_frac_batch(pic, ref, pairs)
    for i, (x, y, w, h, mvx, mvy) in enumerate(meta):
        oc, ob = O.search_frac_costs(pic, ref, x, y, w, h, mvx, mvy)
        np.testing.assert_array_equal(costs[i], oc, err_msg=str(meta[i]))
        assert tuple(best[i]) == ob, meta[i]


def test_search_frac_extreme_pixels(api):
    """0/255 checkerboards drive the int16 truncation paths of the filters"""
    g = rng(81)
    ref = np.where(g.integers(0, 2, (64, 64)) > 0, 255, 0).astype(np.uint8)
    pic = np.where(g.integers(0, 2, (64, 64)) > 0, 255, 0).astype(np.uint8)
    pairs = [(8, 8, 8 + dx, 8 + dy, 16, 16) for dx in (-1, 0, 2) for dy in (-2, 0, 1)]
    costs, best = api.search_frac_batch(pic, ref, pairs)
    for i, p in enumerate(pairs):
        oc, ob = O.search_frac_costs(pic, ref, p[0], p[1], 16, 16, p[2] - p[0], p[3] - p[1])
        np.testing.assert_array_equal(costs[i], oc)
        assert tuple(best[i]) == ob


# ------------------------------------------------------------------ batched ME
def test_ctu_sad_grid(api):
    """one CTU per workgroup: all candidates x all 85 square PUs == kvz_image_calc_sad per (PU, candidate)"""
    g = rng(90)
    H, W = 200, 264                       # ragged: last CTU row 8 px high, last CTU column 8 px wide
    ref = g.integers(0, 256, (H, W), dtype=np.uint8)
    pic = np.clip(np.roll(ref, (2, -3), axis=(0, 1)).astype(np.int32) + g.integers(-5, 6, (H, W)), 0, 255).astype(np.uint8)
    ctus = [(x, y, mvx, mvy) for y in range(0, H, 64) for x in range(0, W, 64)
            for (mvx, mvy) in ((0, 0), (3, -2), (-70, 5), (40, 150))]
    grid = [(dx, dy) for dy in (-6, -3, 0, 3, 6) for dx in (-6, -3, 0, 3, 6)]           # speed_tests.c:205-236
    hexbs = [(0, 0), (-2, 0), (-1, -2), (1, -2), (2, 0), (1, 2), (-1, 2), (1, 0), (0, 1), (-1, 0), (0, -1)]
    wide = [(-64, -64), (64, 64), (17, -33), (-1, 1), (65, 0), (0, -100)]                # incl. two out-of-range offsets
    for offs in (grid, hexbs, wide):
        got = api.ctu_sad_grid_batch(pic, ref, ctus, offs)
        want = O.ctu_sad_grid(pic, ref, ctus, offs)
        np.testing.assert_array_equal(got, want)
    # hierarchy: a 16x16 cost equals the sum of its four 8x8 costs where all are valid
    c = api.ctu_sad_grid_batch(pic, ref, [(0, 0, 1, 1)], grid)[0]
    s8 = c[:, 21:].reshape(-1, 8, 8).astype(np.int64)
    s16 = s8.reshape(-1, 4, 2, 4, 2).sum(axis=(2, 4)).reshape(-1, 16)
    np.testing.assert_array_equal(c[:, 5:21], s16)
    assert (c[:, 0] == s8.sum(axis=(1, 2))).all()


def test_transform_16x16_tile_tails(api):
    """16x16 transforms run four blocks per matrix-core tile: every count modulo 4, extreme inputs, both directions; an unknown
    tuning key is refused"""
    from kvazaar_amd import _lib
    L = _lib.load()
    g = rng(21)
    for count in (1, 2, 3, 4, 5, 6, 7, 131):
        x = g.integers(-32768, 32768, (count, 256)).astype(np.int16)
        x[0, :] = 32767 if count % 2 else -32768
        for kind in ("dct", "idct"):
            np.testing.assert_array_equal(api.transform_batch(kind, 16, x), O.transform_batch(kind, 16, x), err_msg="%s count %d" % (kind, count))
    assert L.kvz_hip_set_tuning(b"no_such_key", 1) != 0


# ---- intra group ----
from patterns import intra_ref_cases  # noqa: E402


def _intra_orig(refs, log2_width, seed):
    """original blocks that resemble one of the predictions plus noise, so costs span small and large values"""
    g = rng(seed)
    n = 1 << log2_width
    count = refs.shape[0]
    base = O.intra_predict_batch(refs, log2_width, [int(g.integers(0, 35))])[:, 0, :].astype(np.int32)
    noise = g.integers(-12, 13, (count, n * n))
    orig = np.clip(base + noise, 0, 255).astype(np.uint8)
    orig[::4] = g.integers(0, 256, (len(orig[::4]), n * n), dtype=np.uint8)
    return orig


@pytest.mark.parametrize("log2_width", [2, 3, 4, 5])
@pytest.mark.parametrize("flags", [3, 1, 0, 2, 4])
def test_intra_predict(api, log2_width, flags):
    refs = intra_ref_cases(log2_width, 23, 500 + log2_width)
    modes = list(range(35))
    got = api.intra_predict_batch(refs, log2_width, modes, flags)
    if flags & 4:
        # bare strategies: angular_pred / intra_pred_planar on the given references, plain DC
        for i, r in enumerate(refs):
            left, top = r[:65], r[65:]
            np.testing.assert_array_equal(got[i, 0], O.intra_pred_planar(log2_width, top, left))
            for m in range(2, 35):
                np.testing.assert_array_equal(got[i, m], O.angular_pred(log2_width, m, top, left), err_msg="pu %d mode %d" % (i, m))
    else:
        want = O.intra_predict_batch(refs, log2_width, modes, is_luma=flags & 1, filter_boundary=(flags >> 1) & 1)
        np.testing.assert_array_equal(got, want)
    # a short list in another order, one PU
    got = api.intra_predict_batch(refs[:1], log2_width, [26, 0, 10], 3)
    np.testing.assert_array_equal(got, O.intra_predict_batch(refs[:1], log2_width, [26, 0, 10]))


@pytest.mark.parametrize("log2_width", [2, 3, 4, 5])
@pytest.mark.parametrize("count", [1, 7, 64, 203])
def test_intra_rough_costs(api, log2_width, count):
    refs = intra_ref_cases(log2_width, count, 540 + log2_width)
    orig = _intra_orig(refs, log2_width, 9 + count)
    for fb in (1, 0):
        satd, sad = api.intra_rough_batch(refs, log2_width, orig, 1 | (fb << 1), with_sad=True)
        want_satd, want_sad = O.intra_rough_costs_batch(refs, log2_width, orig, fb)
        np.testing.assert_array_equal(satd, want_satd)
        np.testing.assert_array_equal(sad, want_sad)
    np.testing.assert_array_equal(api.intra_rough_batch(refs, log2_width, orig), want_satd if fb == 1 else
                                  O.intra_rough_costs_batch(refs, log2_width, orig, 1)[0])


def test_intra_rough_matches_predict_plus_satd_at_frame_scale(api):
    """size-independent property at one 1080p frame of 8x8 PUs: the fused costs equal satd_8x8 of the
    predictions the predict kernel writes, for a sample of modes"""
    g = rng(61)
    count = 32400
    refs = g.integers(0, 256, (count, 130), dtype=np.uint8)
    refs[:, 65] = refs[:, 0]
    orig = g.integers(0, 256, (count, 64), dtype=np.uint8)
    satd = api.intra_rough_batch(refs, 3, orig)
    modes = [0, 1, 2, 10, 17, 18, 26, 34]
    pred = api.intra_predict_batch(refs, 3, modes)
    for k, m in enumerate(modes):
        c = api.cost_nxn_batch("satd", 8, np.ascontiguousarray(pred[:, k, :]), orig)
        np.testing.assert_array_equal(satd[:, m], c, err_msg="mode %d" % m)


@pytest.mark.parametrize("log2_width_c", [2, 3, 4, 5])
def test_intra_chroma_rough_search(api, log2_width_c):
    """search_intra_chroma_rough (search_intra.c:329-370): for the five candidate chroma modes, cost = satd(U prediction) +
    satd(V prediction) with kvz_intra_predict(..., COLOR_U / COLOR_V, filter_boundary false) -- the rough kernel with
    KVZ_HIP_INTRA_LUMA clear (no reference smoothing, no DC / boundary filters), one launch per plane; the host adds the
    two tables and sorts like sort_modes"""
    count, n = 40, 1 << log2_width_c
    refs_u = intra_ref_cases(log2_width_c, count, 610 + log2_width_c)
    refs_v = intra_ref_cases(log2_width_c, count, 650 + log2_width_c)
    orig_u = _intra_orig(refs_u, log2_width_c, 3)
    orig_v = _intra_orig(refs_v, log2_width_c, 4)
    got = api.intra_rough_batch(refs_u, log2_width_c, orig_u, flags=0).astype(np.int64) + \
        api.intra_rough_batch(refs_v, log2_width_c, orig_v, flags=0).astype(np.int64)
    for luma_mode in (0, 26, 14):
        modes = [0, 26, 10, 1, 34 if luma_mode in (0, 26, 10, 1) else luma_mode]       # search_intra.c:766-778
        want = np.zeros((count, 5), np.int64)
        for refs, orig in ((refs_u, orig_u), (refs_v, orig_v)):
            pred = O.intra_predict_batch(refs, log2_width_c, modes, is_luma=0, filter_boundary=0)
            for k in range(5):
                want[:, k] += O.cost_nxn_batch("satd", n, np.ascontiguousarray(pred[:, k, :]), orig)
        np.testing.assert_array_equal(got[:, modes], want)
        # the decision: same order after the reference's insertion sort (stable: ties keep list order)
        np.testing.assert_array_equal(np.argsort(got[:, modes], axis=1, kind="stable"), np.argsort(want, axis=1, kind="stable"))


@pytest.mark.parametrize("color", [0, 1, 2])
@pytest.mark.parametrize("log2_width", [2, 3, 4, 5])
def test_intra_build_reference(api, log2_width, color):
    """kvz_intra_build_reference at every PU position of a 1080p reconstruction (17 x 30 LCUs, ragged last LCU row), read
    from a plane wider than the picture"""
    from patterns import intra_ref_positions
    g = rng(700 + 10 * log2_width + color)
    pic_w, pic_h, c = 1920, 1080, 1 if color else 0
    plane = g.integers(0, 256, (pic_h >> c, (pic_w >> c) + 24), dtype=np.uint8)
    xy = intra_ref_positions(log2_width, color, pic_w, pic_h)
    got = api.intra_build_reference_batch(log2_width, color, plane, pic_w, pic_h, xy)
    np.testing.assert_array_equal(got, O.intra_build_reference_batch(log2_width, color, plane, pic_w, pic_h, xy))


def test_intra_build_reference_small_pictures_and_bad_positions(api):
    """pictures narrower than one LCU (every PU on a picture edge), and positions the kernel must refuse without reading"""
    g = rng(77)
    for (pic_w, pic_h) in ((8, 8), (16, 8), (8, 64), (72, 24), (64, 64)):
        for color in (0, 1):
            plane = g.integers(0, 256, (pic_h >> color, pic_w >> color), dtype=np.uint8)
            for log2_width in (2, 3, 4, 5):
                from patterns import intra_ref_positions
                xy = intra_ref_positions(log2_width, color, pic_w, pic_h)
                if len(xy) == 0:
                    continue
                np.testing.assert_array_equal(api.intra_build_reference_batch(log2_width, color, plane, pic_w, pic_h, xy),
                                              O.intra_build_reference_batch(log2_width, color, plane, pic_w, pic_h, xy))
    plane = g.integers(1, 256, (64, 64), dtype=np.uint8)
    xy = np.array([(8, 8), (-4, 0), (0, -8), (60, 0), (0, 60), (6, 8), (8, 2), (64, 0), (1 << 30, 1 << 30), (56, 56)], np.int32)
    got = api.intra_build_reference_batch(3, 0, plane, 64, 64, xy)
    want = O.intra_build_reference_batch(3, 0, plane, 64, 64, xy[[0, 9]])
    np.testing.assert_array_equal(got[[0, 9]], want)
    assert not got[1:9].any()


def test_intra_references_feed_the_rough_search_on_the_device(api):
    """build -> rough search chained on one stream through device buffers only (the references never visit the host):
    same 35 costs per PU as the oracle's rough search on the oracle's references"""
    from kvazaar_amd import _lib
    from kvazaar_amd.api import DeviceBuffer, check
    from patterns import intra_ref_positions
    L = _lib.init()
    g = rng(78)
    pic_w, pic_h, lg = 192, 136, 3
    rec = g.integers(0, 256, (pic_h, pic_w), dtype=np.uint8)
    xy = intra_ref_positions(lg, 0, pic_w, pic_h)
    count = len(xy)
    orig = g.integers(0, 256, (count, 64), dtype=np.uint8)
    d_rec, d_xy, d_orig = DeviceBuffer.from_numpy(rec), DeviceBuffer.from_numpy(xy), DeviceBuffer.from_numpy(orig)
    d_refs, d_satd = DeviceBuffer(130 * count), DeviceBuffer(4 * 35 * count)
    st = L.kvz_hip_stream_create()
    assert st
    try:
        check(L.kvz_hip_intra_build_reference_batch(lg, 0, d_rec.ptr, pic_w, pic_w, pic_h, d_xy.ptr, count, d_refs.ptr, st), "build")
        check(L.kvz_hip_intra_rough_batch(lg, 3, d_refs.ptr, d_orig.ptr, count, d_satd.ptr, None, st), "rough")
        satd = d_satd.to_numpy(np.uint32, (count, 35), stream=st)
    finally:
        L.kvz_hip_stream_destroy(st)
    refs = O.intra_build_reference_batch(lg, 0, rec, pic_w, pic_h, xy)
    np.testing.assert_array_equal(satd, O.intra_rough_costs_batch(refs, lg, orig, 1)[0])


def test_intra_argument_errors(api):
    from kvazaar_amd._lib import KvzHipError
    refs = intra_ref_cases(3, 2, 1)
    with pytest.raises(KvzHipError):
        api.intra_predict_batch(refs, 6, [0])
    with pytest.raises(KvzHipError):
        api.intra_predict_batch(refs, 3, [35])
    with pytest.raises(KvzHipError):
        api.intra_rough_batch(refs, 3, np.zeros((2, 64), np.uint8), flags=4)
    assert api.intra_rough_batch(refs[:0], 3, np.zeros((0, 64), np.uint8)).shape == (0, 35)
    plane, xy = np.zeros((64, 64), np.uint8), np.zeros((1, 2), np.int32)
    for bad in (dict(log2_width=1), dict(log2_width=6), dict(color=3), dict(pic_w=60), dict(pic_h=0), dict(pic_w=72)):
        a = dict(log2_width=3, color=0, plane=plane, pic_w=64, pic_h=64, xy=xy)
        a.update(bad)
        with pytest.raises(KvzHipError):
            api.intra_build_reference_batch(**a)
    assert api.intra_build_reference_batch(3, 0, plane, 64, 64, xy[:0]).shape == (0, 130)


# ---- motion search of whole PUs ----
from patterns import ME_RESULT, me_frames, me_params, me_pus_in_tile, me_random_pus  # noqa: E402

ME_CONFIGS = [
    dict(), dict(early_termination=2, fme_level=2, lambda_cost=35), dict(early_termination=0, lambda_cost=4),
    dict(fme_level=0, lambda_cost=60), dict(fme_level=1), dict(fme_level=3, max_steps=2),
    dict(wpp_owf=1, ref_delay_px=10, max_ref_lcu_down=1, max_ref_lcu_right=1),
    dict(wpp_owf=1, ref_delay_px=8, max_ref_lcu_down=0, max_ref_lcu_right=2, lambda_cost=9),
    dict(algorithm=1), dict(algorithm=1, early_termination=0, max_steps=3, lambda_cost=50), dict(algorithm=1, fme_level=2, early_termination=2),
    dict(algorithm=2), dict(algorithm=2, early_termination=0, lambda_cost=6), dict(algorithm=2, wpp_owf=1, ref_delay_px=10, fme_level=3),
    dict(algorithm=3, search_range=8), dict(algorithm=3, search_range=16, lambda_cost=40, wpp_owf=1, ref_delay_px=8, fme_level=2),
    # kvz_mv_constraint (kvazaar.h:113-119; fracmv_within_tile search_inter.c:142-171): the frame as one tile, then real tiles
    dict(mv_constraint=1), dict(mv_constraint=2, lambda_cost=7, early_termination=0), dict(mv_constraint=3, algorithm=1),
    dict(mv_constraint=4), dict(mv_constraint=4, algorithm=2, fme_level=2), dict(mv_constraint=4, algorithm=3, search_range=8, lambda_cost=33),
    dict(mv_constraint=3, tile=(64, 0, 128, 128)), dict(mv_constraint=4, tile=(0, 64, 192, 64), lambda_cost=11),
    dict(mv_constraint=4, tile=(64, 64, 64, 64), early_termination=0, fme_level=3),
    dict(mv_constraint=0, tile=(64, 0, 128, 128), wpp_owf=1, ref_delay_px=10, max_ref_lcu_down=1, max_ref_lcu_right=1),
    dict(mv_constraint=4, tile=(0, 64, 192, 64), wpp_owf=1, ref_delay_px=8, max_ref_lcu_down=0, max_ref_lcu_right=1, algorithm=1),
]


def _me_compare(api, pic, ref, pus, prm, msg):
    got = api.search_pu_batch(pic, ref, pus, prm).view(ME_RESULT).reshape(-1)
    want = O.search_pu_batch(pic, ref, pus, prm)
    for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
        np.testing.assert_array_equal(got[f], want[f], err_msg="%s %s" % (f, msg))


@pytest.mark.parametrize("cfg", range(len(ME_CONFIGS)))
def test_search_pu(api, cfg):
    prm = me_params(**ME_CONFIGS[cfg])
    for k, motion in enumerate(((3, -2), (-7, 5), (0, 0), (14, 9))):
        pic, ref = me_frames(192, 128, 900 + k, motion)
        pus = me_pus_in_tile(me_random_pus(192, 128, 60, 177 + 10 * cfg + k, hint=(-4 * motion[0] + 2, -4 * motion[1])), prm)
        _me_compare(api, pic, ref, pus, prm, "cfg %d motion %s" % (cfg, motion))


def test_search_pu_tile_constraint_keeps_reads_inside_the_shard(api):
    """A CTU-row shard is a full-width tile.  With mv_constraint 4 no chosen vector may make the fractional
    interpolation read outside the tile; proven by poisoning everything outside the tile's rows in the reference plane:
    results must not change.  Without the constraint the same poison does change them (the check has teeth)."""
    pic, ref = me_frames(256, 320, 41, (2, 9))
    tile = (0, 128, 256, 64)
    pus = me_random_pus(256, 320, 160, 19, hint=(-8, -36), sizes=((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (32, 16)))
    prm = me_params(mv_constraint=4, tile=tile, lambda_cost=4, early_termination=0)
    pus = me_pus_in_tile(pus, prm)
    poisoned = ref.copy()
    g = np.random.default_rng(3)
    poisoned[:128] = g.integers(0, 256, poisoned[:128].shape)
    poisoned[192:] = g.integers(0, 256, poisoned[192:].shape)
    a = api.search_pu_batch(pic, ref, pus, prm)
    b = api.search_pu_batch(pic, poisoned, pus, prm)
    np.testing.assert_array_equal(a, b)
    _me_compare(api, pic, poisoned, pus, prm, "constrained, poisoned")
    free = me_params(mv_constraint=0, tile=tile, lambda_cost=4, early_termination=0)
    c, d = api.search_pu_batch(pic, ref, pus, free), api.search_pu_batch(pic, poisoned, pus, free)
    assert (c != d).any()
    mv = a.view(ME_RESULT).reshape(-1)["mv"]
    top = (pus["y"] - 128) * 4 + mv[:, 1]
    bottom = (192 - pus["y"] - pus["height"]) * 4 - mv[:, 1]
    frac = (mv[:, 0] % 4 != 0) | (mv[:, 1] % 4 != 0)
    assert (top >= np.where(frac, 16, 0)).all() and (bottom >= np.where(frac, 16, 0)).all()


def test_search_pu_hint_leaves_no_result_unwritten(api):
    """size_classes names only the small class: PUs of the other classes and malformed descriptors read back as
    cost 0xFFFFFFFF / reserved -1 instead of uninitialised memory"""
    pic, ref = me_frames(192, 128, 5, (1, 1))
    pus = me_random_pus(192, 128, 40, 3, sizes=((8, 8), (16, 16), (32, 32), (64, 64)))
    pus[7]["width"] = 10
    prm = me_params()
    prm["size_classes"] = 1
    r = api.search_pu_batch(pic, ref, pus, prm).view(ME_RESULT).reshape(-1)
    small = (pus["width"] <= 16) & (pus["height"] <= 16) & (pus["width"] != 10)
    assert (r["cost"][~small] == 0xFFFFFFFF).all() and (r["reserved"][~small] == -1).all()
    want = O.search_pu_batch(pic, ref, pus[small], me_params())
    for f in ("mv", "cost", "bitcost"):
        np.testing.assert_array_equal(r[small][f], want[f], err_msg=f)


MV_RDO_CONFIGS = [
    dict(mv_rdo=1), dict(mv_rdo=1, refs_before=3, ref_idx=2, lambda_cost=11), dict(mv_rdo=1, refs_before=2, ref_idx=0, algorithm=1, fme_level=2),
    dict(mv_rdo=1, refs_before=4, ref_idx=1, algorithm=2, early_termination=0), dict(mv_rdo=1, fme_level=0, refs_before=2, ref_idx=1),
    dict(mv_rdo=1, refs_before=5, ref_idx=4, mv_constraint=4, lambda_cost=40),
]


@pytest.mark.parametrize("cfg", range(len(MV_RDO_CONFIGS)))
def test_search_pu_mv_rdo(api, cfg):
    """--mv-rdo: kvz_calc_mvd_cost_cabac / kvz_get_mvd_coding_cost_cabac (rdo.c:883-1060) as the search's cost model, from per-PU
    CABAC snapshots; every PU size (one workgroup per PU), SMP / AMP shapes included"""
    from patterns import me_cabac_states
    prm = me_params(**MV_RDO_CONFIGS[cfg])
    for k, motion in enumerate(((3, -2), (-7, 5), (0, 0))):
        pic, ref = me_frames(192, 128, 900 + k, motion)
        pus = me_random_pus(192, 128, 60, 277 + 10 * cfg + k, hint=(-4 * motion[0] + 2, -4 * motion[1]),
                            sizes=((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (64, 32), (24, 32), (32, 8), (8, 4), (16, 12)))
        pus["x"] = np.minimum(pus["x"], 192 - pus["width"]); pus["y"] = np.minimum(pus["y"], 128 - pus["height"])
        cab = me_cabac_states(9, 15 + k + cfg)
        pus["reserved"] = np.arange(len(pus)) % 9
        got = api.search_pu_batch(pic, ref, pus, prm, cabac=cab).view(ME_RESULT).reshape(-1)
        want = O.search_pu_batch(pic, ref, pus, prm, cabac=cab)
        for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
            np.testing.assert_array_equal(got[f], want[f], err_msg="%s cfg %d motion %s" % (f, cfg, motion))
    from kvazaar_amd._lib import KvzHipError
    with pytest.raises(KvzHipError):
        api.search_pu_batch(pic, ref, pus, prm)            # mv_rdo without the snapshots is refused


def test_search_pu_flat_and_borders(api):
    """flat frames (every candidate ties: the reference's first-wins order decides) and vectors that leave the frame"""
    prm = me_params(lambda_cost=3, early_termination=0)
    flat = np.full((128, 192), 77, np.uint8)
    pus = me_random_pus(192, 128, 30, 5)
    _me_compare(api, flat, flat, pus, prm, "flat")
    pic, ref = me_frames(192, 128, 31, (20, -17))
    pus = me_random_pus(192, 128, 60, 6, sizes=((8, 8), (16, 16), (64, 64), (32, 32)))
    pus["x"] = np.where(np.arange(60) % 2 == 0, 0, 192 - pus["width"])       # hug the left / right border
    pus["y"] = np.where(np.arange(60) % 3 == 0, 0, 128 - pus["height"])
    pus["extra_mv"] = np.array([[-130, 90], [150, -120], [4, 300]] * 20, dtype=np.int16)   # start outside the frame
    _me_compare(api, pic, ref, pus, prm, "borders")


def test_search_pu_frame_of_ctus_and_bad_descriptors(api):
    """every 8x8..64x64 PU of a 6 x 4 CTU frame in one launch; malformed descriptors are flagged, not searched"""
    prm = me_params()
    pic, ref = me_frames(384, 256, 77, (-5, 3))
    rows = []
    for n in (64, 32, 16, 8):
        for y in range(0, 256, n):
            for x in range(0, 384, n):
                rows.append((x, y, n))
    pus = me_random_pus(384, 256, len(rows), 8, hint=(22, -12))
    for i, (x, y, n) in enumerate(rows):
        pus[i]["x"], pus[i]["y"], pus[i]["width"], pus[i]["height"] = x, y, n, n
    sel = np.arange(0, len(rows), 7)                       # the oracle takes ~1 ms per PU: compare a strided subset
    got = api.search_pu_batch(pic, ref, pus, prm).view(ME_RESULT).reshape(-1)
    want = O.search_pu_batch(pic, ref, pus[sel], prm)
    for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
        np.testing.assert_array_equal(got[sel][f], want[f], err_msg=f)
    bad = pus[:5].copy()
    bad[0]["width"] = 10; bad[1]["x"] = 380; bad[2]["num_merge_cand"] = 9; bad[3]["height"] = 0
    bad[4]["width"] = 12; bad[4]["height"] = 12          # both dimensions 4 mod 8: no such PU
    r = api.search_pu_batch(pic, ref, bad, prm).view(ME_RESULT).reshape(-1)
    assert (r["cost"] == 0xFFFFFFFF).all() and (r["reserved"] == -1).all()


# ---- SAO group ----
from patterns import sao_blocks, sao_records  # noqa: E402


# widths that are multiples of 4 take the packed-accumulator edge kernel (one dword column when bw == 4), the others the generic one
@pytest.mark.parametrize("bw,bh", [(64, 64), (32, 32), (64, 56), (16, 24), (8, 8), (3, 3), (40, 2), (1, 1), (4, 4), (4, 3), (60, 64), (12, 5), (62, 64)])
def test_sao_statistics_and_ddistortion(api, bw, bh):
    orig, rec = sao_blocks(bw, bh, 21, 40 + bw + bh)
    g = rng(5)
    stats = api.sao_edge_stats_batch(orig, rec, bw, bh)
    offs = g.integers(-7, 8, (len(orig), 4, 5)).astype(np.int32)
    offs[::2, :, 0] = 0
    dd = api.sao_edge_ddistortion_batch(orig, rec, bw, bh, offs)
    bands = api.sao_band_stats_batch(orig, rec, bw, bh)
    bp = g.integers(0, 32, len(orig)).astype(np.int32)
    bo = g.integers(-7, 8, (len(orig), 4)).astype(np.int32)
    bdd = api.sao_band_ddistortion_batch(orig, rec, bw, bh, bp, bo)
    for i in range(len(orig)):
        for eo in range(4):
            np.testing.assert_array_equal(stats[i, eo], O.calc_sao_edge_dir(orig[i], rec[i], eo, bw, bh), err_msg="blk %d class %d" % (i, eo))
            assert dd[i, eo] == O.sao_edge_ddistortion(orig[i], rec[i], bw, bh, eo, offs[i, eo])
        np.testing.assert_array_equal(bands[i], O.calc_sao_bands(orig[i], rec[i], bw, bh))
        assert bdd[i] == O.sao_band_ddistortion(orig[i], rec[i], bw, bh, int(bp[i]), bo[i])
    if bw % 4 == 0 and bh >= 3:
        # the generic kernel on the same shape (tuning knob): both implementations stay covered
        from kvazaar_amd import _lib
        L = _lib.init(0)
        assert L.kvz_hip_set_tuning(b"sao_edge_fast", 0) == 0
        try:
            np.testing.assert_array_equal(api.sao_edge_stats_batch(orig, rec, bw, bh), stats)
            np.testing.assert_array_equal(api.sao_edge_ddistortion_batch(orig, rec, bw, bh, offs), dd)
        finally:
            L.kvz_hip_set_tuning(b"sao_edge_fast", -1)


@pytest.mark.parametrize("color", [0, 1, 2])
def test_sao_reconstruct_plane(api, color):
    """an LCU grid over a plane, blocks trimmed at the border the way kvz_sao_reconstruct does; one sao record per LCU"""
    g = rng(21 + color)
    H, W, n = 136, 200, 64
    plane = g.integers(0, 256, (H, W), dtype=np.uint8)
    plane[20:60, 30:90] = np.where(g.integers(0, 2, (40, 60)) > 0, 252, 2)
    blocks, k = [], 0
    for y in range(0, H, n):
        for x in range(0, W, n):
            x0, y0, x1, y1 = max(x, 1), max(y, 1), min(x + n, W - 1), min(y + n, H - 1)
            blocks.append((x0, y0, x1 - x0, y1 - y0, k))
            k += 1
    infos = sao_records(k, 9 + color)
    infos[2, 0] = 0                                       # SAO_TYPE_NONE: plain copy
    got = api.sao_reconstruct_color_batch(plane, blocks, infos, color)
    want = plane.copy()
    for (x, y, w, h, idx) in blocks:
        if infos[idx, 0] != 0:
            want[y:y + h, x:x + w] = O.sao_reconstruct_color(plane, x, y, w, h, infos[idx], color)
    np.testing.assert_array_equal(got, want)
    # a descriptor that would read outside the plane is skipped, not executed
    edge = sao_records(2, 1)[1:2]
    bad = [(0, 0, 8, 8, 0), (W - 8, H - 8, 8, 8, 0), (4, 4, 8, 8, 3)]
    edge[0, 1] = 2
    np.testing.assert_array_equal(api.sao_reconstruct_color_batch(plane, bad, edge, color), plane)


def test_bipred_candidate_cost(api):
    """search_pu_inter_bipred's candidate score for integer / half / quarter-pel vector pairs, inside and across the frame"""
    g = rng(91)
    pic, ref0 = me_frames(192, 128, 12, (2, -1))
    _, ref1 = me_frames(192, 128, 13, (-3, 2))
    ref1 = np.where(g.integers(0, 40, ref1.shape) == 0, 255, ref1).astype(np.uint8)
    cands = []
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (32, 64), (24, 8), (8, 4), (4, 8), (16, 4), (4, 16), (16, 12), (12, 16)):
        for k in range(14):
            x = int(g.integers(0, 3)) * 64 + int(g.integers(0, (64 - w) // 4 + 1)) * 4
            y = int(g.integers(0, 2)) * 64 + int(g.integers(0, (64 - h) // 4 + 1)) * 4
            big = 400 if k % 5 == 4 else 24
            mv0, mv1 = g.integers(-big, big + 1, 2), g.integers(-big, big + 1, 2)
            if k % 3 == 0:
                mv0 = (mv0 // 4) * 4
            if k % 4 == 1:
                mv1 = (mv1 // 4) * 4
            cands.append((x, y, w, h, int(mv0[0]), int(mv0[1]), int(mv1[0]), int(mv1[1])))
    got = api.bipred_cost_batch(pic, ref0, ref1, cands)
    for c, v in zip(cands, got):
        assert v == O.bipred_luma_satd(pic, ref0, ref1, c[0], c[1], c[2], c[3], c[4:6], c[6:8])[0], c
    bad = api.bipred_cost_batch(pic, ref0, ref1, [(0, 0, 10, 8, 0, 0, 0, 0), (190, 0, 8, 8, 0, 0, 0, 0), (0, 0, 12, 12, 0, 0, 0, 0)])
    assert (bad == 0xFFFFFFFF).all()


def test_search_pu_size_class_hint(api):
    """size_classes only prunes launches: with the hint naming the classes present the results are unchanged"""
    pic, ref = me_frames(192, 128, 5, (4, 1))
    for cls, sizes in ((1, ((8, 8), (16, 8), (16, 16))), (2, ((32, 32), (24, 16), (32, 8))), (4, ((64, 64), (64, 32), (48, 16))), (3, ((8, 8), (32, 32)))):
        pus = me_random_pus(192, 128, 40, 50 + cls, sizes=sizes)
        plain = api.search_pu_batch(pic, ref, pus, me_params())
        prm = me_params()
        prm["size_classes"] = cls
        np.testing.assert_array_equal(api.search_pu_batch(pic, ref, pus, prm), plain)


def test_invalid_arguments_leave_a_message(api):
    """every KVZ_HIP_ERR_INVALID return records which entry refused its arguments"""
    from kvazaar_amd import _lib
    L = _lib.init(0)
    assert L.kvz_hip_sad_nxn_batch(7, None, None, 1, None, None) != 0
    assert b"kvz_hip_sad_nxn_batch" in L.kvz_hip_last_error()
    assert L.kvz_hip_transform_batch(0, 5, None, None, 1, None) != 0
    assert b"kvz_hip_transform_batch" in L.kvz_hip_last_error()


def test_frame_graph_replay_matches_the_oracle(api):
    """a frame's launch sequence captured once as a hipGraph (kvz_hip_graph_begin/_end) with the motion search on a
    forked stream, replayed on NEW buffer contents: every replay must equal the oracle like an eager launch does"""
    import ctypes as C
    from kvazaar_amd import _lib
    from kvazaar_amd.api import DeviceBuffer
    L = _lib.init(0)
    s, side = L.kvz_hip_stream_create(), L.kvz_hip_stream_create()
    fork, join = L.kvz_hip_event_create(), L.kvz_hip_event_create()
    count, n32 = 500, 40
    prm = me_params()
    pus = me_random_pus(192, 128, 48, 21)
    d_a, d_b = DeviceBuffer(count * 64), DeviceBuffer(count * 64)
    d_sad, d_satd = DeviceBuffer(4 * count), DeviceBuffer(4 * count)
    d_res, d_coef = DeviceBuffer(2 * n32 * 1024), DeviceBuffer(2 * n32 * 1024)
    d_pic, d_ref = DeviceBuffer(192 * 128), DeviceBuffer(192 * 128)
    d_pus, d_out = DeviceBuffer.from_numpy(pus.view(np.uint8)), DeviceBuffer(32 * len(pus))

    def enqueue():
        _lib.check(L.kvz_hip_event_record(fork, s), "fork")
        _lib.check(L.kvz_hip_stream_wait_event(side, fork), "fork wait")
        _lib.check(L.kvz_hip_search_pu_batch(d_pic.ptr, 192, 192, 128, d_ref.ptr, 192, 192, 128, d_pus.ptr, len(pus),
                                             prm.ctypes.data, d_out.ptr, side), "search_pu")
        _lib.check(L.kvz_hip_sad_nxn_batch(8, d_a.ptr, d_b.ptr, count, d_sad.ptr, s), "sad")
        _lib.check(L.kvz_hip_satd_nxn_batch(8, d_a.ptr, d_b.ptr, count, d_satd.ptr, s), "satd")
        _lib.check(L.kvz_hip_transform_batch(0, 32, d_res.ptr, d_coef.ptr, n32, s), "dct")
        _lib.check(L.kvz_hip_event_record(join, side), "join")
        _lib.check(L.kvz_hip_stream_wait_event(s, join), "join wait")

    graph = C.c_void_p()
    _lib.check(L.kvz_hip_graph_begin(s), "graph_begin")
    enqueue()
    _lib.check(L.kvz_hip_graph_end(s, C.byref(graph)), "graph_end")
    assert graph.value
    try:
        for frame in range(3):
            a, b = _blocks(8, count, 4000 + frame, "random")
            res = rng(4100 + frame).integers(-255, 256, (n32, 1024)).astype(np.int16)
            pic, ref = me_frames(192, 128, 4200 + frame, (2 + frame, -1))
            for buf, arr in ((d_a, a), (d_b, b), (d_res, res), (d_pic, pic), (d_ref, ref)):
                arr = np.ascontiguousarray(arr)
                _lib.check(L.kvz_hip_memcpy_h2d(buf.ptr, arr.ctypes.data, arr.nbytes, s), "h2d")
            _lib.check(L.kvz_hip_stream_sync(s), "sync")
            _lib.check(L.kvz_hip_graph_launch(graph, s), "graph_launch")
            np.testing.assert_array_equal(d_sad.to_numpy(np.uint32, (count,), s), O.cost_nxn_batch("sad", 8, a, b))
            np.testing.assert_array_equal(d_satd.to_numpy(np.uint32, (count,), s), O.cost_nxn_batch("satd", 8, a, b))
            np.testing.assert_array_equal(d_coef.to_numpy(np.int16, (n32, 1024), s), O.transform_batch("dct", 32, res))
            got = d_out.to_numpy(np.int32, (len(pus), 8), s).view(ME_RESULT).reshape(-1)
            want = O.search_pu_batch(pic, ref, pus, prm)
            for f in ("mv", "cost", "bitcost"):
                np.testing.assert_array_equal(got[f], want[f], err_msg="%s frame %d" % (f, frame))
    finally:
        L.kvz_hip_graph_destroy(graph)
        L.kvz_hip_event_destroy(fork); L.kvz_hip_event_destroy(join)
        L.kvz_hip_stream_destroy(s); L.kvz_hip_stream_destroy(side)
    assert L.kvz_hip_graph_launch(None, None) != 0 and b"kvz_hip_graph_launch" in L.kvz_hip_last_error()


def test_search_pu_kernel_variants(api):
    """the search kernels exist in two builds -- with the fracmv_within_tile rule compiled in (WPP / OWF availability or an
    mv_constraint active) and without it (the common case, lighter on scalar registers); a constraint that can never bind must give the
    unconstrained result through the other build"""
    pic, ref = me_frames(192, 128, 77, (5, -3))
    pus = me_random_pus(192, 128, 50, 91, sizes=((32, 32), (32, 16), (24, 32), (64, 64), (64, 32), (48, 64), (16, 16), (8, 8)))
    pus["x"] = np.clip(pus["x"], 64, 192 - 64 - pus["width"]); pus["y"] = np.clip(pus["y"], 32, 128 - 32 - pus["height"])
    pus["extra_mv"] = 0
    pus["mv_cand"] = np.clip(pus["mv_cand"], -8, 8)
    pus["merge"]["mv"] = np.clip(pus["merge"]["mv"], -8, 8)
    prm = me_params(max_steps=2, early_termination=0)
    plain = api.search_pu_batch(pic, ref, pus, prm).view(ME_RESULT).reshape(-1)
    want = O.search_pu_batch(pic, ref, pus, prm)
    loose = api.search_pu_batch(pic, ref, pus, me_params(max_steps=2, early_termination=0, mv_constraint=1)).view(ME_RESULT).reshape(-1)
    for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
        np.testing.assert_array_equal(plain[f], want[f], err_msg=f)
        np.testing.assert_array_equal(loose[f], want[f], err_msg="constrained build, %s" % f)


# ---- deblocking ----
from patterns import deblock_case, deblock_params  # noqa: E402

DEBLOCK_CONFIGS = [dict(w=192, h=128, qp=34), dict(w=200, h=136, qp=38, beta=2, tc=-1), dict(w=128, h=64, qp=30, per_cu_qp=1),
                   dict(w=192, h=128, qp=36, slice_is_b=1), dict(w=136, h=72, qp=45, tc=3, per_cu_qp=1, slice_is_b=1),
                   dict(w=64, h=64, qp=22, beta=-3), dict(w=192, h=64, qp=40, chroma=0), dict(w=72, h=200, qp=51, beta=6, tc=6),
                   dict(w=8, h=8, qp=40), dict(w=640, h=360, qp=37, per_cu_qp=1, slice_is_b=1)]


@pytest.mark.parametrize("cfg", range(len(DEBLOCK_CONFIGS)))
def test_deblock_frame(api, cfg):
    c = dict(DEBLOCK_CONFIGS[cfg])
    w, h = c.pop("w"), c.pop("h")
    prm = deblock_params(**c)
    chroma = bool(prm["chroma"][0])
    for seed in range(3):
        y, u, v, cus = deblock_case(w, h, 300 * cfg + seed, slice_is_b=int(prm["slice_is_b"][0]), qp=int(prm["qp"][0]),
                                    intra_share=(0.35, 0.0, 1.0)[seed])
        want = O.deblock_frame(y, u if chroma else None, v if chroma else None, cus, prm)
        got = api.deblock_frame(y, u if chroma else None, v if chroma else None, cus, prm)
        np.testing.assert_array_equal(got[0], want[0], err_msg="luma cfg %d seed %d" % (cfg, seed))
        if chroma:
            np.testing.assert_array_equal(got[1], want[1], err_msg="u cfg %d seed %d" % (cfg, seed))
            np.testing.assert_array_equal(got[2], want[2], err_msg="v cfg %d seed %d" % (cfg, seed))


def test_deblock_argument_errors(api):
    from kvazaar_amd import _lib
    L = _lib.init(0)
    prm = deblock_params()
    assert L.kvz_hip_deblock_frame(None, 64, None, None, 32, 64, 64, None, prm.ctypes.data, None) != 0
    assert b"kvz_hip_deblock_frame" in L.kvz_hip_last_error()


def test_pair_kernels_wave_and_lane_groups(api):
    """frame-sized batches launch one wave per descriptor and every group of 8 descriptors picks its own mode: groups of
    large pairs (a wave each), groups of small ones (8 lanes each), mixed groups, a ragged last group -- and the grid-stride
    kernel for large batches (forced through the tuning knob) must agree with all of them"""
    from kvazaar_amd import _lib
    L = _lib.init(0)
    g = rng(77)
    pic = g.integers(0, 256, (160, 256), dtype=np.uint8)
    ref = g.integers(0, 256, (160, 256), dtype=np.uint8)
    pairs = []
    def add(n, sizes):
        for _ in range(n):
            w, h = sizes[int(g.integers(0, len(sizes)))]
            x1, y1 = int(g.integers(0, 256 - w + 1)), int(g.integers(0, 160 - h + 1))
            pairs.append((x1, y1, x1 + int(g.integers(-9, 10)), y1 + int(g.integers(-9, 10)), w, h))
    add(16, ((64, 64), (32, 32), (64, 32), (48, 40)))          # two all-large groups
    add(16, ((8, 8), (16, 16), (4, 4), (12, 8)))               # two all-small groups
    add(24, ((64, 64), (8, 8), (32, 32), (16, 8), (24, 24)))   # mixed groups
    add(5, ((64, 64), (32, 64)))                               # ragged, large
    want_sad = [O.image_calc("sad", pic, ref, *p) for p in pairs]
    want_satd = [O.image_calc("satd", pic, ref, *p) for p in pairs]
    inside = [p for p in pairs if 0 <= p[2] and p[2] + p[4] <= 256 and 0 <= p[3] and p[3] + p[5] <= 160 and p[4] == p[5]]
    want_ssd = [O.pixels_calc_ssd(pic, p[1] * 256 + p[0], ref, p[3] * 256 + p[2], 256, 256, p[4]) for p in inside]
    for knob in (1, 0):
        assert L.kvz_hip_set_tuning(b"pair_wave_kernel", knob) == 0
        try:
            np.testing.assert_array_equal(api.image_calc_sad_batch(pic, ref, pairs), want_sad, err_msg="sad knob %d" % knob)
            np.testing.assert_array_equal(api.image_calc_satd_batch(pic, ref, pairs), want_satd, err_msg="satd knob %d" % knob)
            np.testing.assert_array_equal(api.pixels_calc_ssd_batch(pic, ref, inside), want_ssd, err_msg="ssd knob %d" % knob)
        finally:
            L.kvz_hip_set_tuning(b"pair_wave_kernel", -1)


@pytest.mark.parametrize("kind", ["luma", "luma14", "chroma", "chroma14"])
def test_sample_filters_small_and_odd_blocks(api, kind):
    """many blocks of at most 8x8 (the dot-product path), odd shapes that take the generic path, windows that leave the
    frame, mixed with 16-pixel blocks, a ragged tail"""
    g = rng(62)
    frame = g.integers(0, 256, (48, 64), dtype=np.uint8)
    pad = 40
    padded = np.pad(frame, pad, mode="edge")
    nfrac = 4 if kind.startswith("luma") else 8
    blocks = []
    for k in range(43):
        w, h = ((8, 8), (4, 4), (8, 4), (4, 8), (8, 8), (2, 2), (6, 6), (8, 2))[k % 8]
        if 16 <= k < 24:
            w, h = ((16, 16), (8, 8), (16, 8), (8, 8), (8, 8), (8, 16), (8, 8), (12, 8))[k - 16]      # mixed runs
        blocks.append((int(g.integers(-6, 64)), int(g.integers(-6, 48)), int(g.integers(0, nfrac)), int(g.integers(0, nfrac)), w, h))
    got = api.sample_batch(kind, frame, blocks)
    for b, o in zip(blocks, got):
        x, y, fx, fy, w, h = b
        np.testing.assert_array_equal(o, O.sample(kind, padded, x + pad, y + pad, w, h, fx, fy), err_msg=str(b))


AMP_SMP_SHAPES = ((8, 4), (4, 8), (16, 4), (4, 16), (16, 12), (12, 16), (8, 8), (16, 16), (32, 8), (24, 32))


@pytest.mark.parametrize("cfg", [0, 3, 4, 6, 8, 11, 14])
def test_search_pu_amp_smp_shapes(api, cfg):
    """PU shapes with a dimension that is 4 mod 8 (--smp, --amp): dword row segments in the SAD rounds, satd_any_size's
    4x4 blocks for the integer position, satd_any_size_quad's origin-anchored 8x8 grid (possibly empty) for the candidates"""
    prm = me_params(**ME_CONFIGS[cfg])
    for k, motion in enumerate(((3, -2), (-6, 5), (0, 0))):
        pic, ref = me_frames(192, 128, 950 + k, motion)
        pus = me_random_pus(192, 128, 64, 31 + 10 * cfg + k, hint=(-4 * motion[0] + 1, -4 * motion[1]), sizes=AMP_SMP_SHAPES)
        pus["x"] = (pus["x"] // 4) * 4 + 4 * (np.arange(len(pus)) % 2)
        pus["x"] = np.minimum(pus["x"], 192 - pus["width"])
        _me_compare(api, pic, ref, pus, prm, "amp/smp cfg %d motion %s" % (cfg, motion))
    # vectors that leave the frame with a 4-wide block
    pic, ref = me_frames(192, 128, 33, (18, -15))
    pus = me_random_pus(192, 128, 40, 8, sizes=((4, 8), (4, 16), (12, 16), (8, 4), (16, 4), (16, 12)))
    pus["x"] = np.where(np.arange(40) % 2 == 0, 0, 192 - pus["width"])
    pus["y"] = np.where(np.arange(40) % 3 == 0, 0, 128 - pus["height"])
    pus["extra_mv"] = np.array([[-130, 90], [150, -120]] * 20, dtype=np.int16)
    _me_compare(api, pic, ref, pus, prm, "amp/smp borders cfg %d" % cfg)


# ---- context: stream ordering and several contexts per process (include/kvz_hip.h, "context") ----
def test_null_stream_is_ordered_after_the_default_stream_producer(api):
    """The round-1 failure (gpurun_out/full_test.log): inputs produced by torch on the legacy default stream, then a
    NULL-stream entry with NO synchronisation in between.  The library's default stream is a blocking stream, so the entry
    must see the finished inputs; and a consumer on the default stream must see the entry's results."""
    import torch
    from kvazaar_amd import _lib
    L = _lib.load()
    dev = torch.device("cuda", 0)
    n = 1 << 20
    for rep in range(4):
        g = torch.Generator(device=dev); g.manual_seed(100 + rep)
        base = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device=dev, generator=g)
        # a long producer chain on torch's (= the legacy default) stream, still running when the entry is enqueued
        cur = base
        for _ in range(6):
            cur = (cur.to(torch.int16) * 3 + 7).remainder(256).to(torch.uint8)
        ref = (cur.to(torch.int16) + 5).clamp_(0, 255).to(torch.uint8)
        sad = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(L.kvz_hip_sad_nxn_batch(8, cur.data_ptr(), ref.data_ptr(), n, sad.data_ptr(), None), "sad on the NULL stream")
        total = sad.to(torch.int64).sum()                       # consumer on the default stream, again without a sync
        want = (cur.to(torch.int16) - ref.to(torch.int16)).abs().sum(dim=1, dtype=torch.int32)
        assert bool((sad == want).all()) and int(total) == int(want.to(torch.int64).sum())


def test_contexts_per_device_and_thread_binding(api):
    """kvz_hip_init(d) is per device; an index beyond the device count is refused (it used to return OK and run on
    device 0); threads bind themselves with kvz_hip_set_device and work concurrently"""
    import threading
    from kvazaar_amd import _lib
    L = _lib.load()
    ndev = L.kvz_hip_device_count()
    assert ndev >= 1
    assert L.kvz_hip_init(ndev) == -2 and b"out of range" in L.kvz_hip_last_error()
    assert L.kvz_hip_init(0) == 0 and L.kvz_hip_get_device() == 0
    assert L.kvz_hip_abi_version() == 4
    errs = []

    def work(seed, device):
        try:
            assert L.kvz_hip_set_device(device) == 0 and L.kvz_hip_get_device() == device
            a, b = _blocks(16, 3000, seed, "near")
            for _ in range(5):
                np.testing.assert_array_equal(api.cost_nxn_batch("satd", 16, a, b), O.cost_nxn_many("satd", 16, a, b, threads=1))
        except Exception as e:                      # noqa: BLE001 -- reported by the main thread
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(50 + i, i % ndev)) for i in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs


def test_halo_exchange_through_the_c_abi(api):
    """kvz_hip_halo_exchange / kvz_hip_memcpy_peer: three row shards of one plane held by one process, each pushing its edge rows
    into its neighbours' halos -- on different devices when the box has several (device i % count, peer access enabled on first
    use), on one device otherwise; afterwards every extended buffer equals the matching rows of the whole plane.  A stream of
    another device and a calling thread on a third device are refused."""
    import ctypes as C
    from kvazaar_amd import _lib, shard as S
    L = _lib.load()
    ndev = L.kvz_hip_device_count()
    W, H, margin, world = 256, 64 * 7 + 24, 40, 3
    plane = rng(77).integers(0, 256, (H, W), dtype=np.uint8)

    class ShardPlane(C.Structure):
        _fields_ = [("ext", C.c_void_p), ("device", C.c_int32), ("top", C.c_int32), ("rows", C.c_int32)]
    shards, bufs, recs = [], [], []
    for r in range(world):
        sh = S.RowShard(W, H, world, r, margin)
        dev = r % ndev
        assert L.kvz_hip_set_device(dev) == 0
        ext = np.zeros((sh.ext_rows, W), np.uint8)
        ext[sh.top:sh.top + sh.rows] = plane[sh.y_lo:sh.y_hi]
        b = api.DeviceBuffer.from_numpy(ext)
        shards.append(sh); bufs.append(b); recs.append(ShardPlane(b.ptr, dev, sh.top, sh.rows))
    for r in range(world):
        assert L.kvz_hip_set_device(recs[r].device) == 0
        up = C.byref(recs[r - 1]) if r > 0 else None
        down = C.byref(recs[r + 1]) if r < world - 1 else None
        _lib.check(L.kvz_hip_halo_exchange(C.byref(recs[r]), up, down, W, margin, None), "halo_exchange")
        _lib.check(L.kvz_hip_stream_sync(None), "sync")
    for r in range(world):
        assert L.kvz_hip_set_device(recs[r].device) == 0
        got = bufs[r].to_numpy(np.uint8, (shards[r].ext_rows, W))
        np.testing.assert_array_equal(got, plane[shards[r].ext_lo:shards[r].ext_hi], err_msg="shard %d" % r)
    # refusals: the calling thread must work on self->device; a shard thinner than the margin
    assert L.kvz_hip_set_device(recs[0].device) == 0
    thin = ShardPlane(bufs[0].ptr, recs[0].device, 0, margin - 1)
    assert L.kvz_hip_halo_exchange(C.byref(thin), None, C.byref(recs[1]), W, margin, None) == -2
    if ndev >= 2:
        assert L.kvz_hip_set_device(1) == 0
        assert L.kvz_hip_halo_exchange(C.byref(recs[0]), None, C.byref(recs[1]), W, margin, None) == -2      # recs[0] lives on device 0
        st1 = L.kvz_hip_stream_create()                                                                       # a stream of device 1 ...
        assert L.kvz_hip_set_device(0) == 0
        assert L.kvz_hip_memcpy_peer(bufs[1].ptr, 1, bufs[0].ptr, 0, 16, st1) == -2                           # ... handed in on device 0
        assert b"stream belongs to device" in L.kvz_hip_last_error()
        L.kvz_hip_stream_destroy(st1)
    assert L.kvz_hip_set_device(0) == 0


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_transform_skip_kinds(api, n):
    """KVZ_HIP_TRSKIP / KVZ_HIP_ITRSKIP = kvz_transformskip / kvz_itransformskip (transform.c:150-180): a shift by
    15 - 8 - log2(n) with the reference's int16 stores; pinned through the oracle's quantize_residual (trskip path) below
    and, closed form, here"""
    g = rng(300 + n)
    x = g.integers(-32768, 32768, (257, n * n)).astype(np.int16)
    shift = 15 - 8 - {4: 2, 8: 3, 16: 4, 32: 5}[n]
    np.testing.assert_array_equal(api.transform_batch("trskip", n, x), (x.astype(np.int32) << shift).astype(np.int16))
    np.testing.assert_array_equal(api.transform_batch("itrskip", n, x), ((x.astype(np.int32) + (1 << (shift - 1))) >> shift).astype(np.int16))
    # the chain residual -> trskip -> quant -> dequant -> itrskip -> reconstruct equals the fused trskip path of the oracle (4x4, the only
    # size the encoder uses it for)
    if n == 4:
        ref = g.integers(0, 256, (500, 16), dtype=np.uint8)
        pred = np.clip(ref.astype(np.int32) + g.integers(-40, 41, ref.shape), 0, 255).astype(np.uint8)
        rec, coef, has = O.quantize_residual_many(ref, pred, 4, 30, 0, 0, 0, use_trskip=1)
        res = ref.astype(np.int16) - pred.astype(np.int16)
        q = api.quant_batch(api.transform_batch("trskip", 4, res), 4, 30, 0, 0)
        np.testing.assert_array_equal(q, coef)
        back = api.transform_batch("itrskip", 4, api.dequant_batch(q, 4, 30, 0))
        want = np.clip((back.astype(np.int32) + pred).astype(np.int16), 0, 255).astype(np.uint8)
        nz = (q != 0).any(axis=1)
        np.testing.assert_array_equal(rec[nz], want[nz])


# ---- AMVP / merge candidate derivation (the driver half) ----
from patterns import INTER_CAND_CONFIGS, ME_PU as ME_PU_DT, inter_cand_case  # noqa: E402


@pytest.mark.parametrize("name", [c[0] for c in INTER_CAND_CONFIGS])
def test_inter_candidates(api, name):
    """every inter PU of seeded random CU maps: completed search descriptors and merge lists, byte for byte"""
    for seed in range(1, 5):
        p, cus, col, refm, pus = inter_cand_case(name, seed)
        want_pus, want_merge = O.inter_candidates(p, cus, col, refm, pus)
        got_pus, got_merge = api.inter_candidates_batch(p, cus, col, refm, pus)
        np.testing.assert_array_equal(got_pus.view(ME_PU_DT).ravel(), want_pus, err_msg="%s seed %d" % (name, seed))
        np.testing.assert_array_equal(got_merge.ravel(), want_merge.view(np.uint8).ravel(), err_msg="%s seed %d" % (name, seed))
    # without the searched picture's CU array the start vector stays zero, everything else is unchanged
    got2, _ = api.inter_candidates_batch(p, cus, col, None, pus)
    want2 = want_pus.copy()
    want2["extra_mv"] = 0
    np.testing.assert_array_equal(got2.view(ME_PU_DT).ravel(), want2)


def test_inter_candidates_bad_descriptors_and_arguments(api):
    from kvazaar_amd._lib import KvzHipError
    p, cus, col, refm, pus = inter_cand_case("p_one_ref", 0)
    bad = pus[:6].copy()
    bad["x"][0] = -8
    bad["y"][1] = 2
    bad["width"][2] = 0
    bad["x"][3], bad["width"][3] = 160, 16                 # leaves the 168-wide picture
    bad["height"][4] = 68
    got, merge = api.inter_candidates_batch(p, cus, col, refm, bad)
    got = got.view(ME_PU_DT).ravel()
    assert (got["num_merge_cand"][:5] == -1).all() and got["num_merge_cand"][5] == 5
    assert not merge[:5].any() and not got["mv_cand"][:5].any()
    want, _ = O.inter_candidates(p, cus, col, refm, pus[5:6])
    np.testing.assert_array_equal(got[5:6], want)
    for field, value in (("num_refs", 17), ("ref_idx", 16), ("pic_width", 0), ("cus_stride", 8), ("in_width", 64)):
        q = p.copy()
        q[field] = value
        with pytest.raises(KvzHipError):
            api.inter_candidates_batch(q, cus, col, refm, pus)
    q = p.copy()
    q["ref_LX"][0, 0, 0] = 16
    with pytest.raises(KvzHipError):
        api.inter_candidates_batch(q, cus, col, refm, pus)
    q = p.copy()
    q["ref_LX"][0, 0, 1:], q["ref_LX"][0, 1, :], q["col_ref_LX"][0, :, 1:] = 255, 255, 255     # past the lists' ends: never read
    np.testing.assert_array_equal(api.inter_candidates_batch(q, cus, col, refm, pus)[0], api.inter_candidates_batch(p, cus, col, refm, pus)[0])
    with pytest.raises(KvzHipError):
        api.inter_candidates_batch(p, cus, None, refm, pus)         # tmvp on, references present: the collocated CUs are needed
    assert api.inter_candidates_batch(p, cus, col, refm, pus[:0])[0].shape == (0, 64)


def test_candidates_feed_the_search_on_the_device(api):
    """derive -> search chained on one stream through device buffers only: the descriptors never visit the host between the two
    entries; same vectors and costs as the oracle's search on the oracle's candidates"""
    from kvazaar_amd import _lib
    from kvazaar_amd.api import DeviceBuffer, check
    L = _lib.init()
    p, cus, col, refm, pus = inter_cand_case("p_one_ref", 7)
    w, h = int(p["pic_width"][0]), int(p["pic_height"][0])
    keep = [i for i in range(len(pus)) if (pus[i]["width"] % 8 == 0 or pus[i]["height"] % 8 == 0)]      # not 4x4-mod-8 in both (kvz_hip_me_pu)
    pus = pus[keep]
    cur, ref = me_frames(w, h, 31)
    prm = me_params(lambda_cost=22)
    d_cus, d_col, d_pus = DeviceBuffer.from_numpy(cus), DeviceBuffer.from_numpy(col), DeviceBuffer.from_numpy(pus)
    d_cur, d_ref = DeviceBuffer.from_numpy(cur), DeviceBuffer.from_numpy(ref)
    d_res = DeviceBuffer(32 * len(pus))
    st = L.kvz_hip_stream_create()
    try:
        check(L.kvz_hip_inter_candidates_batch(d_cus.ptr, d_col.ptr, d_col.ptr, np.ascontiguousarray(p).ctypes.data, d_pus.ptr, len(pus), None, st), "candidates")
        check(L.kvz_hip_search_pu_batch(d_cur.ptr, w, w, h, d_ref.ptr, w, w, h, d_pus.ptr, len(pus), np.ascontiguousarray(prm).ctypes.data, d_res.ptr, st), "search")
        got = d_res.to_numpy(np.int32, (len(pus), 8), stream=st)
    finally:
        L.kvz_hip_stream_destroy(st)
    want_pus, _ = O.inter_candidates(p, cus, col, col, pus)
    want = O.search_pu_batch(cur, ref, want_pus, prm)
    np.testing.assert_array_equal(got, np.asarray(want).view(np.int32).reshape(len(pus), 8))


@pytest.mark.parametrize("cfg", [0, 1, 4, 8, 11, 14, 19])
def test_search_pu_with_a_cost_to_beat(api, cfg):
    """kvz_hip_me_params.cost_to_beat: the later pictures of a multi-reference frame (search_inter.c:1239-1252), every size class"""
    from patterns import cost_to_beat_case
    prm = me_params(**ME_CONFIGS[cfg])
    for k, motion in enumerate(((3, -2), (-7, 5))):
        pic, ref = me_frames(192, 128, 910 + k, motion)
        pus = me_pus_in_tile(me_random_pus(192, 128, 120, 177 + 10 * cfg + k, hint=(-4 * motion[0] + 2, -4 * motion[1])), prm)
        free = O.search_pu_batch(pic, ref, pus, prm)
        beat = cost_to_beat_case(free["cost"], 3 * cfg + k)
        want = O.search_pu_batch(pic, ref, pus, prm, cost_to_beat=beat)
        got = api.search_pu_batch(pic, ref, pus, prm, cost_to_beat=beat)
        np.testing.assert_array_equal(got, np.asarray(want).view(np.int32).reshape(len(pus), 8), err_msg="cfg %d motion %s" % (cfg, motion))
        for hint in (1, 2, 4):                               # one class per launch: the limit is indexed by the PU's place in the batch
            q = prm.copy()
            q["size_classes"] = hint
            sel = [i for i in range(len(pus)) if (1 if max(pus[i]["width"], pus[i]["height"]) <= 16 else 2 if max(pus[i]["width"], pus[i]["height"]) <= 32 else 4) == hint]
            got_h = api.search_pu_batch(pic, ref, pus, q, cost_to_beat=beat)
            np.testing.assert_array_equal(got_h[sel], got[sel])


@pytest.mark.parametrize("cfg", [0, 4, 8, 14, 19, 22])
def test_search_pu_over_several_pictures_in_one_launch(api, cfg):
    """kvz_hip_search_pu_multi_batch: PUs of five picture pairs interleaved in one launch (every size class, with and without the
    hint) against one launch per pair; a plane index beyond the table flags its PU"""
    prm = me_params(**ME_CONFIGS[cfg])
    pairs = [me_frames(192, 128, 700 + k, motion) for k, motion in enumerate(((3, -2), (-7, 5), (0, 0), (11, 6), (-2, -9)))]
    pus = me_pus_in_tile(me_random_pus(192, 128, 200, 55 + cfg, hint=(-10, 8)), prm)
    owner = np.arange(len(pus)) % 5
    pus["pad"] = (owner << 2) | (np.arange(len(pus)) % 4)            # bits 0-1 are the candidate entry's, ignored here
    want = np.zeros((len(pus), 8), np.int32)
    for k, (pic, ref) in enumerate(pairs):
        sel = np.where(owner == k)[0]
        want[sel] = api.search_pu_batch(pic, ref, pus[sel], prm)
        np.testing.assert_array_equal(want[sel], np.asarray(O.search_pu_batch(pic, ref, pus[sel], prm)).view(np.int32).reshape(len(sel), 8))
    pics, refs = [p for p, _ in pairs], [r for _, r in pairs]
    np.testing.assert_array_equal(api.search_pu_multi_batch(pics, refs, pus, prm), want)
    for hint in (1, 2, 4):
        q = prm.copy()
        q["size_classes"] = hint
        sel = [i for i in range(len(pus)) if (1 if max(pus[i]["width"], pus[i]["height"]) <= 16 else 2 if max(pus[i]["width"], pus[i]["height"]) <= 32 else 4) == hint]
        np.testing.assert_array_equal(api.search_pu_multi_batch(pics, refs, pus, q)[sel], want[sel])
    bad = pus[:8].copy()
    bad["pad"][3] = 5 << 2
    bad["pad"][5] = -4
    got = api.search_pu_multi_batch(pics, refs, bad, prm)
    assert got[3, 7] == -1 and got[5, 7] == -1 and (got[[0, 1, 2, 4, 6, 7], 7] == 0).all()


def test_inter_candidates_of_several_pictures_in_one_launch(api):
    """kvz_hip_inter_candidates_multi_batch: PUs of six pictures (P and B slices, one to four references, a tile, different sizes)
    interleaved in one launch against one launch per picture; a picture the one-picture entry would refuse, and an index beyond the
    table, give num_merge_cand -1 for their PUs only"""
    names = ["p_one_ref", "p_three_refs", "b_two_sided", "b_hier", "p_tile", "p_no_tmvp"]
    cases = [inter_cand_case(n, 3) for n in names]
    parts, owners, want_pus, want_merge = [], [], [], []
    for k, (p, cus, col, refm, pus) in enumerate(cases):
        pus = pus[:60].copy()
        pus["pad"] = (k << 2) | (pus["pad"] & 3)
        a, b = api.inter_candidates_batch(p, cus, col, refm, pus)
        parts.append(pus); owners += [k] * len(pus); want_pus.append(a.view(ME_PU_DT).ravel()); want_merge.append(b)
    order = np.random.default_rng(4).permutation(len(owners))
    allpus = np.concatenate(parts)[order]
    want_pus, want_merge = np.concatenate(want_pus)[order], np.concatenate(want_merge)[order]
    pictures = [(p, cus, col, refm) for (p, cus, col, refm, _) in cases]
    got_pus, got_merge = api.inter_candidates_multi_batch(pictures, allpus)
    np.testing.assert_array_equal(got_pus.view(ME_PU_DT).ravel(), want_pus)
    np.testing.assert_array_equal(got_merge, want_merge)
    # picture 1 made unusable (17 references), one PU pointing past the table
    broken = [(p.copy(), cus, col, refm) for (p, cus, col, refm) in pictures]
    broken[1][0]["num_refs"] = 17
    stray = allpus.copy()
    stray["pad"][0] = 6 << 2
    got2 = api.inter_candidates_multi_batch(broken, stray)[0].view(ME_PU_DT).ravel()
    owner = np.array(owners)[order]
    assert got2["num_merge_cand"][0] == -1 and (got2["num_merge_cand"][owner == 1] == -1).all()
    keep = (owner != 1) & (np.arange(len(owner)) != 0)
    np.testing.assert_array_equal(got2[keep], want_pus[keep])
    # a list entry beyond 15 in one picture (what the one-picture entry refuses on the host): that picture's PUs are flagged, the
    # entry is not masked into range, the other pictures are derived as before
    bad_list = [(p.copy(), cus, col, refm) for (p, cus, col, refm) in pictures]
    bad_list[2][0]["ref_LX"][0, 0, 0] = 16
    got3 = api.inter_candidates_multi_batch(bad_list, allpus)[0].view(ME_PU_DT).ravel()
    assert (got3["num_merge_cand"][owner == 2] == -1).all()
    np.testing.assert_array_equal(got3[owner != 2], want_pus[owner != 2])
    with pytest.raises(Exception):
        api.inter_candidates_batch(bad_list[2][0], cases[2][1], cases[2][2], cases[2][3], cases[2][4][:4])
