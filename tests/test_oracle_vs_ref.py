"""Pins the oracle (oracle/kvz_oracle.c) against the COMPILED REFERENCE
(oracle/_ref/libkvzref.so = /root/reference built by oracle/Makefile), function
by function, on the reference's own test patterns plus random and adversarial
inputs.  CPU only.  Skipped when the prebuilt reference library is absent."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import ref_lib as R
from patterns import intra_ref_cases, intra_ref_positions, lcg_bytes, rng

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")

SIZES = (4, 8, 16, 32, 64)


def _blocks(n, count, seed, mode):
    g = rng(seed)
    if mode == "random":
        a = g.integers(0, 256, (count, n * n), dtype=np.uint8)
        b = g.integers(0, 256, (count, n * n), dtype=np.uint8)
    elif mode == "extreme":
        a = np.zeros((count, n * n), np.uint8)
        b = np.full((count, n * n), 255, np.uint8)
        a[1::2], b[1::2] = 255, 0
    else:  # near: small differences (typical ME residual)
        a = g.integers(0, 256, (count, n * n), dtype=np.uint8)
        b = np.clip(a.astype(np.int32) + g.integers(-6, 7, a.shape), 0, 255).astype(np.uint8)
    return a, b


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("mode", ["random", "extreme", "near"])
@pytest.mark.parametrize("kind", ["sad", "satd"])
def test_cost_nxn(kind, n, mode):
    a, b = _blocks(n, 24, 100 + n, mode)
    np.testing.assert_array_equal(O.cost_nxn_batch(kind, n, a, b), R.cost_nxn_batch(kind, n, a, b))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kind", ["sad", "satd"])
def test_cost_nxn_dual(kind, n):
    g = rng(7 + n)
    if n > 32:
        pytest.skip("pred_buffer holds 32x32 (strategies-picture.h:34); 64x64_dual overlaps it")
    orig = g.integers(0, 256, (16, n * n), dtype=np.uint8)
    preds = g.integers(0, 256, (16, 2048), dtype=np.uint8)
    np.testing.assert_array_equal(O.cost_nxn_dual_batch(kind, n, preds, orig),
                                  R.cost_nxn_dual_batch(kind, n, preds, orig))


def test_reg_sad_shapes():
    g = rng(3)
    a = g.integers(0, 256, 64 * 80, dtype=np.uint8)
    b = g.integers(0, 256, 96 * 80, dtype=np.uint8)
    dims = [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (32, 64), (32, 16), (16, 32), (16, 8), (8, 16),
            (8, 4), (4, 8), (48, 16), (16, 48), (24, 16), (16, 24), (12, 4), (4, 12), (64, 63), (1, 1), (7, 3)]
    for (w, h) in dims:
        for (o1, o2) in ((0, 0), (5, 9), (64 * 3 + 1, 96 * 7 + 13)):
            assert O.reg_sad(a, b, o1, o2, w, h, 64, 96) == R.reg_sad(a, b, o1, o2, w, h, 64, 96)


def test_satd_any_size_shapes():
    g = rng(4)
    a = g.integers(0, 256, 80 * 80, dtype=np.uint8)
    b = g.integers(0, 256, 100 * 80, dtype=np.uint8)
    for w in (4, 8, 12, 16, 24, 32, 48, 64):
        for h in (4, 8, 12, 16, 24, 32, 48, 64):
            assert O.satd_any_size(w, h, a, 3, 80, b, 7, 100) == R.satd_any_size(w, h, a, 3, 80, b, 7, 100)


def test_satd_any_size_quad_incl_quirk():
    g = rng(5)
    preds = [g.integers(0, 256, 64 * 72, dtype=np.uint8) for _ in range(4)]
    orig = g.integers(0, 256, 100 * 80, dtype=np.uint8)
    for w in (4, 8, 12, 16, 24, 32, 64):
        for h in (4, 8, 12, 16, 24, 32, 64):
            o = O.satd_any_size_quad(w, h, preds, 64, orig, 11, 100)
            r = R.satd_any_size_quad(w, h, preds, 64, orig, 11, 100)
            np.testing.assert_array_equal(o, r, err_msg="w=%d h=%d" % (w, h))


def test_pixels_calc_ssd():
    g = rng(6)
    a = g.integers(0, 256, 70 * 70, dtype=np.uint8)
    b = g.integers(0, 256, 70 * 70, dtype=np.uint8)
    for w in (4, 8, 16, 32, 64):
        assert O.pixels_calc_ssd(a, 2, b, 5, 70, 66, w) == R.pixels_calc_ssd(a, 2, b, 5, 70, 66, w)
    z, m = np.zeros(64 * 64, np.uint8), np.full(64 * 64, 255, np.uint8)
    assert O.pixels_calc_ssd(z, 0, m, 0, 64, 64, 64) == R.pixels_calc_ssd(z, 0, m, 0, 64, 64, 64)


def test_image_calc_sad_and_satd_edges():
    g = rng(8)
    pic = g.integers(0, 256, (48, 64), dtype=np.uint8)
    ref = g.integers(0, 256, (48, 64), dtype=np.uint8)
    for (bw, bh) in ((8, 8), (16, 16), (16, 8), (32, 32)):
        for (px, py) in ((0, 0), (16, 8), (64 - bw, 48 - bh)):
            for (dx, dy) in ((0, 0), (-3, -3), (5, -70), (-100, 2), (70, 70), (3, 0), (0, 60), (-bw, -bh),
                             (64, 48), (63 - px, 47 - py)):
                args = (pic, ref, px, py, px + dx, py + dy, bw, bh)
                assert O.image_calc("sad", *args) == R.image_calc("sad", *args), args[2:]
                assert O.image_calc("satd", *args) == R.image_calc("satd", *args), args[2:]


@pytest.mark.parametrize("kind,n", [("dct", 4), ("dct", 8), ("dct", 16), ("dct", 32), ("idct", 4), ("idct", 8),
                                     ("idct", 16), ("idct", 32), ("dst", 4), ("idst", 4)])
def test_transform(kind, n):
    g = rng(20 + n)
    res = g.integers(-255, 256, (16, n * n)).astype(np.int16)          # encoder's residual domain
    full = g.integers(-32768, 32768, (16, n * n)).astype(np.int16)     # wrap / clip behaviour
    edge = np.array([[32767] * (n * n), [-32768] * (n * n),
                     [32767 if (i + i // n) % 2 else -32768 for i in range(n * n)]], dtype=np.int16)
    for x in (res, full, edge):
        np.testing.assert_array_equal(O.transform_batch(kind, n, x), R.transform_batch(kind, n, x))


def test_dct_matrix_generation_is_orthogonal_like():
    for n in (4, 8, 16, 32):
        M = O.dct_matrix(n).astype(np.int64)
        assert (M[0] == 64).all()
        G = M @ M.T
        off = G - np.diag(np.diag(G))
        assert np.abs(off).max() < 64 * 64 * n * 0.02   # HEVC matrices are near-orthogonal


@pytest.mark.parametrize("w", [4, 8, 16, 32])
@pytest.mark.parametrize("signhide", [0, 1])
def test_quant(w, signhide):
    g = rng(30 + w)
    coef = g.integers(-2000, 2001, (12, w * w)).astype(np.int16)
    coef[3] = g.integers(-32768, 32768, w * w)
    coef[4] = 0
    coef[5, ::7] = 1
    for qp in (0, 5, 22, 27, 37, 51):
        # 32x32 chroma TUs do not exist (4:2:0, max TU 32) and the reference has no
        # scaling-list table for them (scalinglist.c:72-93): luma only at w == 32
        for type_ in ((0,) if w == 32 else (0, 2)):
            for scan in (0, 1, 2):
                for intra_slice in (0, 1):
                    o = O.quant_batch(coef, w, qp, type_, scan, intra_slice, signhide)
                    r = R.quant_batch(coef, w, qp, type_, scan, intra_slice, signhide)
                    np.testing.assert_array_equal(o, r, err_msg="qp=%d type=%d scan=%d" % (qp, type_, scan))


@pytest.mark.parametrize("w", [4, 8, 16, 32])
def test_dequant(w):
    g = rng(40 + w)
    q = g.integers(-300, 301, (8, w * w)).astype(np.int16)
    q[2] = g.integers(-32768, 32768, w * w)
    for qp in (0, 7, 22, 36, 51):
        for type_ in ((0,) if w == 32 else (0, 2, 3)):
            np.testing.assert_array_equal(O.dequant_batch(q, w, qp, type_), R.dequant_batch(q, w, qp, type_))


@pytest.mark.parametrize("w", [4, 8, 16, 32])
def test_quant_dequant_scaling_list(w):
    """default (non-flat) scaling list: the oracle is fed the reference's processed tables"""
    g = rng(45 + w)
    log2 = {4: 2, 8: 3, 16: 4, 32: 5}[w]
    coef = g.integers(-3000, 3001, (6, w * w)).astype(np.int16)
    for qp in (10, 22, 33):
        for intra in (0, 1):
            list_type = (0 if intra else 3) + 0        # luma
            if log2 == 5:
                list_type = 0 if intra else 1          # 32x32: list 3 aliases list 1 (scalinglist.c:88-92)
            qt, dt = R.scaling_tables(log2, list_type, qp % 6, w)
            r = R.quant_batch(coef, w, qp, 0, 0, 0, 0, intra, sl=1)
            o = O.quant_batch(coef, w, qp, 0, 0, 0, 0, intra, quant_coeff=qt)
            np.testing.assert_array_equal(o, r)
            rd = R.dequant_batch(r, w, qp, 0, intra, sl=1)
            od = O.dequant_batch(r, w, qp, 0, intra, dequant_coeff=dt)
            np.testing.assert_array_equal(od, rd)


def test_coeff_abs_sum_kat():
    # tests/coeff_sum_tests.c:29-43
    c = (np.arange(64 * 64, dtype=np.int64) * 16 - 32768).astype(np.int16)
    expected = 2048 * (16 + 32768) // 2 + 2048 * 2047 * 16 // 2
    assert O.coeff_abs_sum(c) == expected == R.coeff_abs_sum(c)


@pytest.mark.parametrize("w", [4, 8, 16, 32])
def test_quantize_residual(w):
    g = rng(50 + w)
    ref_in = g.integers(0, 256, (10, w * w), dtype=np.uint8)
    pred = np.clip(ref_in.astype(np.int32) + g.integers(-40, 41, ref_in.shape), 0, 255).astype(np.uint8)
    pred[0] = ref_in[0]                  # zero residual => no coeffs path
    pred[1] = 255 - ref_in[1]            # large residual
    for qp in (12, 22, 32, 45):
        for color in ((0,) if w == 32 else (0, 1, 2)):
            for intra in (0, 1):
                for trskip in ((0, 1) if w == 4 else (0,)):
                    o = O.quantize_residual_batch(ref_in, pred, w, qp, color, 0, intra, intra, 0, trskip)
                    r = R.quantize_residual_batch(ref_in, pred, w, qp, color, 0, intra, intra, 0, trskip)
                    for a, b, nm in zip(o, r, ("rec", "coeff", "has")):
                        np.testing.assert_array_equal(a, b, err_msg="%s qp=%d color=%d intra=%d" % (nm, qp, color, intra))


@pytest.mark.parametrize("kind", ["luma", "luma14", "chroma", "chroma14"])
def test_sample_filters(kind):
    g = rng(60)
    frame = g.integers(0, 256, (96, 96), dtype=np.uint8)
    frame[40:60, 40:60] = np.where(g.integers(0, 2, (20, 20)) > 0, 255, 0)   # adversarial extremes
    nfrac = 4 if kind.startswith("luma") else 8
    sizes = ((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 4)) if kind.startswith("luma") else \
            ((4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (2, 2))
    for (w, h) in sizes:
        for fx in range(nfrac):
            for fy in range(nfrac):
                o = O.sample(kind, frame, 12, 10, w, h, fx, fy)
                r = R.sample(kind, frame, 12, 10, w, h, fx, fy)
                np.testing.assert_array_equal(o, r, err_msg="%s %dx%d frac=(%d,%d)" % (kind, w, h, fx, fy))


@pytest.mark.parametrize("pattern", ["random", "extreme"])
def test_frac_block_filters(pattern):
    g = rng(70)
    frame = g.integers(0, 256, (96, 96), dtype=np.uint8)
    if pattern == "extreme":
        frame = np.where(g.integers(0, 2, (96, 96)) > 0, 255, 0).astype(np.uint8)
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (16, 8)):
        for ox in (-1, 0, 1):
            for oy in (-1, 0, 1):
                o = O.filter_frac_steps(frame, 10, 9, w, h, (ox, oy))
                r = R.filter_frac_steps(frame, 10, 9, w, h, (ox, oy))
                np.testing.assert_array_equal(o[:, :, :h, :w], r[:, :, :h, :w],
                                              err_msg="%dx%d off=(%d,%d)" % (w, h, ox, oy))


def test_search_frac_costs():
    g = rng(80)
    ref = g.integers(0, 256, (72, 96), dtype=np.uint8)
    # pic = ref shifted by a sub-pel-ish blend so that fractional positions matter
    pic = ((ref.astype(np.int32) + np.roll(ref, 1, axis=1)) // 2).astype(np.uint8)
    # incl. the AMP / SMP shapes, whose candidates satd_any_size_quad scores on an origin-anchored (possibly empty) 8x8 grid
    for (w, h) in ((8, 8), (16, 16), (32, 32), (8, 4), (4, 8), (16, 4), (4, 16), (16, 12), (12, 16)):
        for (x, y) in ((0, 0), (32, 24), (96 - w, 72 - h)):
            for (mvx, mvy) in ((0, 0), (-2, 1), (5, -3), (-40, -40), (90, 70)):
                o = O.search_frac_costs(pic, ref, x, y, w, h, mvx, mvy)
                r = R.search_frac_costs(pic, ref, x, y, w, h, mvx, mvy)
                np.testing.assert_array_equal(o[0], r[0], err_msg=str((w, h, x, y, mvx, mvy)))
                assert o[1] == r[1]


def test_avx2_agrees_where_survey_says_so():
    """avx2 == generic for IDCT on any input and for forward DCT on 9-bit residuals."""
    if not R.has_strategy("dct_32x32", "avx2"):
        pytest.skip("host has no avx2")
    g = rng(90)
    for n in (4, 8, 16, 32):
        res = g.integers(-255, 256, (8, n * n)).astype(np.int16)
        np.testing.assert_array_equal(R.transform_batch("dct", n, res, "avx2"), O.transform_batch("dct", n, res))
        full = g.integers(-32768, 32768, (8, n * n)).astype(np.int16)
        np.testing.assert_array_equal(R.transform_batch("idct", n, full, "avx2"), O.transform_batch("idct", n, full))


def test_reference_encoder_harness_generic_equals_selected():
    """the end-to-end harness used by tests/test_gpu_dropin.py: installing the "generic" strategies into the reference
    encoder's table gives the same bitstream as the selector's own (avx2) choice -- the property the reference relies on"""
    frames = R.synthetic_sequence(128, 64, 3)
    a, n_inst = R.encode(frames, 128, 64, "preset=medium,rdoq=0,qp=30,threads=0", "generic")
    b, _ = R.encode(frames, 128, 64, "preset=medium,rdoq=0,qp=30,threads=0", None)
    assert n_inst >= 40 and len(a) > 200 and a == b


# ---- intra group: angular_pred / intra_pred_planar strategies and kvz_intra_predict ----
@pytest.mark.parametrize("log2_width", [2, 3, 4, 5])
def test_intra_strategies(log2_width):
    refs = intra_ref_cases(log2_width, 10, 300 + log2_width)
    for r in refs:
        left, top = r[:65], r[65:]
        for name in ("generic", "avx2"):
            if not R.has_strategy("angular_pred", name):
                continue
            for mode in range(2, 35):
                np.testing.assert_array_equal(O.angular_pred(log2_width, mode, top, left),
                                              R.angular_pred(log2_width, mode, top, left, name), err_msg="mode %d %s" % (mode, name))
            np.testing.assert_array_equal(O.intra_pred_planar(log2_width, top, left), R.intra_pred_planar(log2_width, top, left, name))


@pytest.mark.parametrize("log2_width", [2, 3, 4, 5])
@pytest.mark.parametrize("color,filter_boundary", [(0, 1), (0, 0), (1, 1)])
def test_intra_predict(log2_width, color, filter_boundary):
    refs = intra_ref_cases(log2_width, 10, 340 + log2_width)
    ours = O.intra_predict_batch(refs, log2_width, list(range(35)), is_luma=int(color == 0), filter_boundary=filter_boundary)
    for i, r in enumerate(refs):
        for mode in range(35):
            np.testing.assert_array_equal(ours[i, mode], R.intra_predict(r, log2_width, mode, color, filter_boundary),
                                          err_msg="ref %d mode %d" % (i, mode))


def test_intra_predict_on_built_references():
    """reference arrays as kvz_intra_build_reference makes them at picture corners / edges / inside an LCU"""
    g = rng(77)
    rec = g.integers(0, 256, 64 * 64, dtype=np.uint8)
    top, left = g.integers(0, 256, 97, dtype=np.uint8), g.integers(0, 256, 97, dtype=np.uint8)
    for log2_width in (2, 3, 4, 5):
        n = 1 << log2_width
        for (x, y) in ((0, 0), (64, 0), (0, 64), (64, 64), (64 + n, 64 + n), (128 - n, 64), (64, 128 - n)):
            r = R.intra_build_reference(log2_width, x, y, 128, 128, rec, top, left, 99)
            for mode in range(35):
                np.testing.assert_array_equal(O.intra_predict_batch(r, log2_width, [mode])[0, 0], R.intra_predict(r, log2_width, mode))


@pytest.mark.parametrize("color", [0, 1, 2])
@pytest.mark.parametrize("log2_width", [2, 3, 4, 5])
def test_intra_build_reference(log2_width, color):
    """the plane-based restatement against kvz_intra_build_reference on an lcu_t cut from the same plane, every PU
    position of a picture with a ragged last LCU column / row; the pixels the PU must not see are poisoned in the lcu_t"""
    g = rng(600 + 10 * log2_width + color)
    pic_w, pic_h = 168, 136
    c = 1 if color else 0
    plane = g.integers(0, 256, (pic_h >> c, pic_w >> c), dtype=np.uint8)
    xy = intra_ref_positions(log2_width, color, pic_w, pic_h)
    ours = O.intra_build_reference_batch(log2_width, color, plane, pic_w, pic_h, xy)
    n2 = 2 << log2_width
    for i, (x, y) in enumerate(xy):
        r = R.intra_build_reference_from_plane(log2_width, color, plane, pic_w, pic_h, x, y, poison=g)
        np.testing.assert_array_equal(ours[i, :n2 + 1], r[:n2 + 1], err_msg="left of (%d, %d)" % (x, y))
        np.testing.assert_array_equal(ours[i, 65:65 + n2 + 1], r[65:65 + n2 + 1], err_msg="top of (%d, %d)" % (x, y))


# ---- motion search: hexagon_search + search_frac with MV costs ----
from patterns import me_frames, me_params, me_pus_in_tile, me_random_pus  # noqa: E402

ME_CONFIGS = [
    dict(),                                                        # preset medium: hexbs, early termination on, subme 4
    dict(early_termination=2, fme_level=2, lambda_cost=35),        # ultrafast/veryfast style
    dict(early_termination=0, lambda_cost=4),
    dict(fme_level=0, lambda_cost=60),
    dict(fme_level=1), dict(fme_level=3, max_steps=2),
    dict(wpp_owf=1, ref_delay_px=10, max_ref_lcu_down=1, max_ref_lcu_right=1),
    dict(wpp_owf=1, ref_delay_px=8, max_ref_lcu_down=0, max_ref_lcu_right=2, lambda_cost=9),
    dict(algorithm=1), dict(algorithm=1, early_termination=0, max_steps=3, lambda_cost=50), dict(algorithm=1, fme_level=2, early_termination=2),
    dict(algorithm=2), dict(algorithm=2, early_termination=0, lambda_cost=6), dict(algorithm=2, wpp_owf=1, ref_delay_px=10, fme_level=3),
    dict(algorithm=3, search_range=8), dict(algorithm=3, search_range=16, lambda_cost=40, wpp_owf=1, ref_delay_px=8, fme_level=2),
    # kvz_mv_constraint (kvazaar.h:113-119), the branches of fracmv_within_tile search_inter.c:142-171: the frame as one tile ...
    dict(mv_constraint=1), dict(mv_constraint=2, lambda_cost=7, early_termination=0), dict(mv_constraint=3, algorithm=1),
    dict(mv_constraint=4), dict(mv_constraint=4, algorithm=2, fme_level=2), dict(mv_constraint=4, algorithm=3, search_range=8, lambda_cost=33),
    # ... and real tiles (state->tile->offset_x/_y, info->origin relative to the tile), alone and under the WPP / OWF rule
    dict(mv_constraint=3, tile=(64, 0, 128, 128)), dict(mv_constraint=4, tile=(0, 64, 192, 64), lambda_cost=11),
    dict(mv_constraint=4, tile=(64, 64, 64, 64), early_termination=0, fme_level=3),
    dict(mv_constraint=0, tile=(64, 0, 128, 128), wpp_owf=1, ref_delay_px=10, max_ref_lcu_down=1, max_ref_lcu_right=1),
    dict(mv_constraint=4, tile=(0, 64, 192, 64), wpp_owf=1, ref_delay_px=8, max_ref_lcu_down=0, max_ref_lcu_right=1, algorithm=1),
]


@pytest.mark.parametrize("cfg", range(len(ME_CONFIGS)))
def test_search_pu(cfg):
    prm = me_params(**ME_CONFIGS[cfg])
    for k, motion in enumerate(((3, -2), (-7, 5), (0, 0), (14, 9))):
        pic, ref = me_frames(192, 128, 900 + k, motion)
        pus = me_pus_in_tile(me_random_pus(192, 128, 40, 77 + 10 * cfg + k, hint=(-4 * motion[0] + 2, -4 * motion[1])), prm)
        a, b = O.search_pu_batch(pic, ref, pus, prm), R.search_pu_batch(pic, ref, pus, prm)
        for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
            np.testing.assert_array_equal(a[f], b[f], err_msg="%s cfg %d motion %s" % (f, cfg, motion))
    # flat frames: every candidate ties, the reference's first-wins order decides
    flat = np.full((128, 192), 77, np.uint8)
    pus = me_pus_in_tile(me_random_pus(192, 128, 12, 5), prm)
    a, b = O.search_pu_batch(flat, flat, pus, prm), R.search_pu_batch(flat, flat, pus, prm)
    np.testing.assert_array_equal(a.view(np.int32), b.view(np.int32))


@pytest.mark.parametrize("cfg", [0, 1, 4, 8, 11, 14, 19])
def test_search_pu_with_a_cost_to_beat(cfg):
    """search_pu_inter_ref for the second and later pictures of a multi-reference frame (search_inter.c:1239-1252): the fractional
    search runs only if the integer result beats *inter_cost, else the integer vector is re-scored with SATD"""
    from patterns import cost_to_beat_case
    prm = me_params(**ME_CONFIGS[cfg])
    changed = 0
    for k, motion in enumerate(((3, -2), (-7, 5), (0, 0))):
        pic, ref = me_frames(192, 128, 910 + k, motion)
        pus = me_pus_in_tile(me_random_pus(192, 128, 40, 177 + 10 * cfg + k, hint=(-4 * motion[0] + 2, -4 * motion[1])), prm)
        free = O.search_pu_batch(pic, ref, pus, prm)
        beat = cost_to_beat_case(free["cost"], 3 * cfg + k)
        a, b = O.search_pu_batch(pic, ref, pus, prm, cost_to_beat=beat), R.search_pu_batch(pic, ref, pus, prm, cost_to_beat=beat)
        for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
            np.testing.assert_array_equal(a[f], b[f], err_msg="%s cfg %d motion %s" % (f, cfg, motion))
        changed += int((a["mv"] != free["mv"]).any(axis=1).sum())
    assert changed > 5 or int(prm["fme_level"][0]) == 0          # the limit did change searches


AMP_SMP_SHAPES = ((8, 4), (4, 8), (16, 4), (4, 16), (16, 12), (12, 16), (8, 8), (16, 16))


@pytest.mark.parametrize("cfg", [0, 3, 4, 6, 8, 11, 14])
def test_search_pu_amp_smp_shapes(cfg):
    """PU shapes with a dimension that is 4 mod 8 (--smp at 8x8 CUs, --amp at 16x16): the integer position is scored by
    satd_any_size (4x4 blocks on the first 4-pixel column / row), the fractional candidates by satd_any_size_quad, whose
    4x4 stages add nothing -- the oracle must follow the reference through both"""
    prm = me_params(**ME_CONFIGS[cfg])
    for k, motion in enumerate(((3, -2), (-6, 5), (0, 0))):
        pic, ref = me_frames(192, 128, 950 + k, motion)
        pus = me_random_pus(192, 128, 48, 31 + 10 * cfg + k, hint=(-4 * motion[0] + 1, -4 * motion[1]), sizes=AMP_SMP_SHAPES)
        pus["x"] = (pus["x"] // 4) * 4 + 4 * (np.arange(len(pus)) % 2)            # 4-aligned origins, as the part offsets give
        pus["x"] = np.minimum(pus["x"], 192 - pus["width"])
        a, b = O.search_pu_batch(pic, ref, pus, prm), R.search_pu_batch(pic, ref, pus, prm)
        for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
            np.testing.assert_array_equal(a[f], b[f], err_msg="%s cfg %d motion %s" % (f, cfg, motion))


# ---- SAO group ----
from patterns import sao_blocks, sao_records  # noqa: E402


@pytest.mark.parametrize("bw,bh", [(64, 64), (32, 32), (64, 56), (16, 24), (8, 8), (3, 3), (40, 2)])
def test_sao_statistics_and_ddistortion(bw, bh):
    orig, rec = sao_blocks(bw, bh, 8, 40 + bw + bh)
    g = rng(5)
    for name in ("generic", "avx2"):
        if not R.has_strategy("sao_edge_ddistortion", name):
            continue
        if name != "generic" and min(bw, bh) < 4:
            continue        # frame dimensions are multiples of 8: the avx2 strategy is never handed such blocks
        for i in range(len(orig)):
            for eo in range(4):
                np.testing.assert_array_equal(O.calc_sao_edge_dir(orig[i], rec[i], eo, bw, bh), R.calc_sao_edge_dir(orig[i], rec[i], eo, bw, bh, name))
                offs = g.integers(-7, 8, 5)
                # callers always pass offsets[SAO_EO_CAT0] == 0 (sao.c:406-407); the avx2 strategy relies on it, generic
                # honours any value -- the oracle follows generic, so exercise a nonzero entry only against generic
                if name != "generic" or i % 2:
                    offs[0] = 0
                assert O.sao_edge_ddistortion(orig[i], rec[i], bw, bh, eo, offs) == R.sao_edge_ddistortion(orig[i], rec[i], bw, bh, eo, offs, name)
            bands = g.integers(-7, 8, 4)
            bp = int(g.integers(0, 32))
            assert O.sao_band_ddistortion(orig[i], rec[i], bw, bh, bp, bands) == R.sao_band_ddistortion(orig[i], rec[i], bw, bh, bp, bands, name)


@pytest.mark.parametrize("color", [0, 1, 2])
def test_sao_reconstruct_color(color):
    g = rng(11 + color)
    plane = g.integers(0, 256, (80, 96), dtype=np.uint8)
    plane[10:30, 10:40] = np.where(g.integers(0, 2, (20, 30)) > 0, 250, 3)
    recs = sao_records(12, 3 + color)
    for name in ("generic", "avx2"):
        if not R.has_strategy("sao_reconstruct_color", name):
            continue
        for i, s in enumerate(recs):
            for (x, y, bw, bh) in ((1, 1, 64, 64), (5, 3, 32, 32), (1, 7, 61, 13), (17, 2, 8, 70)):
                np.testing.assert_array_equal(O.sao_reconstruct_color(plane, x, y, bw, bh, s, color),
                                              R.sao_reconstruct_color(plane, x, y, bw, bh, s, color, name), err_msg="%s rec %d %s" % (name, i, (x, y, bw, bh)))


def test_sao_info_layout_matches_the_mirror_used_by_the_drop_in():
    """strategy.hip reads sao_info_t through a mirror struct of 17 ints (sao.h:42-50)"""
    L = R.lib()
    L.ref_sizeof_sao_info.restype = C.c_int
    assert L.ref_sizeof_sao_info() == 17 * 4


def test_bipred_candidate_cost():
    """search_pu_inter_bipred's candidate score: luma of kvz_inter_recon_bipred + satd_any_size, for integer, half- and
    quarter-pel vector pairs, inside the frame and across its borders"""
    g = rng(91)
    pic, ref0 = me_frames(192, 128, 12, (2, -1))
    _, ref1 = me_frames(192, 128, 13, (-3, 2))
    ref1 = np.where(g.integers(0, 40, ref1.shape) == 0, 255, ref1).astype(np.uint8)
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (32, 64), (24, 8), (8, 4), (4, 8), (16, 4), (4, 16), (16, 12), (12, 16)):
        for k in range(14):
            # a PU never straddles an LCU (the reference writes the prediction into the 64x64 lcu->rec)
            x = int(g.integers(0, 3)) * 64 + int(g.integers(0, (64 - w) // 4 + 1)) * 4
            y = int(g.integers(0, 2)) * 64 + int(g.integers(0, (64 - h) // 4 + 1)) * 4
            big = 400 if k % 5 == 4 else 24
            mv0, mv1 = g.integers(-big, big + 1, 2), g.integers(-big, big + 1, 2)
            if k % 3 == 0:
                mv0 = (mv0 // 4) * 4
            if k % 4 == 1:
                mv1 = (mv1 // 4) * 4
            a = O.bipred_luma_satd(pic, ref0, ref1, x, y, w, h, mv0, mv1)
            b = R.bipred_luma_satd(pic, ref0, ref1, x, y, w, h, mv0, mv1)
            assert a[0] == b[0], (w, h, x, y, mv0, mv1)
            np.testing.assert_array_equal(a[1], b[1])


# ---- deblocking: the oracle filters every vertical edge of the frame, then every horizontal one; the
# reference goes LCU by LCU with its deferred rightmost 4 pixels -- the planes must come out identical ----
from patterns import deblock_case, deblock_params  # noqa: E402

DEBLOCK_CONFIGS = [dict(w=192, h=128, qp=34), dict(w=200, h=136, qp=38, beta=2, tc=-1), dict(w=128, h=64, qp=30, per_cu_qp=1),
                   dict(w=192, h=128, qp=36, slice_is_b=1), dict(w=136, h=72, qp=45, tc=3, per_cu_qp=1, slice_is_b=1),
                   dict(w=64, h=64, qp=22, beta=-3), dict(w=192, h=64, qp=40, chroma=0), dict(w=72, h=200, qp=51, beta=6, tc=6)]


@pytest.mark.parametrize("cfg", range(len(DEBLOCK_CONFIGS)))
def test_deblock_frame(cfg):
    c = dict(DEBLOCK_CONFIGS[cfg])
    w, h = c.pop("w"), c.pop("h")
    prm = deblock_params(**c)
    changed = 0
    for seed in range(3):
        y, u, v, cus = deblock_case(w, h, 100 * cfg + seed, slice_is_b=int(prm["slice_is_b"][0]), qp=int(prm["qp"][0]),
                                    intra_share=(0.35, 0.0, 1.0)[seed])
        want = R.deblock_frame(y, u, v, cus, prm)
        got = O.deblock_frame(y, u, v, cus, prm)
        for a, b, name in zip(got, want, "yuv"):
            np.testing.assert_array_equal(a, b, err_msg="plane %s cfg %d seed %d" % (name, cfg, seed))
        changed += int((want[0] != y).sum())
        if not int(prm["chroma"][0]):
            np.testing.assert_array_equal(want[1], u)
    assert changed > 0                      # the filter did something


# ---- the encoder's own searches, recorded and replayed ----
@pytest.mark.parametrize("opts", [
    "preset=medium,ref=1,bipred=0,gop=0,rdoq=0,qp=30,threads=0,smp=0,amp=0,period=0",
    "preset=fast,ref=1,bipred=0,gop=0,rdoq=0,qp=37,threads=0,me=dia,subme=2,deblock=1,sao=off,owf=0,wpp=0,period=0",
    "preset=medium,ref=1,bipred=0,gop=0,rdoq=0,qp=24,threads=0,me=tz,me-early-termination=sensitive,mv-constraint=frametilemargin,period=0",
])
def test_recorded_encoder_searches_replay(opts):
    """the reference ENCODER runs (kvz_api); the harness notes every 2Nx2N inter search with the candidates the encoder derived;
    the recorded decisions must be what the reference's static search (ref_me_harness.c) and the oracle compute from the
    recorded inputs -- and recording must not change the bitstream"""
    from patterns import front_groups
    w, h = 192, 128
    frames = R.synthetic_sequence(w, h, 3, seed=11)
    plain, _ = R.encode(frames, w, h, opts)
    rec = R.record_inter_searches(frames, w, h, opts)
    assert rec["bitstream"] == plain and rec["skipped"] == 0 and len(rec["pus"]) >= 2 * 6 * 85 // 2
    m = rec["meta"]
    for f in range(len(rec["pic"])):
        sel = np.where(m[:, 0] == f)[0]
        prm = rec["params"].copy()
        prm["lambda_cost"] = m[sel[0], 4]
        a = R.search_pu_batch(rec["pic"][f], rec["ref"][f], rec["pus"][sel], prm)
        b = O.search_pu_batch(rec["pic"][f], rec["ref"][f], rec["pus"][sel], prm)
        for fld in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
            np.testing.assert_array_equal(a[fld], rec["results"][sel][fld], err_msg="static search, %s" % fld)
            np.testing.assert_array_equal(b[fld], rec["results"][sel][fld], err_msg="oracle, %s" % fld)
        groups = front_groups(m[sel])
        assert sum(len(g_) for g_ in groups) == len(sel) and max(len(g_) for g_ in groups) <= 3      # <= LCUs on a wavefront of a 3 x 2 grid


# ---- --mv-rdo: MV bits from the CABAC model (rdo.c:883-1060) ----
MV_RDO_CONFIGS = [
    dict(mv_rdo=1), dict(mv_rdo=1, refs_before=3, ref_idx=2, lambda_cost=11), dict(mv_rdo=1, refs_before=2, ref_idx=0, algorithm=1, fme_level=2),
    dict(mv_rdo=1, refs_before=4, ref_idx=1, algorithm=2, early_termination=0), dict(mv_rdo=1, fme_level=0, refs_before=2, ref_idx=1),
    dict(mv_rdo=1, refs_before=5, ref_idx=4, mv_constraint=4, lambda_cost=40),
]


def test_cabac_tables_are_the_references():
    """the oracle restates ITU-T H.265 Tables 9-46 / 9-47 and the two derived ones; they must equal the reference's
    kvz_g_auc_lpst_table, kvz_g_auc_next_state_mps / _lps and kvz_g_auc_renorm_table (cabac.c:28-75)"""
    import ctypes as C
    L, LO = R.lib(), O.lib()
    L.ref_cabac_table.restype = C.c_int; L.ref_cabac_table.argtypes = [C.c_int, C.c_int]
    LO.orc_cabac_table.restype = C.c_int; LO.orc_cabac_table.argtypes = [C.c_int, C.c_int]
    for kind, n in ((0, 256), (1, 128), (2, 128), (3, 32)):
        assert [L.ref_cabac_table(kind, i) for i in range(n)] == [LO.orc_cabac_table(kind, i) for i in range(n)], kind


@pytest.mark.parametrize("cfg", range(len(MV_RDO_CONFIGS)))
def test_search_pu_mv_rdo(cfg):
    """the search with kvz_calc_mvd_cost_cabac as its cost function and kvz_get_mvd_coding_cost_cabac in select_mv_cand, from random
    (valid) CABAC states: the oracle's bit counting must follow the reference's counting-mode encoder exactly"""
    from patterns import me_cabac_states
    prm = me_params(**MV_RDO_CONFIGS[cfg])
    differs = 0
    for k, motion in enumerate(((3, -2), (-7, 5), (0, 0))):
        pic, ref = me_frames(192, 128, 900 + k, motion)
        pus = me_random_pus(192, 128, 40, 77 + 10 * cfg + k, hint=(-4 * motion[0] + 2, -4 * motion[1]))
        cab = me_cabac_states(7, 5 + k + cfg)
        pus["reserved"] = np.arange(len(pus)) % 7
        a, b = O.search_pu_batch(pic, ref, pus, prm, cabac=cab), R.search_pu_batch(pic, ref, pus, prm, cabac=cab)
        for f in ("mv", "cost", "bitcost", "merged", "merge_idx", "mv_cand"):
            np.testing.assert_array_equal(a[f], b[f], err_msg="%s cfg %d motion %s" % (f, cfg, motion))
        est = dict(MV_RDO_CONFIGS[cfg]); est.pop("mv_rdo")
        differs += int((O.search_pu_batch(pic, ref, pus, me_params(**est))["bitcost"] != a["bitcost"]).sum())
    assert differs > 60          # the CABAC bit counts are not the exp-Golomb estimate


# ---- AMVP / merge candidate derivation (the driver half): inter.c:1209-1446 ----
from patterns import INTER_CAND_CONFIGS, inter_cand_case  # noqa: E402


@pytest.mark.parametrize("name", [c[0] for c in INTER_CAND_CONFIGS])
def test_inter_candidates(name):
    """every inter PU of random CU maps (all partition modes, intra and unset neighbours, vectors at the int16 limits): the merge
    list, the AMVP pair for the searched picture, the start vector and the flattened merge view of the search descriptor"""
    total, seed = 0, 0
    while total < 400:
        p, cus, col, refm, pus = inter_cand_case(name, seed)
        seed += 1
        want_pus, want_merge = R.inter_candidates(p, cus, col, refm, pus)
        got_pus, got_merge = O.inter_candidates(p, cus, col, refm, pus)
        for i in range(len(pus)):
            where = "%s seed %d PU %s" % (name, seed - 1, tuple(int(pus[i][k]) for k in ("x", "y", "width", "height", "pad")))
            assert got_pus[i]["num_merge_cand"] == want_pus[i]["num_merge_cand"] == 5, where
            np.testing.assert_array_equal(got_merge[i], want_merge[i], err_msg=where)
            np.testing.assert_array_equal(got_pus[i]["mv_cand"], want_pus[i]["mv_cand"], err_msg=where)
            np.testing.assert_array_equal(got_pus[i]["extra_mv"], want_pus[i]["extra_mv"], err_msg=where)
        np.testing.assert_array_equal(got_pus, want_pus)
        total += len(pus)


def recorded_candidate_case(w=192, h=128, n_frames=4, opts="preset=medium,ref=1,bipred=0,gop=0,rdoq=0,qp=30,threads=0,smp=0,amp=0,period=0",
                            snapshots=4000, seed=11):
    frames = R.synthetic_sequence(w, h, n_frames, seed=seed)
    return R.record_inter_searches(frames, w, h, opts, snapshots=snapshots)


def candidates_from_snapshots(rec, derive):
    """for every snapshot: lcu->cu put back into a picture-sized SCU map whose other records are noise, the candidates derived by
    `derive(params, cus, col, ref_cus, pus)` -> completed descriptors; returns (derived, recorded) descriptor arrays"""
    from patterns import CU_INFO, place_lcu_snapshot
    g = rng(5)
    m, prm = rec["meta"], rec["snap_params"]
    got, want = [], []
    noise = None
    for k, r in enumerate(rec["snap_index"]):
        f = int(m[r, 0])
        col = rec["snap_col"][f]
        if noise is None:
            noise = np.zeros(col.shape, dtype=CU_INFO)
            noise["type"], noise["mv_dir"] = 2, 1
            noise["mv"] = g.integers(-500, 500, noise["mv"].shape)
        cus = noise.copy()
        place_lcu_snapshot(cus, rec["snap_cus"][k], int(m[r, 1]), int(m[r, 2]))
        pu = rec["pus"][r:r + 1].copy()
        for fld in ("mv_cand", "extra_mv", "num_merge_cand", "merge"):
            pu[fld] = 0
        got.append(derive(prm[f:f + 1], cus, col, col, pu)[0])
        want.append(rec["pus"][r])
    return np.array(got), np.array(want)


def test_candidates_on_recorded_encoder_states():
    """the candidates the reference ENCODER derived during real encodes (P frames, TMVP on: POC 1..3) against the oracle's
    derivation from snapshots of what the encoder's functions read -- lcu->cu with its work-tree leftovers, the collocated
    picture's CU array, the POC tables"""
    rec = recorded_candidate_case()
    assert len(rec["snap_index"]) >= 1000 and rec["skipped"] == 0
    got, want = candidates_from_snapshots(rec, lambda *a: O.inter_candidates(*a)[0])
    for fld in ("num_merge_cand", "merge", "mv_cand", "extra_mv"):
        np.testing.assert_array_equal(got[fld], want[fld], err_msg=fld)
    # the states are not trivial: spatial / temporal candidates and non-zero predictors occur
    assert (np.abs(want["mv_cand"]).sum(axis=(1, 2)) > 0).mean() > 0.3 and (np.abs(want["extra_mv"]).sum(axis=1) > 0).any()


def test_candidate_helpers_vs_the_reference_unit_test_functions():
    """is_a0_cand_coded / is_b0_cand_coded / get_spatial_merge_candidates themselves (file-local in inter.c, reached like
    tests/mv_cand_tests.c reaches them) for every PU of every partition mode, and the test's own known answers"""
    from patterns import MV_CAND_KAT_A0, MV_CAND_KAT_B0, MV_CAND_KAT_SPATIAL, valid_pu_geometries
    geoms = valid_pu_geometries(192)
    for a, b in zip(O.mv_cand_helpers(geoms, 192, 192), R.mv_cand_helpers(geoms, 192, 192)):
        np.testing.assert_array_equal(a, b)
    (x, y, w, h, pw, ph), want = MV_CAND_KAT_SPATIAL
    assert tuple(R.mv_cand_helpers([(x, y, w, h)], pw, ph)[2][0]) == want
    assert [bool(v) for v in R.mv_cand_helpers([g for g, _ in MV_CAND_KAT_A0], 1920, 1080)[0]] == [e for _, e in MV_CAND_KAT_A0]
    assert [bool(v) for v in R.mv_cand_helpers([g for g, _ in MV_CAND_KAT_B0], 1920, 1080)[1]] == [e for _, e in MV_CAND_KAT_B0]
